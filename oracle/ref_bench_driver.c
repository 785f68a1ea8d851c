/* ref_bench_driver.c -- TEST INFRASTRUCTURE (build container only; the binary travels under oracle/_ref/).
 *
 * Runs the reference's OWN benchmark functions (device/bench/bench_*.c, compiled from where they lie with
 * the reference's headers and -DSE_ENABLE_TIMERS, timed by the reference's own timer.c) linked against the
 * PRODUCT library: every ckks_* / ntt / fft / sampler call inside them is a batch-of-one call into the
 * MI355X kernels.  What the numbers mean: single-call latency of the lower surface, not throughput.
 *
 *   ref_bench_gpu <sym|asym|ifft|ntt|uniform|ternary|cbd>     (cwd must hold adapter_output_data/)
 */
#include <stdio.h>
#include <string.h>

extern void bench_ifft(void);
extern void bench_ntt(void);
extern void bench_sample_uniform(void);
extern void bench_sample_ternary_small(void);
extern void bench_sample_poly_cbd(void);
extern void bench_sym(void);
extern void bench_asym(void);

int main(int argc, char **argv)
{
    if (argc < 2)
    {
        fprintf(stderr, "usage: %s <sym|asym|ifft|ntt|uniform|ternary|cbd>\n", argv[0]);
        return 2;
    }
    const char *t = argv[1];
    if (!strcmp(t, "sym")) bench_sym();
    else if (!strcmp(t, "asym")) bench_asym();
    else if (!strcmp(t, "ifft")) bench_ifft();
    else if (!strcmp(t, "ntt")) bench_ntt();
    else if (!strcmp(t, "uniform")) bench_sample_uniform();
    else if (!strcmp(t, "ternary")) bench_sample_ternary_small();
    else if (!strcmp(t, "cbd")) bench_sample_poly_cbd();
    else
    {
        fprintf(stderr, "unknown benchmark %s\n", t);
        return 2;
    }
    printf("ref-bench %s done\n", t);
    return 0;
}
