/*
 * ref_tests_driver.c -- TEST INFRASTRUCTURE ONLY.  Our own main() around the REFERENCE'S OWN test
 * functions (device/test/*.c, compiled from where they lie by `make -C oracle reftests`, with the
 * reference's own headers) LINKED AGAINST libseal_embedded_amd.so instead of device/lib: the reference's
 * test suite relinked, unchanged, onto the MI355X path ("the lower surface used by
 * tests/bench").  The reference's checks are se_assert()s (standard assert, user_defines.h:37): a
 * failing check aborts the process.
 *
 *   oracle/_ref/ref_tests_gpu <test> [n] [nprimes]      # CWD holds adapter_output_data/
 *
 * Only symbols the product library exports are resolved from it; the single reference .c file linked in
 * besides the tests is polymodmult.c (the schoolbook multiply the NTT test checks against -- test-only in
 * the reference as well, polymodmult.h:15).
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern void test_ckks_encode(size_t n);
extern void test_ckks_encode_encrypt_sym(size_t n, size_t nprimes);
extern void test_enc_zero_sym(size_t n, size_t nprimes);
extern void test_ckks_encode_encrypt_asym(size_t n, size_t nprimes);
extern void test_enc_zero_asym(size_t n, size_t nprimes);
extern void test_poly_mult_ntt(size_t n, size_t nprimes);
extern void test_fft(size_t n);
extern void test_sample_poly_uniform(size_t n);
extern void test_sample_poly_ternary(size_t n);
extern void test_sample_poly_ternary_small(size_t n);
extern void test_ckks_api_sym(void);
extern void test_ckks_api_asym(void);

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const char *t  = argv[1];
    size_t n       = argc > 2 ? (size_t)atol(argv[2]) : 4096;
    size_t nprimes = argc > 3 ? (size_t)atol(argv[3]) : 3;
    if (!strcmp(t, "encode")) test_ckks_encode(n);
    else if (!strcmp(t, "sym")) test_ckks_encode_encrypt_sym(n, nprimes);
    else if (!strcmp(t, "zero_sym")) test_enc_zero_sym(n, nprimes);
    else if (!strcmp(t, "asym")) test_ckks_encode_encrypt_asym(n, nprimes);
    else if (!strcmp(t, "zero_asym")) test_enc_zero_asym(n, nprimes);
    else if (!strcmp(t, "ntt")) test_poly_mult_ntt(n, nprimes);
    else if (!strcmp(t, "fft")) test_fft(n);
    else if (!strcmp(t, "uniform")) test_sample_poly_uniform(n);
    else if (!strcmp(t, "ternary")) test_sample_poly_ternary(n);
    else if (!strcmp(t, "ternary_small")) test_sample_poly_ternary_small(n);
    else if (!strcmp(t, "api_sym")) test_ckks_api_sym();
    else if (!strcmp(t, "api_asym")) test_ckks_api_asym();
    else return 2;
    fflush(stdout);
    fprintf(stderr, "REF_TEST_DONE %s %zu %zu\n", t, n, nprimes);
    return 0;
}
