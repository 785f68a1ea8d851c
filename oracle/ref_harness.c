/*
 * ref_harness.c -- thin driver around the *compiled reference* (SEAL-Embedded device/lib).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is our own code; it is compiled together with the
 * reference's sources where they lie under /root/reference (never copied into this repo) by
 * oracle/Makefile into oracle/_ref/libse_ref.so.  It exposes flat C entry points (ctypes
 * friendly) that imitate the calling pattern of device/test/ckks_tests_sym.c:103-172 and
 * device/test/ckks_tests_asym.c:120-208, with explicit seeds so outputs are reproducible.
 *
 * Used to (a) validate oracle/se_oracle.c, (b) generate tests/golden/ fixtures, (c) optionally
 * serve as the "reference" kind of CPU baseline in bench.py.
 *
 * The reference must be built with -fno-strict-aliasing: its reduce_pte_core reads a uint64
 * through a uint32* (ckks_common.c:226-230) and gcc -O3 miscompiles it otherwise.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "ckks_asym.h"
#include "ckks_common.h"
#include "ckks_sym.h"
#include "defines.h"
#include "fft.h"
#include "modulo.h"
#include "ntt.h"
#include "parameters.h"
#include "rng.h"
#include "sample.h"
#include "seal_embedded.h"
#include "uintmodarith.h"
#include "util_print.h"
#include "intt.h"
#include "ckks_tests_common.h"

/* The reference prints unconditionally (ckks_sym.c:153-155, seal_embedded.c:48).  Silence
 * stdout around calls so pytest / bench output stays clean. */
static int g_saved_stdout = -1;
static void hush(void)
{
    fflush(stdout);
    g_saved_stdout = dup(1);
    int dn         = open("/dev/null", O_WRONLY);
    dup2(dn, 1);
    close(dn);
}
static void unhush(void)
{
    fflush(stdout);
    if (g_saved_stdout >= 0)
    {
        dup2(g_saved_stdout, 1);
        close(g_saved_stdout);
        g_saved_stdout = -1;
    }
}

/* One independent reference instance (own pool, Parms, PRNGs) -- the lower-level ckks_*
 * functions take every pointer explicitly, so instances are usable from separate threads. */
typedef struct
{
    Parms parms;
    SE_PTRS ptrs;
    ZZ *pool;
    SE_PRNG prng, share_prng;
    ZZ *s_save, *c1_save, *u_save, *e1_save;
    int asym;
} refh;

void *refh_open(size_t n, size_t nprimes, int asym)
{
    refh *h = (refh *)calloc(1, sizeof(refh));
    hush();
    h->asym                = asym;
    h->parms.sample_s      = 0;
    h->parms.is_asymmetric = asym ? 1 : 0;
    h->parms.small_s       = 1;
    h->parms.small_u       = 1;
    h->parms.pk_from_file  = 0;
    if (asym)
    {
        h->pool = ckks_mempool_setup_asym(n);
        ckks_set_ptrs_asym(n, h->pool, &h->ptrs);
    }
    else
    {
        h->pool = ckks_mempool_setup_sym(n);
        ckks_set_ptrs_sym(n, h->pool, &h->ptrs);
    }
    ckks_setup(n, nprimes, h->ptrs.index_map_ptr, &h->parms);
    h->s_save  = (ZZ *)calloc(n, sizeof(ZZ));
    h->c1_save = (ZZ *)calloc(n, sizeof(ZZ));
    h->u_save  = (ZZ *)calloc(n, sizeof(ZZ));
    h->e1_save = (ZZ *)calloc(n, sizeof(ZZ));
    unhush();
    return h;
}

void refh_close(void *vh)
{
    refh *h = (refh *)vh;
    free(h->pool);
    free(h->s_save);
    free(h->c1_save);
    free(h->u_save);
    free(h->e1_save);
    delete_parameters(&h->parms);
    free(h);
}

/* parameter read-back */
double refh_scale(void *vh) { return ((refh *)vh)->parms.scale; }
void refh_moduli(void *vh, uint32_t *q, uint32_t *cr_lo, uint32_t *cr_hi)
{
    refh *h = (refh *)vh;
    for (size_t j = 0; j < h->parms.nprimes; j++)
    {
        q[j]     = h->parms.moduli[j].value;
        cr_lo[j] = h->parms.moduli[j].const_ratio[0];
        cr_hi[j] = h->parms.moduli[j].const_ratio[1];
    }
}
void refh_index_map(void *vh, uint16_t *out)
{
    refh *h = (refh *)vh;
    memcpy(out, h->ptrs.index_map_ptr, h->parms.coeff_count * sizeof(uint16_t));
}

/* sk in the 2-bit packed form the adapter writes to sk_<n>.dat (fileops.c:140-170) */
void refh_set_sk(void *vh, const uint8_t *sk_packed)
{
    refh *h = (refh *)vh;
    memcpy(h->ptrs.ternary, sk_packed, h->parms.coeff_count / 4);
}

/* --- stage-level entry points ------------------------------------------------------------ */

int refh_encode(void *vh, const float *v, size_t vlen, int64_t *out)
{
    refh *h  = (refh *)vh;
    size_t n = h->parms.coeff_count;
    memset(h->ptrs.values, 0, (n / 2) * sizeof(flpt));
    memcpy(h->ptrs.values, v, vlen * sizeof(flpt));
    /* conj_vals is fully overwritten by the scatter (index map is a bijection) */
    bool ok = ckks_encode_base(&h->parms, h->ptrs.values, n / 2, h->ptrs.index_map_ptr,
                               h->ptrs.ifft_roots, h->ptrs.conj_vals);
    if (out) memcpy(out, h->ptrs.conj_vals_int_ptr, n * sizeof(int64_t));
    return ok ? 1 : 0;
}

/* ckks_encode_base with the index at which it gave up: the reference reports that index only in the
 * message it prints before `return false` (ckks_common.c:195-204), so stdout is captured into a
 * temporary file around the call and the message parsed.  Returns n on success.  out[0 .. index) are the
 * coefficients the reference's in-place loop had converted by then. */
long refh_encode_ex(void *vh, const float *v, size_t vlen, int64_t *out)
{
    refh *h  = (refh *)vh;
    size_t n = h->parms.coeff_count;
    memset(h->ptrs.values, 0, (n / 2) * sizeof(flpt));
    memcpy(h->ptrs.values, v, vlen * sizeof(flpt));
    fflush(stdout);
    FILE *cap = tmpfile();
    int saved = dup(1);
    dup2(fileno(cap), 1);
    bool ok = ckks_encode_base(&h->parms, h->ptrs.values, n / 2, h->ptrs.index_map_ptr,
                               h->ptrs.ifft_roots, h->ptrs.conj_vals);
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
    long idx = (long)n;
    if (!ok)
    {
        char line[256];
        idx = -1;
        rewind(cap);
        while (fgets(line, sizeof line, cap))
        {
            size_t i;
            if (sscanf(line, "Error! Value at index %zu", &i) == 1) idx = (long)i;
        }
    }
    fclose(cap);
    if (out) memcpy(out, h->ptrs.conj_vals_int_ptr, n * sizeof(int64_t));
    return idx;
}

void refh_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen)
{
    shake256(out, outlen, in, inlen);
}

void refh_prng_block(const uint8_t *seed, uint64_t ctr, uint8_t *out, size_t outlen)
{
    SE_PRNG p;
    memcpy(p.seed, seed, SE_PRNG_SEED_BYTE_COUNT);
    p.counter = ctr;
    prng_fill_buffer(outlen, &p, out);
}

static void set_prime(refh *h, size_t j)
{
    ckks_reset_primes(&h->parms);
    for (size_t i = 0; i < j; i++) next_modulus(&h->parms);
}

uint64_t refh_sample_uniform(void *vh, size_t j, const uint8_t *seed, uint64_t ctr, uint32_t *out)
{
    refh *h = (refh *)vh;
    set_prime(h, j);
    SE_PRNG p;
    memcpy(p.seed, seed, SE_PRNG_SEED_BYTE_COUNT);
    p.counter = ctr;
    sample_poly_uniform(&h->parms, &p, out);
    return p.counter;
}

uint64_t refh_sample_ternary_small(void *vh, const uint8_t *seed, uint64_t ctr, uint8_t *packed)
{
    refh *h = (refh *)vh;
    SE_PRNG p;
    memcpy(p.seed, seed, SE_PRNG_SEED_BYTE_COUNT);
    p.counter = ctr;
    sample_small_poly_ternary_prng_96(h->parms.coeff_count, &p, (ZZ *)packed);
    return p.counter;
}

uint64_t refh_cbd_int8(void *vh, const uint8_t *seed, uint64_t ctr, int8_t *out)
{
    refh *h = (refh *)vh;
    SE_PRNG p;
    memcpy(p.seed, seed, SE_PRNG_SEED_BYTE_COUNT);
    p.counter = ctr;
    sample_poly_cbd_generic_prng_16(h->parms.coeff_count, &p, out);
    return p.counter;
}

uint64_t refh_cbd_add(void *vh, const uint8_t *seed, uint64_t ctr, int64_t *inout)
{
    refh *h = (refh *)vh;
    SE_PRNG p;
    memcpy(p.seed, seed, SE_PRNG_SEED_BYTE_COUNT);
    p.counter = ctr;
    sample_add_poly_cbd_generic_inpl_prng_16(inout, h->parms.coeff_count, &p);
    return p.counter;
}

void refh_expand_ternary(void *vh, size_t j, const uint8_t *packed, uint32_t *out)
{
    refh *h = (refh *)vh;
    set_prime(h, j);
    expand_poly_ternary((const ZZ *)packed, &h->parms, out);
}

void refh_ntt_roots(void *vh, size_t j, uint32_t *roots)
{
    refh *h = (refh *)vh;
    set_prime(h, j);
    ntt_roots_initialize(&h->parms, roots);
}

void refh_ntt(void *vh, size_t j, uint32_t *vec)
{
    refh *h  = (refh *)vh;
    size_t n = h->parms.coeff_count;
    set_prime(h, j);
    ZZ *roots = (ZZ *)malloc(n * sizeof(ZZ));
    ntt_roots_initialize(&h->parms, roots);
    ntt_inpl(&h->parms, roots, vec);
    free(roots);
}

void refh_reduce_pte(void *vh, size_t j, const int64_t *in, uint32_t *out)
{
    refh *h = (refh *)vh;
    set_prime(h, j);
    reduce_set_pte(&h->parms, in, out);
}

void refh_reduce_e_small(void *vh, size_t j, const int8_t *e, uint32_t *out)
{
    refh *h = (refh *)vh;
    set_prime(h, j);
    reduce_set_e_small(&h->parms, e, out);
}

/* word-level KAT access (device/test/modulo_tests.c, uintmodarith_tests.c) */
uint32_t refh_barrett32(void *vh, size_t j, uint32_t x)
{
    return barrett_reduce_32input_32modulus(x, &((refh *)vh)->parms.moduli[j]);
}
uint32_t refh_barrett64(void *vh, size_t j, uint64_t x)
{
    uint32_t w[2] = {(uint32_t)x, (uint32_t)(x >> 32)};
    return barrett_reduce_64input_32modulus(w, &((refh *)vh)->parms.moduli[j]);
}
uint32_t refh_mul_mod(void *vh, size_t j, uint32_t a, uint32_t b)
{
    return mul_mod(a, b, &((refh *)vh)->parms.moduli[j]);
}

/* --- whole-path: symmetric (device/test/ckks_tests_sym.c:121-172 calling pattern) ---------- */
int refh_encrypt_sym(void *vh, const float *v, size_t vlen, const uint8_t *share_seed,
                     const uint8_t *seed, uint32_t *c0 /*[np][n]*/, uint32_t *c1_a /*[np][n]*/,
                     uint32_t *c1_alias /*[np][n] or NULL*/, int64_t *pte /*[n] or NULL*/,
                     uint32_t *ntt_s /*[np][n] or NULL*/, uint64_t *end_ctr)
{
    refh *h  = (refh *)vh;
    size_t n = h->parms.coeff_count;
    uint8_t sseed[64], eseed[64];
    memcpy(sseed, share_seed, 64);
    memcpy(eseed, seed, 64);
    ckks_reset_primes(&h->parms);
    if (!refh_encode(vh, v, vlen, NULL)) return 0;
    ckks_sym_init(&h->parms, sseed, eseed, &h->share_prng, &h->prng, h->ptrs.conj_vals_int_ptr);
    if (pte) memcpy(pte, h->ptrs.conj_vals_int_ptr, n * sizeof(int64_t));
    for (size_t j = 0; j < h->parms.nprimes; j++)
    {
        ckks_encode_encrypt_sym(&h->parms, h->ptrs.conj_vals_int_ptr, NULL, &h->share_prng,
                                h->ptrs.ternary, h->ptrs.ntt_pte_ptr, h->ptrs.ntt_roots_ptr,
                                h->ptrs.c0_ptr, h->ptrs.c1_ptr, h->s_save, h->c1_save);
        memcpy(c0 + j * n, h->ptrs.c0_ptr, n * sizeof(ZZ));
        memcpy(c1_a + j * n, h->c1_save, n * sizeof(ZZ));
        if (c1_alias) memcpy(c1_alias + j * n, h->ptrs.c1_ptr, n * sizeof(ZZ));
        if (ntt_s) memcpy(ntt_s + j * n, h->s_save, n * sizeof(ZZ));
        if (j + 1 < h->parms.nprimes) ckks_next_prime_sym(&h->parms, h->ptrs.ternary);
    }
    if (end_ctr) *end_ctr = h->share_prng.counter;
    return 1;
}

/* --- whole-path: asymmetric with an in-memory public key ---------------------------------- */
int refh_encrypt_asym(void *vh, const float *v, size_t vlen, const uint8_t *seed,
                      const uint32_t *pk0 /*[np][n]*/, const uint32_t *pk1, uint32_t *c0,
                      uint32_t *c1, int64_t *pte, uint8_t *u_packed, int8_t *e1,
                      uint64_t *end_ctr)
{
    refh *h  = (refh *)vh;
    size_t n = h->parms.coeff_count;
    uint8_t eseed[64];
    memcpy(eseed, seed, 64);
    ckks_reset_primes(&h->parms);
    if (!refh_encode(vh, v, vlen, NULL)) return 0;
    ckks_asym_init(&h->parms, eseed, &h->prng, h->ptrs.conj_vals_int_ptr, h->ptrs.ternary,
                   h->ptrs.e1_ptr);
    if (pte) memcpy(pte, h->ptrs.conj_vals_int_ptr, n * sizeof(int64_t));
    if (u_packed) memcpy(u_packed, h->ptrs.ternary, n / 4);
    if (e1) memcpy(e1, h->ptrs.e1_ptr, n);
    if (end_ctr) *end_ctr = h->prng.counter;
    for (size_t j = 0; j < h->parms.nprimes; j++)
    {
        memcpy(h->ptrs.c0_ptr, pk0 + j * n, n * sizeof(ZZ));
        memcpy(h->ptrs.c1_ptr, pk1 + j * n, n * sizeof(ZZ));
        ckks_encode_encrypt_asym(&h->parms, h->ptrs.conj_vals_int_ptr, h->ptrs.ternary,
                                 h->ptrs.e1_ptr, h->ptrs.ntt_roots_ptr, h->ptrs.ntt_pte_ptr,
                                 h->u_save, h->e1_save, h->ptrs.c0_ptr, h->ptrs.c1_ptr);
        memcpy(c0 + j * n, h->ptrs.c0_ptr, n * sizeof(ZZ));
        memcpy(c1 + j * n, h->ptrs.c1_ptr, n * sizeof(ZZ));
        if (j + 1 < h->parms.nprimes) ckks_next_prime_asym(&h->parms, h->ptrs.ternary);
    }
    return 1;
}

/* --- public key from fixed seeds through the reference's own gen_pk ------------------------ */
void refh_gen_pk(size_t n, size_t nprimes, const uint8_t *sk_packed, const uint8_t *pk_seed,
                 const uint8_t *ep_seed, uint32_t *pk0, uint32_t *pk1)
{
    refh *h = (refh *)refh_open(n, nprimes, 0);
    refh_set_sk(h, sk_packed);
    int8_t *ep  = (int8_t *)malloc(n);
    ZZ *ntt_ep  = (ZZ *)malloc(n * sizeof(ZZ));
    ZZ *p0      = (ZZ *)malloc(n * sizeof(ZZ));
    ZZ *p1      = (ZZ *)malloc(n * sizeof(ZZ));
    SE_PRNG epr;
    memcpy(epr.seed, ep_seed, 64);
    epr.counter = 0;
    sample_poly_cbd_generic_prng_16(n, &epr, ep);
    uint8_t seedbuf[64];
    ckks_reset_primes(&h->parms);
    for (size_t j = 0; j < nprimes; j++)
    {
        memcpy(seedbuf, pk_seed, 64);
        gen_pk(&h->parms, h->ptrs.ternary, h->ptrs.ntt_roots_ptr, seedbuf, &h->share_prng,
               h->s_save, ep, ntt_ep, p0, p1);
        memcpy(pk0 + j * n, p0, n * sizeof(ZZ));
        memcpy(pk1 + j * n, p1, n * sizeof(ZZ));
        if (j + 1 < nprimes) ckks_next_prime_sym(&h->parms, h->ptrs.ternary);
    }
    free(ep);
    free(ntt_ep);
    free(p0);
    free(p1);
    refh_close(h);
}

/* --- API level: se_setup + se_encrypt_seeded with a collecting callback -------------------
 * Needs <SE_DATA_PATH>/sk_<n>.dat (and pk files for asym) relative to the CWD; the Python side
 * prepares a temp dir and chdir()s.  Returns bytes delivered, or -1 if the encode failed. */
static uint8_t *g_sink;
static size_t g_sink_len, g_sink_cap, g_ncalls;
static size_t sink_cb(void *data, size_t len)
{
    if (g_sink_len + len <= g_sink_cap) memcpy(g_sink + g_sink_len, data, len);
    g_sink_len += len;
    g_ncalls++;
    return len;
}

long refh_api_encrypt(size_t n, size_t nprimes, int asym, const float *v, size_t vlen_bytes,
                      const uint8_t *share_seed, const uint8_t *seed, uint8_t *out, size_t cap,
                      size_t *ncalls)
{
    uint8_t s1[64], s2[64];
    memcpy(s1, share_seed, 64);
    memcpy(s2, seed, 64);
    hush();
    SE_PARMS *sp = se_setup(n, nprimes, 0.0, asym ? SE_ASYM_ENCR : SE_SYM_ENCR);
    g_sink       = out;
    g_sink_len   = 0;
    g_sink_cap   = cap;
    g_ncalls     = 0;
    bool ok = se_encrypt_seeded(s1, s2, sink_cb, (void *)v, vlen_bytes, false, sp);
    se_cleanup(sp);
    unhush();
    if (ncalls) *ncalls = g_ncalls;
    return ok ? (long)g_sink_len : -1;
}

/* the same call with print = true; the library's stdout (incl. its unconditional setup chatter, T2)
 * goes to `path`, the caller filters the "c0: " / "c1: " lines (seal_embedded.c:160-163) */
long refh_api_encrypt_print(size_t n, size_t nprimes, int asym, const float *v, size_t vlen_bytes,
                            const uint8_t *share_seed, const uint8_t *seed, const char *path)
{
    uint8_t s1[64], s2[64];
    memcpy(s1, share_seed, 64);
    memcpy(s2, seed, 64);
    fflush(stdout);
    int saved = dup(1);
    int fd    = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    dup2(fd, 1);
    close(fd);
    SE_PARMS *sp = se_setup(n, nprimes, 0.0, asym ? SE_ASYM_ENCR : SE_SYM_ENCR);
    bool ok      = se_encrypt_seeded(s1, s2, NULL, (void *)v, vlen_bytes, true, sp);
    se_cleanup(sp);
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
    return ok ? 1 : 0;
}

/* --- timed CPU baseline over the reference itself: bench_sym.c:96-130 region ---------------
 * (encode + ckks_sym_init + per-prime ckks_encode_encrypt_sym; keys resident), one reference
 * instance per thread over a contiguous shard of the batch. */
typedef struct
{
    size_t n, nprimes, lo, hi;
    const float *values;
    const uint8_t *share_seeds, *seeds, *sk;
    uint32_t *c0, *c1;
    refh *h;
} ref_job;

static void *ref_worker(void *arg)
{
    ref_job *jb  = (ref_job *)arg;
    size_t n = jb->n, np = jb->nprimes;
    uint32_t *s0 = (uint32_t *)malloc(np * n * 4), *s1 = (uint32_t *)malloc(np * n * 4);
    for (size_t b = jb->lo; b < jb->hi; b++)
    {
        uint32_t *o0 = jb->c0 ? jb->c0 + b * np * n : s0;
        uint32_t *o1 = jb->c1 ? jb->c1 + b * np * n : s1;
        refh_encrypt_sym(jb->h, jb->values + b * (n / 2), n / 2, jb->share_seeds + 64 * b,
                         jb->seeds + 64 * b, o0, o1, NULL, NULL, NULL, NULL);
    }
    free(s0);
    free(s1);
    return NULL;
}

int refh_encrypt_sym_batch(size_t n, size_t nprimes, const float *values, size_t B,
                           const uint8_t *share_seeds, const uint8_t *seeds, const uint8_t *sk,
                           uint32_t *c0, uint32_t *c1, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    ref_job *jobs = (ref_job *)malloc(nthreads * sizeof(ref_job));
    for (int t = 0; t < nthreads; t++)
    { /* instances are created serially: refh_open redirects stdout */
        ref_job jb = {n, nprimes, B * t / nthreads, B * (t + 1) / nthreads, values, share_seeds,
                      seeds, sk, c0, c1, (refh *)refh_open(n, nprimes, 0)};
        refh_set_sk(jb.h, sk);
        jobs[t] = jb;
    }
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, ref_worker, &jobs[t]);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; t++) refh_close(jobs[t].h);
    free(th);
    free(jobs);
    return 1;
}

/* --- the same for the public-key path (bench_asym.c region: encode + ckks_asym_init + per-prime
 *     ckks_encode_encrypt_asym, pk resident in memory) and for encode-only (BASELINE config 5:
 *     ckks_encode_base + per prime reduce_set_pte + ntt_inpl) ------------------------------------ */
typedef struct
{
    size_t n, nprimes, lo, hi;
    const float *values;
    const uint8_t *seeds;
    const uint32_t *pk0, *pk1;
    uint32_t *c0, *c1;
    refh *h;
    int mode; /* 1 asym, 2 encode + ntt */
} ref_job2;

static void *ref_worker2(void *arg)
{
    ref_job2 *jb = (ref_job2 *)arg;
    size_t n = jb->n, np = jb->nprimes;
    uint32_t *s0 = (uint32_t *)malloc(np * n * 4), *s1 = (uint32_t *)malloc(np * n * 4);
    for (size_t b = jb->lo; b < jb->hi; b++)
    {
        uint32_t *o0 = jb->c0 ? jb->c0 + b * np * n : s0;
        uint32_t *o1 = jb->c1 ? jb->c1 + b * np * n : s1;
        if (jb->mode == 1)
            refh_encrypt_asym(jb->h, jb->values + b * (n / 2), n / 2, jb->seeds + 64 * b, jb->pk0, jb->pk1,
                              o0, o1, NULL, NULL, NULL, NULL);
        else
        {
            refh *h = jb->h;
            ckks_reset_primes(&h->parms);
            refh_encode(h, jb->values + b * (n / 2), n / 2, NULL);
            for (size_t j = 0; j < np; j++)
            {
                reduce_set_pte(&h->parms, h->ptrs.conj_vals_int_ptr, o0 + j * n);
                ntt_roots_initialize(&h->parms, h->ptrs.ntt_roots_ptr);
                ntt_inpl(&h->parms, h->ptrs.ntt_roots_ptr, o0 + j * n);
                if (j + 1 < np) next_modulus(&h->parms);
            }
        }
    }
    free(s0);
    free(s1);
    return NULL;
}

static int ref_batch2(size_t n, size_t nprimes, int mode, const float *values, size_t B,
                      const uint8_t *seeds, const uint32_t *pk0, const uint32_t *pk1, uint32_t *c0,
                      uint32_t *c1, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    pthread_t *th  = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    ref_job2 *jobs = (ref_job2 *)malloc(nthreads * sizeof(ref_job2));
    for (int t = 0; t < nthreads; t++)
    { /* instances are created serially: refh_open redirects stdout */
        ref_job2 jb = {n, nprimes, B * t / nthreads, B * (t + 1) / nthreads, values, seeds, pk0, pk1,
                       c0, c1, (refh *)refh_open(n, nprimes, mode == 1), mode};
        jobs[t] = jb;
    }
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, ref_worker2, &jobs[t]);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; t++) refh_close(jobs[t].h);
    free(th);
    free(jobs);
    return 1;
}

int refh_encrypt_asym_batch(size_t n, size_t nprimes, const float *values, size_t B, const uint8_t *seeds,
                            const uint32_t *pk0, const uint32_t *pk1, uint32_t *c0, uint32_t *c1,
                            int nthreads)
{
    return ref_batch2(n, nprimes, 1, values, B, seeds, pk0, pk1, c0, c1, nthreads);
}

int refh_encode_ntt_batch(size_t n, size_t nprimes, const float *values, size_t B, uint32_t *out,
                          int nthreads)
{
    return ref_batch2(n, nprimes, 2, values, B, NULL, NULL, NULL, out, NULL, nthreads);
}

/* --- the reference's own text printers, captured into a file (format pin for se_formats.cpp) --- */
void refh_print_to_file(const char *path, const char *name, const uint32_t *poly, size_t n,
                        const float *values, size_t vlen)
{
    fflush(stdout);
    int saved = dup(1);
    int fd    = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    dup2(fd, 1);
    close(fd);
    if (values) print_poly_flpt_full("v (cleartext)", values, vlen);
    if (poly) print_poly_full(name, poly, n);
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
}

/* --- verification side: the reference's own test helpers (device/test/ckks_tests_common.c) ---- */
void refh_intt(void *vh, size_t j, uint32_t *vec)
{
    refh *h  = (refh *)vh;
    size_t n = h->parms.coeff_count;
    set_prime(h, j);
    hush();
    ZZ *roots = (ZZ *)malloc(n * sizeof(ZZ));
    intt_roots_initialize(&h->parms, roots);
    intt_inpl(&h->parms, roots, vec);
    free(roots);
    unhush();
}

void refh_decrypt(void *vh, size_t j, const uint32_t *c0, const uint32_t *c1, const uint32_t *ntt_s,
                  uint32_t *out)
{
    refh *h = (refh *)vh;
    set_prime(h, j);
    ckks_decrypt(c0, c1, ntt_s, false, &h->parms, out);
}

void refh_decode(void *vh, size_t j, const uint32_t *pt, size_t values_len, float *out)
{
    refh *h  = (refh *)vh;
    size_t n = h->parms.coeff_count;
    set_prime(h, j);
    hush();
    double complex *tmp = (double complex *)calloc(n, sizeof(double complex));
    ckks_decode(pt, values_len, h->ptrs.index_map_ptr, &h->parms, tmp, out);
    free(tmp);
    unhush();
}
