/*
 * oracle_selftest.c -- TEST INFRASTRUCTURE ONLY: drives every entry point of se_oracle.c once under
 * AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle sanitize`).  It checks internal
 * consistency only (round trips, determinism, the exact pseudo-decrypt identity the reference's own
 * tests use, device/test/ckks_tests_common.c:206); the pinning against the reference's golden
 * vectors is tests/test_oracle.py's job.  The point of this binary is that the sanitizers watch the
 * checker's memory accesses and integer arithmetic while all of its code runs, including the
 * threaded batch drivers and the edge inputs (extreme magnitudes, overflow, all shapes).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "se_oracle.h"

#define CHECK(c)                                                        \
    do                                                                  \
    {                                                                   \
        if (!(c))                                                       \
        {                                                               \
            fprintf(stderr, "selftest FAILED: %s (line %d)\n", #c, __LINE__); \
            exit(1);                                                    \
        }                                                               \
    } while (0)

static void fill_seed(uint8_t *s, int tag)
{
    for (int i = 0; i < 64; i++) s[i] = (uint8_t)(tag * 37 + i * 11 + 5);
}

static void run_shape(size_t n, size_t np)
{
    seo_params p;
    CHECK(seo_params_init(&p, n, np) == 0);
    uint16_t *map = malloc(n * sizeof(uint16_t));
    seo_index_map(n, p.logn, map);
    uint8_t seen[16384] = {0};
    for (size_t i = 0; i < n; i++)
    {
        CHECK(map[i] < n && !seen[map[i]]);
        seen[map[i]] = 1;
    }
    uint8_t s1[64], s2[64], s3[64];
    fill_seed(s1, 1), fill_seed(s2, 2), fill_seed(s3, 3);

    /* word layer on extreme operands */
    for (size_t j = 0; j < np; j++)
    {
        uint32_t q = p.q[j];
        CHECK(seo_barrett32(0xFFFFFFFFu, &p, j) == 0xFFFFFFFFu % q);
        CHECK(seo_barrett64(0xFFFFFFFFFFFFFFFFull, &p, j) == (uint32_t)(0xFFFFFFFFFFFFFFFFull % q));
        CHECK(seo_mul_mod(q - 1, q - 1, &p, j) == 1);
        CHECK(seo_add_mod(q - 1, q - 1, q) == q - 2 && seo_neg_mod(0, q) == 0 && seo_sub_mod(0, 1, q) == q - 1);
    }

    /* secret key by the ternary sampler, public key by gen_pk */
    uint8_t *sk = malloc(n / 4);
    uint64_t ctr = 0;
    seo_sample_ternary_small(n, s3, &ctr, sk);
    CHECK(ctr >= (n + 95) / 96);
    uint32_t *pk0 = malloc(np * n * 4), *pk1 = malloc(np * n * 4);
    seo_gen_pk(&p, sk, s1, s2, pk0, pk1);

    float *v = malloc((n / 2) * sizeof(float));
    for (size_t i = 0; i < n / 2; i++) v[i] = (float)((double)((i * 2654435761ull) % 1000ull) / 1000 - 0.5); /* fits every parameter set */
    uint32_t *c0 = malloc(np * n * 4), *c1 = malloc(np * n * 4), *ntt_pte = malloc(np * n * 4);
    uint32_t *d0 = malloc(np * n * 4), *d1 = malloc(np * n * 4);
    int64_t *pte = malloc(n * sizeof(int64_t));
    uint64_t end = 0;
    CHECK(seo_encrypt_sym(&p, map, v, n / 2, s1, s2, sk, c0, c1, pte, ntt_pte, &end) == 1);
    CHECK(end >= np);
    /* exact pseudo-decrypt: c0 + c1 . NTT(s) == NTT(m + e)   (ckks_tests_common.c:206) */
    uint32_t *s_exp = malloc(n * 4), *roots = malloc(n * 4), *dec = malloc(n * 4);
    for (size_t j = 0; j < np; j++)
    {
        seo_expand_ternary(sk, n, p.q[j], s_exp);
        seo_ntt_roots(&p, j, roots);
        seo_ntt_inpl(&p, j, roots, s_exp);
        seo_decrypt(&p, j, c0 + j * n, c1 + j * n, s_exp, dec);
        CHECK(memcmp(dec, ntt_pte + j * n, n * 4) == 0);
        /* INTT(NTT(x)) == x, decode within the reference's 0.1 (ckks_tests_common.c:132) */
        seo_intt_inpl(&p, j, dec);
        uint32_t *red = malloc(n * 4);
        seo_reduce_pte(&p, j, pte, red);
        for (size_t i = 0; i < n; i++) CHECK(dec[i] == red[i] % p.q[j]);
        float *back = malloc((n / 2) * sizeof(float));
        seo_decode(&p, j, map, dec, n / 2, back);
        for (size_t i = 0; i < n / 2; i++) CHECK(fabsf(back[i] - v[i]) < 0.1f);
        free(back);
        free(red);
    }
    /* determinism + the public-key path */
    CHECK(seo_encrypt_sym(&p, map, v, n / 2, s1, s2, sk, d0, d1, NULL, NULL, NULL) == 1);
    CHECK(memcmp(c0, d0, np * n * 4) == 0 && memcmp(c1, d1, np * n * 4) == 0);
    uint8_t *u = malloc(n / 4);
    int8_t *e1 = malloc(n);
    CHECK(seo_encrypt_asym(&p, map, v, n / 2, s2, pk0, pk1, c0, c1, pte, u, e1, &end) == 1);
    for (size_t j = 0; j < np; j++)
        for (size_t i = 0; i < n; i++) CHECK(c0[j * n + i] < p.q[j] && c1[j * n + i] < p.q[j]);

    /* extreme plaintexts: int64 limits through the signed reduction, 2^63 exactly, overflow */
    int64_t *big = calloc(n, sizeof(int64_t));
    big[0] = INT64_MAX, big[1] = INT64_MIN, big[2] = -(int64_t)p.q[0], big[3] = INT64_MIN + 1;
    uint32_t *red = malloc(n * 4);
    seo_reduce_pte(&p, 0, big, red);
    CHECK(red[2] == p.q[0]); /* the reference's non-canonical q (ckks_common.c:234) */
    for (size_t i = 0; i < n / 2; i++) v[i] = 3.0e38f;
    CHECK(seo_encode(&p, v, n / 2, map, pte) == 0);
    for (size_t i = 0; i < n / 2; i++) v[i] = (float)ldexp(1.0, 63) / (float)p.scale;
    (void)seo_encode(&p, v, n / 2, map, pte);

    /* threaded batch drivers (shard boundaries, scratch per thread) */
    size_t B = 5;
    float *vb = malloc(B * (n / 2) * sizeof(float));
    uint8_t *ss = malloc(B * 64), *sd = malloc(B * 64);
    for (size_t b = 0; b < B; b++)
    {
        for (size_t i = 0; i < n / 2; i++) vb[b * (n / 2) + i] = (float)((int)((i + 7 * b) % 256)) / -10.0f;
        fill_seed(ss + 64 * b, 10 + (int)b), fill_seed(sd + 64 * b, 20 + (int)b);
    }
    uint32_t *b0 = malloc(B * np * n * 4), *b1 = malloc(B * np * n * 4);
    CHECK(seo_encrypt_sym_batch(&p, vb, B, ss, sd, sk, b0, b1, 3) == 1);
    CHECK(seo_encrypt_sym(&p, map, vb + 4 * (n / 2), n / 2, ss + 256, sd + 256, sk, d0, d1, NULL, NULL, NULL) == 1);
    CHECK(memcmp(b0 + 4 * np * n, d0, np * n * 4) == 0 && memcmp(b1 + 4 * np * n, d1, np * n * 4) == 0);
    CHECK(seo_encrypt_asym_batch(&p, vb, B, sd, pk0, pk1, b0, b1, 2) == 1);
    CHECK(seo_encode_ntt_batch(&p, vb, B, b0, 4) == 1);
    CHECK(seo_encrypt_sym_batch(&p, vb, B, ss, sd, sk, NULL, NULL, 2) == 1);

    /* FFT pair and PRNG framing */
    double *x = malloc(2 * n * sizeof(double));
    for (size_t i = 0; i < 2 * n; i++) x[i] = (double)((i * 7919) % 1000) - 500.0;
    seo_ifft_inpl(x, n, p.logn);
    seo_fft_inpl(x, n, p.logn);
    for (size_t i = 0; i < 2 * n; i++)
        CHECK(fabs(x[i] / (double)n - ((double)((i * 7919) % 1000) - 500.0)) < 1e-6);
    uint8_t blk[300], h1[32], h2[32];
    seo_prng_block(s1, 0xFFFFFFFFFFFFFFFFull, blk, sizeof(blk));
    uint8_t msg[72];
    memcpy(msg, s1, 64);
    memset(msg + 64, 0xFF, 8);
    seo_shake256(h1, 32, msg, 72);
    memcpy(h2, blk, 32);
    CHECK(memcmp(h1, h2, 32) == 0);
    CHECK(seo_fnv1a64("a", 1, 0) == 0xaf63dc4c8601ec8cull);

    free(x); free(b0); free(b1); free(vb); free(ss); free(sd); free(red); free(big); free(u); free(e1);
    free(s_exp); free(roots); free(dec); free(c0); free(c1); free(d0); free(d1); free(ntt_pte); free(pte);
    free(v); free(pk0); free(pk1); free(sk); free(map);
    printf("oracle selftest ok: n=%zu primes=%zu\n", n, np);
}

int main(void)
{
    run_shape(1024, 1);
    run_shape(2048, 1);
    run_shape(4096, 3);
    run_shape(8192, 6);
    run_shape(16384, 13);
    seo_params bad;
    CHECK(seo_params_init(&bad, 3000, 1) != 0 && seo_params_init(&bad, 4096, 4) != 0);
    printf("oracle selftest: all shapes clean under ASan + UBSan\n");
    return 0;
}
