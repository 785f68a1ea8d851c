/*
 * se_oracle.h -- CPU restatement of SEAL-Embedded's device/lib encode+encrypt path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under seal-embedded_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker / reported CPU baseline.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py)
 * against (a) the reference's own known-answer values (device/test/modulo_tests.c,
 * uintmodarith_tests.c), (b) golden vectors under tests/golden/ that were produced by
 * the *compiled reference* (oracle/_ref, built from /root/reference/device/lib with
 * -O3 -fno-strict-aliasing; generator: tests/golden/make_golden.py), and (c) hashlib's
 * SHAKE256 for the PRNG layer.
 *
 * Only the reference's DEFAULT macro configuration is restated
 * (device/lib/user_defines.h:68,80,94,106: IFFT on-the-fly, NTT one-shot,
 * index map persistent, sk persistent, 32-bit residues, float inputs).
 */
#ifndef SE_ORACLE_H
#define SE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEO_MAX_PRIMES 13
#define SEO_SEED_BYTES 64

typedef struct
{
    size_t n;        /* polynomial degree (1024..16384, power of two) */
    size_t logn;
    size_t nprimes;
    uint32_t q[SEO_MAX_PRIMES];     /* modulus chain                      */
    uint32_t cr_lo[SEO_MAX_PRIMES]; /* floor(2^64/q) low word             */
    uint32_t cr_hi[SEO_MAX_PRIMES]; /* floor(2^64/q) high word            */
    uint32_t psi[SEO_MAX_PRIMES];   /* primitive 2n-th root per (n, q)    */
    double scale;
} seo_params;

/* parameters.c:176-230, modulus.c:23-56, ntt.c:199-291.  Returns 0 on success. */
int seo_params_init(seo_params *p, size_t n, size_t nprimes);

/* word arithmetic (modulo.h, uintmodarith.h) */
uint32_t seo_barrett32(uint32_t x, const seo_params *p, size_t j);
uint32_t seo_barrett64(uint64_t x, const seo_params *p, size_t j);
uint32_t seo_mul_mod(uint32_t a, uint32_t b, const seo_params *p, size_t j);
uint32_t seo_add_mod(uint32_t a, uint32_t b, uint32_t q);
uint32_t seo_neg_mod(uint32_t a, uint32_t q);
uint32_t seo_sub_mod(uint32_t a, uint32_t b, uint32_t q);

/* PRNG (rng.h:78-91, shake256/fips202.c:105-128, keccakf1600.c) */
void seo_keccak_f1600(uint64_t st[25]);
void seo_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen);
void seo_prng_block(const uint8_t seed[SEO_SEED_BYTES], uint64_t ctr, uint8_t *out, size_t outlen);

/* encode (ckks_common.c:32-68,105-215; fft.c:39-45,69-144; fft.h:48-55) */
size_t seo_bitrev(size_t x, size_t nbits);
void seo_index_map(size_t n, size_t logn, uint16_t *map /*[n]*/);
void seo_ifft_twiddles(size_t n, size_t logn, double *w_re_im /*[2n]*/);
void seo_ifft_inpl(double *x_re_im /*[2n] interleaved*/, size_t n, size_t logn);
/* returns 1 on success, 0 if a coefficient overflows int64 (reference returns false) */
size_t seo_encode_ex(const seo_params *p, const float *values, size_t values_len, const uint16_t *map,
                     int64_t *out /*[n]*/);   /* first failing index, n when none */
int seo_encode(const seo_params *p, const float *values, size_t values_len, const uint16_t *map,
               int64_t *out /*[n]*/);

/* samplers (sample.c) */
void seo_cbd_add(int64_t *poly, size_t n, const uint8_t seed[64], uint64_t *ctr);
void seo_cbd_int8(int8_t *poly, size_t n, const uint8_t seed[64], uint64_t *ctr);
void seo_sample_uniform(const seo_params *p, size_t j, const uint8_t seed[64], uint64_t *ctr,
                        uint32_t *poly /*[n]*/);
void seo_sample_ternary_small(size_t n, const uint8_t seed[64], uint64_t *ctr,
                              uint8_t *packed /*[n/4]*/);
void seo_expand_ternary(const uint8_t *packed, size_t n, uint32_t q, uint32_t *out);

/* NTT (ntt.c:24-60,124-189) */
void seo_ntt_roots(const seo_params *p, size_t j, uint32_t *roots /*[n]*/);
void seo_ntt_inpl(const seo_params *p, size_t j, const uint32_t *roots, uint32_t *vec);

/* RNS reduction (ckks_common.c:224-265) */
void seo_reduce_pte(const seo_params *p, size_t j, const int64_t *in, uint32_t *out);
void seo_reduce_e_small(const seo_params *p, size_t j, const int8_t *e, uint32_t *out);

/*
 * Whole-path entry points.  Output layout: c0/c1 are [nprimes][n] uint32, NTT form,
 * bit-reversed order -- the byte stream the reference hands to its send callback, prime by
 * prime (seal_embedded.c:145-205), except that c1 is the true `a` (captured in the reference
 * through c1_save, ckks_sym.c:227), not the aliased ntt(m+e) buffer.  Optional outputs may be
 * NULL: pte[n] = plaintext + error as int64; ntt_pte[nprimes][n] = NTT(m+e mod q_j).
 * end_ctr (optional) receives the final counter of the shareable (sym) / single (asym) PRNG.
 * Return 1 on success, 0 on encode overflow.
 */
int seo_encrypt_sym(const seo_params *p, const uint16_t *map, const float *values,
                    size_t values_len, const uint8_t share_seed[64], const uint8_t seed[64],
                    const uint8_t *sk_packed, uint32_t *c0, uint32_t *c1, int64_t *pte,
                    uint32_t *ntt_pte, uint64_t *end_ctr);

int seo_encrypt_asym(const seo_params *p, const uint16_t *map, const float *values,
                     size_t values_len, const uint8_t seed[64], const uint32_t *pk0,
                     const uint32_t *pk1, uint32_t *c0, uint32_t *c1, int64_t *pte,
                     uint8_t *u_packed, int8_t *e1, uint64_t *end_ctr);

/* gen_pk equivalent (ckks_asym.c:159-171 driven as in device/test/ckks_tests_asym.c:174-208):
 * ep = n CBD samples from PRNG(ep_seed, ctr 0..); per prime the shareable PRNG is re-seeded with
 * pk_seed at counter 0; pk1_j = a_j, pk0_j = -(a_j . NTT(s)) + NTT(ep mod q_j). */
void seo_gen_pk(const seo_params *p, const uint8_t *sk_packed, const uint8_t pk_seed[64],
                const uint8_t ep_seed[64], uint32_t *pk0, uint32_t *pk1);

/* ---- verification side: what the reference's round-trip tests use ---- */
/* intt_inpl (intt.c:26-58,144-222 + the inv_n / last_inv_sn constants of :230-420, which are
 * n^-1 and (s_last * n)^-1 mod q): exact inverse of seo_ntt_inpl. */
void seo_intt_inpl(const seo_params *p, size_t j, uint32_t *vec);
/* fft_inpl (fft.c:146-213): rounds h = 1,2,..,n/2; (u, v) -> (u + v*s, u - v*s), s = e^{2 pi i bitrev(h+j)/2n} */
void seo_fft_inpl(double *x_re_im, size_t n, size_t logn);
/* ckks_decrypt (device/test/ckks_tests_common.c:136-153): out = c0 + c1 . ntt_s mod q_j */
void seo_decrypt(const seo_params *p, size_t j, const uint32_t *c0, const uint32_t *c1,
                 const uint32_t *ntt_s, uint32_t *out);
/* ckks_decode (device/test/ckks_tests_common.c:59-115): centred lift, / scale, fft, slot pick */
void seo_decode(const seo_params *p, size_t j, const uint16_t *map, const uint32_t *pt,
                size_t values_len, float *out);

/* Batched drivers used for the timed CPU baseline (one thread each; caller shards). */
int seo_encrypt_sym_batch(const seo_params *p, const float *values /*[B][n/2]*/, size_t B,
                          const uint8_t *share_seeds /*[B][64]*/, const uint8_t *seeds /*[B][64]*/,
                          const uint8_t *sk_packed, uint32_t *c0 /*[B][np][n] or NULL*/,
                          uint32_t *c1 /*[B][np][n] or NULL*/, int nthreads);

int seo_encrypt_asym_batch(const seo_params *p, const float *values /*[B][n/2]*/, size_t B,
                           const uint8_t *seeds /*[B][64]*/, const uint32_t *pk0 /*[np][n]*/,
                           const uint32_t *pk1, uint32_t *c0 /*[B][np][n] or NULL*/,
                           uint32_t *c1 /*[B][np][n] or NULL*/, int nthreads);
/* BASELINE config 5: out[b][j] = NTT_j(reduce_set_pte(encode(values[b]))) ([B][np][n] or NULL) */
int seo_encode_ntt_batch(const seo_params *p, const float *values /*[B][n/2]*/, size_t B,
                         uint32_t *out, int nthreads);

uint64_t seo_fnv1a64(const void *data, size_t len, uint64_t h);

#ifdef __cplusplus
}
#endif
#endif
