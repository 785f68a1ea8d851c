/*
 * seal_embedded_amd.h -- C ABI of the MI355X-native CKKS encode+encrypt library
 *                        (libseal_embedded_amd.so).
 *
 * Two layers, both plain C (pointers and sizes only, no torch / C++ types):
 *
 *  1. The reference's own API surface for this path, same names / argument meaning / error
 *     behaviour, so existing SEAL-Embedded callers link unchanged:
 *        se_setup_custom, se_setup, se_setup_default, se_encrypt_seeded, se_encrypt, se_cleanup
 *        (replaces /root/reference/device/lib/seal_embedded.h:91-130, seal_embedded.c:24-235)
 *     Each call runs a batch of ONE through the GPU kernels.
 *
 *  2. Batched entry points (new): whole batches of independent plaintexts, host-pointer and
 *     device-pointer flavours, plus stage-level batched operators that mirror the lower surface
 *     the reference's tests/bench call directly (ckks_encode_base, ntt_inpl, prng_fill_buffer,
 *     sample_poly_uniform, sample_small_poly_ternary_prng_96, sample_poly_cbd_generic_prng_16).
 *
 * Ciphertext layout everywhere: uint32 [ct][prime][coeff] -- per ciphertext the polynomials of
 * prime 0..np-1, each n little-endian uint32 residues in [0,q_j), NTT form, bit-reversed order
 * (ntt.h:19-24).  c0 and c1 are separate slabs of that shape; record b of c0 followed by record b
 * of c1 per prime is exactly the byte stream the reference hands to its SEND_FNCT_PTR
 * (seal_embedded.c:196-203).
 *
 * All functions return SE_SUCCESS (0) or a negative SE_ERR_* code unless stated otherwise.
 * The library needs a HIP device: there is no CPU fallback; without one se_amd_create fails.
 */
#ifndef SEAL_EMBEDDED_AMD_H
#define SEAL_EMBEDDED_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

/* complex128 as the reference spells it (`double complex`, ckks_common.h:38); C++ translation
 * units see the same type through the GNU `_Complex` keyword */
#ifdef __cplusplus
typedef double _Complex se_complex;
#else
#include <complex.h>
typedef double complex se_complex;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: seal_embedded.h:35-40 --------------------------------------------------- */
#define SE_SUCCESS 0
#define SE_ERR_NO_MEMORY -12
#define SE_ERR_INVALD_ARGUMENT -22
#define SE_ERR_UNKNOWN -1000
#define SE_ERR_MINIMUM -9999
/* additions (inside the reserved range) */
#define SE_ERR_NO_DEVICE -19
#define SE_ERR_HIP -1001
#define SE_ERR_NO_KEY -1002

/* ---- reference types, same layout (defines.h:366-379, modulus.h:22-30, parameters.h:43-67,
 *      ckks_common.h:36-52, seal_embedded.h:52-65; default config: SE_USE_MALLOC, 32-bit ZZ) ---- */
typedef uint32_t ZZ;
typedef float flpt;
#define SE_PRNG_SEED_BYTE_COUNT 64 /* defines.h:67 */

typedef struct Modulus
{
    ZZ value;
    ZZ const_ratio[2]; /* floor(2^64/q): low word, high word */
} Modulus;

typedef struct
{
    size_t coeff_count;
    size_t logn;
    Modulus *moduli;
    Modulus *curr_modulus;
    size_t curr_modulus_idx;
    size_t nprimes;
    double scale;
    bool is_asymmetric;
    bool pk_from_file;
    bool sample_s;
    bool small_s;
    bool small_u;
} Parms;

typedef struct SE_PTRS
{
    se_complex *conj_vals;  /* n complex128; first 8n bytes hold the int64 plaintext after encode */
    se_complex *ifft_roots; /* 0, as in the reference's default SE_IFFT_OTF (ckks_sym.c:91) */
    flpt *values;           /* n/2 floats */
    ZZ *ternary;            /* 2-bit packed s (sym) / u of the last call (asym), n/4 bytes */
    int64_t *conj_vals_int_ptr;
    ZZ *c0_ptr; /* n residues of the current prime */
    ZZ *c1_ptr;
    uint16_t *index_map_ptr;
    ZZ *ntt_roots_ptr; /* one-shot NTT roots of the last prime processed (ntt.c:40-52) */
    ZZ *ntt_pte_ptr;
    int8_t *e1_ptr; /* e1 of the last asymmetric call */
} SE_PTRS;

typedef struct
{
    Parms *parms;
    SE_PTRS *se_ptrs;
} SE_PARMS;

typedef enum { SE_SYM_ENCR, SE_ASYM_ENCR } EncryptType;
typedef size_t (*SEND_FNCT_PTR)(void *, size_t);
typedef ssize_t (*RND_FNCT_PTR)(void *, size_t, unsigned int flags); /* seal_embedded.h:73 */

/* ---- layer 1: the reference API (seal_embedded.h:91-130) ----------------------------------
 * Behaviour kept: default parameter sets only (custom moduli are unreachable in the reference,
 * ckks_common.c:90); `scale` is overridden by the parameter set (parameters.c:190-226); one
 * parameter set per process (static singletons, seal_embedded.c:18-22); the secret key is read
 * from <SE_DATA_PATH>/sk_<n>.dat and the public key from pk{0,1}_ntt_<n>_<q>.dat relative to the
 * CWD (fileops.c:140-204; SE_DATA_PATH = "adapter_output_data", overridable through the
 * SE_AMD_DATA_PATH environment variable); NULL seeds draw from getrandom (rng.h:45-53); the
 * callback receives c0 then c1 of each prime, n*4 bytes each, from library-owned buffers.
 * Divergence (documented in DESIGN.md): in symmetric mode the reference's default build hands
 * the callback ntt(m+e) in place of c1 because the two buffers alias (ckks_sym.c:86-88).  We
 * deliver the correct c1 = a unless SE_AMD_REFERENCE_C1_ALIAS=1 is set in the environment, in
 * which case the reference's byte stream is reproduced bit for bit. */
SE_PARMS *se_setup_custom(size_t degree, size_t nprimes, const ZZ *modulus_vals, const ZZ *ratios,
                          double scale, EncryptType encrypt_type);
SE_PARMS *se_setup(size_t degree, size_t nprimes, double scale, EncryptType encrypt_type);
SE_PARMS *se_setup_default(EncryptType encrypt_type);
bool se_encrypt_seeded(uint8_t *shareable_seed, uint8_t *seed, SEND_FNCT_PTR network_send_function,
                       void *v, size_t vlen_bytes, bool print, SE_PARMS *se_parms);
bool se_encrypt(SEND_FNCT_PTR network_send_function, void *v, size_t vlen_bytes, bool print,
                SE_PARMS *se_parms);
void se_cleanup(SE_PARMS *se_parms);

/* New beside them: batched host-pointer entry on the handle se_setup returned.
 * values [B][n/2] float; share_seeds [B][64] (ignored for asymmetric; may be NULL then);
 * seeds [B][64]; c0, c1 [B][np][n] uint32 out.  Symmetric only: c1 may be NULL -- seed-compressed
 * form, the receiver re-expands c1 = a from share_seeds (se_amd_expand_c1_device), which halves
 * the bytes crossing PCIe.  With $SE_AMD_DEVICES = "all" or a comma list at se_setup time the
 * batch is sharded over those devices in contiguous blocks (one host thread per device).  Returns SE_SUCCESS, or the number (>0) of
 * plaintexts whose encoding overflowed int64 (their records are unspecified), or SE_ERR_*. */
int se_encrypt_batch(const SE_PARMS *se_parms, const float *values, size_t B,
                     const uint8_t *share_seeds, const uint8_t *seeds, uint32_t *c0, uint32_t *c1);

/* ---- layer 2: explicit context, batched, device pointers ---------------------------------- */
typedef struct se_amd_ctx se_amd_ctx;

/* Builds the parameter set (parameters.c:176-230), the index map (ckks_common.c:32-68), the IFFT
 * root table with the HOST libm (fft.c:39-45) and the per-prime NTT root tables (ntt.c:24-60)
 * and uploads them to HIP device `device`. */
int se_amd_create(se_amd_ctx **out, size_t degree, size_t nprimes, int device);
void se_amd_destroy(se_amd_ctx *ctx);

/* parameter read-back */
size_t se_amd_degree(const se_amd_ctx *ctx);
size_t se_amd_nprimes(const se_amd_ctx *ctx);
double se_amd_scale(const se_amd_ctx *ctx);
int se_amd_moduli(const se_amd_ctx *ctx, uint32_t *q /*[np]*/);
int se_amd_index_map(const se_amd_ctx *ctx, uint16_t *map /*[n], host*/);

/* keys (host pointers).  sk: 2-bit packed, n/4 bytes, sk_<n>.dat format.  pk: [np][n] uint32,
 * NTT form, the pk{0,1}_ntt_<n>_<q>.dat payloads concatenated over primes. */
int se_amd_set_secret_key(se_amd_ctx *ctx, const uint8_t *sk_packed);
int se_amd_set_public_key(se_amd_ctx *ctx, const uint32_t *pk0, const uint32_t *pk1);
int se_amd_load_keys_from_dir(se_amd_ctx *ctx, const char *dir, int want_pk);
/* gen_pk (ckks_asym.c:159-171 as driven by device/test/ckks_tests_asym.c:174-208) on the GPU: ep =
 * n CBD samples from PRNG(ep_seed); per prime the shareable PRNG restarts from pk_seed at counter 0;
 * pk1_j = a_j, pk0_j = -(a_j . NTT(s)) + NTT(ep mod q_j).  Also installs sk in the context.
 * Outputs [np][n] uint32 (host), the payloads of pk{0,1}_ntt_<n>_<q>.dat. */
int se_amd_gen_public_key(se_amd_ctx *ctx, const uint8_t *sk_packed, const uint8_t *pk_seed,
                          const uint8_t *ep_seed, uint32_t *pk0, uint32_t *pk1);

/* K key pairs in one launch chain (host pointers).  Secret key k: the sample branch
 * of ckks_setup_s (ckks_sym.c:162-179) = sample_small_poly_ternary_prng_96 from PRNG(sk_seeds[k]) at
 * counter 0, 2-bit packed (sk_<n>.dat format) -- or sk_in[k] when sk_in != NULL (sk_seeds may then be
 * NULL).  Public key k: gen_pk per prime exactly as se_amd_gen_public_key, from pk_seeds[k] / ep_seeds[k].
 * Seeds [K][64]; sk_in / sk_out [K][n/4] (sk_out may be NULL); pk0, pk1 [K][np][n] uint32 out.  The keys
 * installed in the context are not touched. */
int se_amd_gen_keys_batch(se_amd_ctx *ctx, size_t K, const uint8_t *sk_in, const uint8_t *sk_seeds,
                          const uint8_t *pk_seeds, const uint8_t *ep_seeds, uint8_t *sk_out, uint32_t *pk0,
                          uint32_t *pk1);

/* Whole path, device pointers, asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * Optional outputs may be NULL: ntt_pte [B][np][n] = NTT(m+e mod q_j); pte [B][n] int64;
 * status [B] bytes (1 ok / 0 encode overflow).  Internal scratch is grown on demand (not
 * stream-ordered: call once with the largest B before timing). */
int se_amd_encrypt_sym_device(se_amd_ctx *ctx, const float *d_values, size_t B,
                              const uint8_t *d_share_seeds, const uint8_t *d_seeds, uint32_t *d_c0,
                              uint32_t *d_c1, uint32_t *d_ntt_pte, int64_t *d_pte,
                              uint8_t *d_status, void *stream);
/* Seed-compressed symmetric ciphertext (the reference has only a stub, seal_embedded.c:184-194):
 * c1 = a is the expansion of the 64-byte shareable seed from counter 0 (sample.c:39-57 over the
 * prime chain), so only (share_seed, c0) need to travel; se_amd_expand_c1_device regenerates c1 on
 * the receiving side, bit-identical to what se_amd_encrypt_sym_device would have returned. */
int se_amd_encrypt_sym_seeded_device(se_amd_ctx *ctx, const float *d_values, size_t B,
                                     const uint8_t *d_share_seeds, const uint8_t *d_seeds,
                                     uint32_t *d_c0, uint8_t *d_status, void *stream);
int se_amd_expand_c1_device(se_amd_ctx *ctx, const uint8_t *d_share_seeds, size_t B, uint32_t *d_c1,
                            void *stream);
int se_amd_encrypt_asym_device(se_amd_ctx *ctx, const float *d_values, size_t B,
                               const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1,
                               uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status,
                               void *stream);
/* BASELINE config 5: encode + RNS reduce + NTT only; out [B][np][n] = NTT(m mod q_j). */
int se_amd_encode_ntt_device(se_amd_ctx *ctx, const float *d_values, size_t B, uint32_t *d_out,
                             int64_t *d_pte, uint8_t *d_status, void *stream);

/* ---- device-resident multi-GPU ----------------------------------------------
 * The path shards embarrassingly (independent plaintexts, own seeds, replicated keys / tables): a GROUP
 * holds one context per HIP device of one node; a batch of B units is cut into contiguous blocks
 * (se_amd_group_partition: the first B % ndev members take one more unit) and block i lives on member i's
 * device -- d_values[i], d_seeds[i], d_c0[i] ... are device pointers ON THAT DEVICE holding count[i]
 * records.  Every member runs the ordinary batched call on its block, driven by a host thread of its
 * own on the group's stream for that device; no collective on the data path.  The calls block until
 * every member has finished.
 *   gather_root < 0 : outputs stay resident on their devices.
 *   gather_root = r : additionally every member writes its finished block into its slice of member r's
 *                     slabs d_c0_all / d_c1_all ([B][np][n] on member r's device, record order = batch
 *                     order) by peer-to-peer copy on its own stream: on an 8-GPU node 7 concurrent
 *                     writers over 7 different xGMI links.  A member whose d_c0[i] already points at
 *                     its slice of the slab (typically the root) is not copied.  d_c1_all may be NULL:
 *                     only c0 is gathered -- with d_c1 == NULL as well this is the seed-compressed
 *                     symmetric form (the caller holds the 64-byte shareable seeds; se_amd_expand_c1_device
 *                     regenerates c1 where it is needed).
 * devices == NULL / ndev == 0: all visible devices.  The same ordinal may be listed more than once (two
 * contexts on one GPU) -- used by the single-GPU tests.  The reference has no counterpart: its boundary
 * is one ciphertext per call (seal_embedded.h:118-124). */
typedef struct se_amd_group se_amd_group;
int se_amd_group_create(se_amd_group **out, size_t degree, size_t nprimes, const int *devices, size_t ndev);
void se_amd_group_destroy(se_amd_group *g);
size_t se_amd_group_size(const se_amd_group *g);
se_amd_ctx *se_amd_group_ctx(se_amd_group *g, size_t i); /* member i's context (keys, stage-level calls) */
int se_amd_group_device(const se_amd_group *g, size_t i);
int se_amd_group_partition(const se_amd_group *g, size_t B, size_t *first /*[ndev]*/, size_t *count /*[ndev]*/);
int se_amd_group_set_secret_key(se_amd_group *g, const uint8_t *sk_packed);
int se_amd_group_set_public_key(se_amd_group *g, const uint32_t *pk0, const uint32_t *pk1);
int se_amd_group_reserve(se_amd_group *g, size_t B); /* scratch for batches of up to B units in total */
int se_amd_encrypt_sym_multi_device(se_amd_group *g, size_t B, const float *const *d_values,
                                    const uint8_t *const *d_share_seeds, const uint8_t *const *d_seeds,
                                    uint32_t *const *d_c0, uint32_t *const *d_c1 /* NULL: seed-compressed */,
                                    uint8_t *const *d_status /* NULL or [ndev] */, int gather_root,
                                    uint32_t *d_c0_all, uint32_t *d_c1_all);
int se_amd_encrypt_asym_multi_device(se_amd_group *g, size_t B, const float *const *d_values,
                                     const uint8_t *const *d_seeds, uint32_t *const *d_c0, uint32_t *const *d_c1,
                                     uint8_t *const *d_status, int gather_root, uint32_t *d_c0_all,
                                     uint32_t *d_c1_all);
int se_amd_encode_ntt_multi_device(se_amd_group *g, size_t B, const float *const *d_values, uint32_t *const *d_out,
                                   uint8_t *const *d_status, int gather_root, uint32_t *d_out_all);

/* Host-pointer entries: synchronous; a chunked pipeline (compute || D2H through a pinned staging
 * ring, or DMA straight into pinned/registered caller memory) that runs at the PCIe link rate.
 * c1 may be NULL for the symmetric form (seed-compressed: only c0 is returned). */
int se_amd_encrypt_sym_host(se_amd_ctx *ctx, const float *values, size_t B,
                            const uint8_t *share_seeds, const uint8_t *seeds, uint32_t *c0,
                            uint32_t *c1, uint32_t *ntt_pte, int64_t *pte, uint8_t *status);
int se_amd_encrypt_asym_host(se_amd_ctx *ctx, const float *values, size_t B, const uint8_t *seeds,
                             uint32_t *c0, uint32_t *c1, uint32_t *ntt_pte, int64_t *pte,
                             uint8_t *status);

/* ---- stage-level batched operators (device pointers) -------------------------------------- */
/* ckks_encode_base (ckks_common.c:105-215): values [B][n/2] -> int64 [B][n]; status [B]. */
int se_amd_encode_device(se_amd_ctx *ctx, const float *d_values, size_t B, int64_t *d_out,
                         uint8_t *d_status, void *stream);
/* ntt_inpl (ntt.c:168-189) for prime j on `count` polynomials [count][n], in place. */
int se_amd_ntt_device(se_amd_ctx *ctx, size_t prime, uint32_t *d_polys, size_t count, void *stream);
/* intt_inpl (intt.c:144-222, test-side in the reference) for prime j on `count` polynomials
 * [count][n], in place: NTT-form bit-reversed in, natural-order canonical out. */
int se_amd_intt_device(se_amd_ctx *ctx, size_t prime, uint32_t *d_polys, size_t count, void *stream);
/* The reference's round-trip check, batched (device/test/ckks_tests_common.c:59-231): for prime j
 * of every ciphertext d = c0 + c1 . NTT(s) (ckks_decrypt), pt = INTT(d), values = ckks_decode(pt).
 * Optional outputs (NULL to skip): d_dec_ntt [B][n], d_pt [B][n] uint32, d_values [B][n/2] float.
 * For a symmetric ciphertext d_dec_ntt equals NTT(m+e mod q_j) exactly. */
int se_amd_decrypt_decode_device(se_amd_ctx *ctx, const uint32_t *d_c0, const uint32_t *d_c1,
                                 size_t B, size_t prime, uint32_t *d_dec_ntt, uint32_t *d_pt,
                                 float *d_values, void *stream);
/* prng_fill_buffer (rng.h:78-91): out[i] = SHAKE256(seed[i] || le64(ctr[i]))[0:outlen]. */
int se_amd_prng_blocks_device(se_amd_ctx *ctx, const uint8_t *d_seeds, const uint64_t *d_ctrs,
                              uint8_t *d_out, size_t outlen, size_t count, void *stream);
/* sample_poly_uniform for primes 0..np-1 in chain order (sample.c:39-57): out [B][np][n];
 * d_ctr_in may be NULL (= 0); d_ctr_out optional [B]. */
int se_amd_sample_uniform_device(se_amd_ctx *ctx, const uint8_t *d_seeds, const uint64_t *d_ctr_in,
                                 size_t B, uint32_t *d_out, uint64_t *d_ctr_out, void *stream);
/* sample_small_poly_ternary_prng_96 (sample.c:218-242): one int8 code (0,1,2) per coefficient
 * [B][n] (the 2-bit packing of sample.c:61-87 is applied by se_amd_pack_ternary_host). */
int se_amd_sample_ternary_device(se_amd_ctx *ctx, const uint8_t *d_seeds, size_t B, int8_t *d_codes,
                                 uint64_t *d_ctr_out, void *stream);
/* sample_poly_cbd_generic_prng_16 (sample.c:311-321): int8 [B][blocks*16], block k of ciphertext
 * b uses counter ctr_base[b] + k (ctr_base NULL = 0). */
int se_amd_sample_cbd_device(se_amd_ctx *ctx, const uint8_t *d_seeds, const uint64_t *d_ctr_base,
                             size_t B, size_t blocks_per_ct, int8_t *d_out, void *stream);
void se_amd_pack_ternary_host(const int8_t *codes, size_t n, uint8_t *packed /*[n/4]*/);
/* Known-answer access to the device word arithmetic the kernels are built from (modulo.h:21-116,
 * uintmodarith.h:26-346 as restated in kernels/modarith.cuh), element-wise for prime j:
 * op 0 barrett 32->32 (a), 1 barrett 64->32 (a), 2 mul_mod(a, b) by 64-bit Barrett, 3 mul_mod(a, b) by
 * the Shoup form, 4 add_mod, 5 neg_mod(a), 6 sub_mod, 7 signed reduction of (int64)a
 * (ckks_common.c:224-237), 8 canonicalisation of a < 4q, 9/10 the two outputs of the Harvey NTT
 * butterfly on (a, b) with root c (ntt.c:156-162), 11/12 of the Gentleman-Sande butterfly
 * (intt.c:188-195).  Operands are uint64 arrays (b, c may be NULL where unused). */
int se_amd_word_ops_device(se_amd_ctx *ctx, size_t prime, int op, const uint64_t *d_a, const uint64_t *d_b,
                           const uint64_t *d_c, uint32_t *d_out, size_t count, void *stream);

/* ---- formats on either side of the path (host only) ---------------------------------------- */
/* SEAL Ciphertext data of size 2 (adapter/fileops.cpp:515-527): out[i + j*n] = c0 prime j,
 * out[i + j*n + np*n] = c1 prime j, one uint64 per coefficient; out has 2*np*n entries. */
void se_amd_pack_seal_ciphertext_host(const uint32_t *c0, const uint32_t *c1, size_t n, size_t np,
                                      uint64_t *out);
/* print_poly_full / print_poly_flpt_full text lines (util_print.h:229-245,491-508):
 * "name : { v0, v1, ... }\n".  Return the length needed (excluding the NUL); write at most cap. */
size_t se_amd_format_poly_text(const char *name, const uint32_t *poly, size_t n, char *buf, size_t cap);
size_t se_amd_format_values_text(const char *name, const float *v, size_t len, char *buf, size_t cap);
/* One ciphertext as the adapter's verify path reads it (api_tests.c:30-42,75-90): optional
 * "v (cleartext)" line, then "c0" and "c1" lines per prime. */
int se_amd_write_ciphertext_text(const char *path, int append, const float *values, size_t vlen,
                                 const uint32_t *c0, const uint32_t *c1, size_t n, size_t np);
/* key files in the device-side formats (fileops.c:140-204) */
int se_amd_save_secret_key_file(const char *dir, size_t n, const uint8_t *sk_packed);
int se_amd_save_public_key_files(const char *dir, size_t n, size_t np, const uint32_t *q,
                                 const uint32_t *pk0, const uint32_t *pk1);

/* ---- profiling hooks ---------------------------------------------------------------------- */
/* When enabled, every kernel launched by the whole-path entries is bracketed by HIP events on the
 * caller's stream; after a stream sync se_amd_stage_ms returns the accumulated milliseconds and
 * launch counts per stage since the last reset. */
enum
{
    SE_AMD_STAGE_CBD = 0,
    SE_AMD_STAGE_UNIFORM = 1,
    SE_AMD_STAGE_TERNARY = 2,
    SE_AMD_STAGE_ENCODE_ENCRYPT = 3, /* fused kernel (asymmetric, encode-only, unsplit symmetric) */
    SE_AMD_STAGE_ENCODE_RNS = 4,     /* split symmetric path: encode -> RNS residues */
    SE_AMD_STAGE_NTT_FUSE = 5,       /* split symmetric path: per-prime NTT + ciphertext arithmetic */
    SE_AMD_STAGE_COUNT = 6
};
int se_amd_set_profiling(se_amd_ctx *ctx, int enabled);
int se_amd_stage_ms(se_amd_ctx *ctx, float *ms /*[SE_AMD_STAGE_COUNT]*/,
                    uint64_t *launches /*[SE_AMD_STAGE_COUNT]*/, int reset);
/* test hook: capacity of the per-ciphertext rejection list of the uniform sampler (default 256);
 * tiny values force the overflow path. */
/* Host-only (no device needed): the setup-time tables the context uploads, for inspection and for
 * CPU-side checks.  Any output pointer may be NULL.  q[np]; const_ratio[np][2] = floor(2^64/q) as
 * {lo, hi} (modulus.c:30-47); index_map[n] (ckks_common.c:32-68); ifft_w[n][2] = (cos, -sin) of
 * 2*pi*bitrev(t)/2n from the host libm (fft.c:39-45); ntt_rw / intt_rw [np][n][2] = (root,
 * floor(root*2^32/q)) with root[bitrev(i)] = psi^i resp. psi^-i (ntt.c:40-52, intt.c:26-58).
 * Returns SE_SUCCESS or SE_ERR_INVALD_ARGUMENT for an unsupported (degree, nprimes). */
int se_amd_host_tables(size_t degree, size_t nprimes, uint32_t *q, uint32_t *const_ratio,
                       double *scale, uint16_t *index_map, double *ifft_w, uint32_t *ntt_rw,
                       uint32_t *intt_rw);
/* SHA-256 (64 hex digits + NUL) of the IFFT root table this context's kernels read, copied back from the device:
 * W[t] = (cos, -sin) of 2*pi*bitrev(t)/2n for t = 0 .. n-1 as little-endian doubles -- the digest SURVEY.md 8(c)
 * trap T8 lists per n (fft.c:39-45: the roots are the HOST libm's; goldens generated on another libm may differ
 * in rare last-bit cases).  Start-up check on the box that runs the kernels (__graft_entry__.smoke()). */
int se_amd_ifft_table_sha256(se_amd_ctx *ctx, char out_hex[65]);
int se_amd_set_reject_list_capacity(se_amd_ctx *ctx, uint32_t cap);
/* test hook: redraw candidates the helper waves precompute per ciphertext (default n/32); tiny
 * values force the pooled fallback for the remaining draws. */
int se_amd_set_speculation_capacity(se_amd_ctx *ctx, uint32_t cap);
/* test hook: ciphertexts per chunk of the host-pointer pipeline (0 = automatic: up to 16384, at most
 * 4 GiB of output per chunk). */
int se_amd_set_host_chunk(se_amd_ctx *ctx, size_t ciphertexts);
/* Pre-allocate the internal scratch for batches of up to B plaintexts (keeps hipMalloc out of
 * the first timed call). */
int se_amd_reserve(se_amd_ctx *ctx, size_t B);
/* Test / A/B hook: which FORM of the samplers and pipelines a call takes.  Every bit only selects among bit-identical
 * forms (the timing ablations of rounds 1-5 that produced WRONG outputs -- bits 1, 2, 4 -- are gone: no setting of this
 * word changes a result):
 *   8 no helper waves, 16 helper waves without speculation, 32 / 64 force the lane / wave form of the chain kernels,
 *   128 early encoder at n = 16384, 256 speculation windows of one guess (forces the miss path), 512 / 1024 force /
 *   forbid the pair-form staged sampler, 4096 ternary window of blocks + 2 counters (forces the fallback).
 * The environment overrides SE_AMD_STAGED, SE_AMD_SPECULATION are read at context creation into
 * members of their own and survive this call. */
int se_amd_set_debug_flags(se_amd_ctx *ctx, uint32_t flags);
/* pipeline shape of the symmetric path (A/B experiments): overlap = use the auxiliary stream,
 * split = 0 fused kernel, 1 per-prime software pipeline, 2 choose per call (default 1, 2). */
int se_amd_set_pipeline(se_amd_ctx *ctx, int overlap, int split);
/* pipeline shape of the public-key path (A/B experiments): chunks the batch is cut into so that the CBD
 * sampler of chunk k+1 runs beside the fused kernel of chunk k (default 1 = serial, or
 * $SE_AMD_ASYM_CHUNKS; batches below 4096 ciphertexts per chunk always run serially). */
int se_amd_set_asym_chunks(se_amd_ctx *ctx, size_t chunks);
const char *se_amd_last_error(void);
const char *se_amd_version(void);

#ifdef __cplusplus
}
#endif

/* ---- layer 1b: the reference's lower surface under its own names (ckks_encode_base, ckks_setup,
 *      ckks_sym_init, ckks_encode_encrypt_sym, gen_pk, ntt_inpl, ifft_inpl, ...) ---------------- */
#include "seal_embedded_amd_lower.h"

#endif /* SEAL_EMBEDDED_AMD_H */
