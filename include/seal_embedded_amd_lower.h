/*
 * seal_embedded_amd_lower.h -- the LOWER surface of SEAL-Embedded's device/lib under the
 * reference's own names and prototypes, served by the MI355X kernels of libseal_embedded_amd.so.
 *
 * This is the interface device/test/ckks_tests_sym.c:103-172, ckks_tests_asym.c:120-208 and
 * device/bench/bench_sym.c:76-143 drive directly: a caller written against the reference's
 * headers links against this library unchanged: include/seal_embedded.h and include/compat/ carry
 * one-line shim headers under the reference's file names (ckks_common.h, ckks_sym.h, ckks_asym.h,
 * ntt.h, intt.h, fft.h, parameters.h, modulus.h, rng.h, sample.h, fileops.h, defines.h; add
 * -Iinclude -Iinclude/compat).
 *
 * Every computational entry is a host-pointer batch-of-ONE call into the GPU kernels (copy in,
 * one or two launches, copy out; no CPU implementation of any operator exists in the library).
 * The first call for a (degree, nprimes) pair creates the GPU context ($SE_AMD_DEVICE, default 0);
 * without a HIP device the call prints the error and exits, which is the reference's own error
 * convention (ckks_sym.c:68-72, fileops.c:60-91).  The throughput interface is the batched layer
 * of seal_embedded_amd.h; this surface exists for drop-in compatibility and parity testing.
 *
 * Default configuration of the reference only (user_defines.h:37-117): IFFT/NTT root arguments
 * are accepted and ignored exactly where the reference's on-the-fly / one-shot options ignore or
 * overwrite them; parameter sets are the default ones (parameters.c:176-230).
 *
 * Included by seal_embedded_amd.h (which defines ZZ, flpt, Modulus, Parms, SE_PTRS).
 */
#ifndef SEAL_EMBEDDED_AMD_LOWER_H
#define SEAL_EMBEDDED_AMD_LOWER_H

#ifndef SEAL_EMBEDDED_AMD_H
#error "include seal_embedded_amd.h (or one of the reference-named shim headers) instead"
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t PolySizeType; /* defines.h:365 */
typedef int32_t ZZsign;      /* defines.h:374 */

/* ---- rng.h:26-114 ------------------------------------------------------------------------- */
typedef struct SE_PRNG
{
    uint8_t seed[SE_PRNG_SEED_BYTE_COUNT];
    uint64_t counter;
} SE_PRNG;
/* counter = 0; seed = seed_in, or getrandom() when NULL (rng.h:40-68) */
void prng_randomize_reset(SE_PRNG *prng, uint8_t *seed_in);
/* buffer = SHAKE256(seed || le64(counter))[0:byte_count]; counter++ (rng.h:78-91) -- on the GPU */
void prng_fill_buffer(size_t byte_count, SE_PRNG *prng, void *buffer);
/* Not in the reference: wipes and frees the device-side operand buffers the lower surface keeps between
 * calls (packed secret key, u, e / e1, PRNG seeds, m + e) and destroys its per-degree GPU contexts; also
 * runs at process exit.  The surface stays usable: state is rebuilt on the next call. */
void se_amd_lower_shutdown(void);
void prng_clear(SE_PRNG *prng);

/* ---- parameters.h:92-135, modulus.h:41-52 (host-side tables; A1) ---------------------------- */
void delete_parameters(Parms *parms);
void reset_primes(Parms *parms);
bool next_modulus(Parms *parms);
void set_parms_ckks(size_t degree, size_t nprimes, Parms *parms);
bool set_modulus(const ZZ q, Modulus *mod);
void set_modulus_custom(const ZZ q, ZZ hw, ZZ lw, Modulus *mod);

/* ---- ckks_common.h:62-169 ------------------------------------------------------------------ */
void ckks_calc_index_map(const Parms *parms, uint16_t *index_map);
void ckks_setup(size_t degree, size_t nprimes, uint16_t *index_map, Parms *parms);
/* NULL moduli/ratios -> ckks_setup.  Non-NULL: the reference recurses forever (ckks_common.c:90);
 * here: error message + exit(1). */
void ckks_setup_custom(size_t degree, size_t nprimes, const ZZ *modulus_vals, const ZZ *ratios,
                       uint16_t *index_map, Parms *parms);
void ckks_reset_primes(Parms *parms);
/* ckks_common.c:105-215.  values_len <= n/2; slots of conj_vals that the scatter does not reach
 * keep the caller's previous contents, as in the reference.  On success the first 8n bytes of
 * conj_vals hold the int64 plaintext.  `ifft_roots` is ignored (SE_IFFT_OTF, user_defines.h:68). */
bool ckks_encode_base(const Parms *parms, const flpt *values, size_t values_len, uint16_t *index_map,
                      se_complex *ifft_roots, se_complex *conj_vals);
void reduce_set_pte(const Parms *parms, const int64_t *conj_vals_int, ZZ *out);
void reduce_add_pte(const Parms *parms, const int64_t *conj_vals_int, ZZ *out);
void reduce_set_e_small(const Parms *parms, const int8_t *e, ZZ *out);
void reduce_add_e_small(const Parms *parms, const int8_t *e, ZZ *out);
/* the banner the reference prints before carving a pool (ckks_common.c:336-380) */
void print_ckks_mempool_size(size_t n, bool sym);

/* ---- fft.h:70-109 -------------------------------------------------------------------------- */
/* roots[i] = e^{+-2 pi i bitrev(i) / 2n} from the host libm (fft.c:47-67); the transforms below
 * ignore `roots` and use the same values (SE_IFFT_OTF / SE_FFT_OTF). */
void calc_fft_roots(size_t n, size_t logn, se_complex *roots);
void calc_ifft_roots(size_t n, size_t logn, se_complex *ifft_roots);
void ifft_inpl(se_complex *vec, size_t n, size_t logn, const se_complex *roots); /* fft.c:69-144 */
void fft_inpl(se_complex *vec, size_t n, size_t logn, const se_complex *roots);  /* fft.c:146-213 */

/* ---- ntt.h:40-54, intt.h:38-47 -------------------------------------------------------------- */
/* one-shot table roots[bitrev(i)] = psi^i mod q (ntt.c:40-52) resp. psi^-i (intt.c:26-58) for the
 * current prime of `parms`; host-side setup table (A10). */
void ntt_roots_initialize(const Parms *parms, ZZ *ntt_roots);
void intt_roots_initialize(const Parms *parms, ZZ *intt_roots);
/* in-place transforms mod parms->curr_modulus on the GPU; `*_roots` must be what the matching
 * *_roots_initialize produced (the device-resident copy of the same table is used). */
void ntt_inpl(const Parms *parms, const ZZ *ntt_roots, ZZ *vec);   /* ntt.c:168-189 */
void intt_inpl(const Parms *parms, const ZZ *intt_roots, ZZ *vec); /* intt.c:144-222 */

/* ---- sample.h:139-348 (the samplers of the default configuration) --------------------------- */
void sample_poly_uniform(const Parms *parms, SE_PRNG *prng, ZZ *poly);                  /* sample.c:39-57 */
void expand_poly_ternary(const ZZ *src, const Parms *parms, ZZ *dest);                  /* sample.c:113-129 */
void expand_poly_ternary_inpl(ZZ *poly, const Parms *parms);                            /* sample.c:131-136 */
/* accessors of the 2-bit packed form (sample.c:61-111): host-side format helpers */
void set_small_poly_idx(size_t idx, uint8_t val_in, ZZ *poly);
uint8_t get_small_poly_idx(const ZZ *poly, size_t idx);
ZZ get_small_poly_idx_expanded(const ZZ *poly, size_t idx, ZZ q);
/* expanded ternary polynomial of the previous prime -> current prime (sample.c:138-153) */
void convert_poly_ternary(const ZZ *src, const Parms *parms, ZZ *dest);
void convert_poly_ternary_inpl(ZZ *poly, const Parms *parms);
/* expanded (non-small) uniform ternary polynomial mod the current prime (sample.c:155-188) */
void sample_poly_ternary(const Parms *parms, SE_PRNG *prng, ZZ *poly);
void sample_small_poly_ternary_prng_96(PolySizeType n, SE_PRNG *prng, ZZ *poly);        /* sample.c:218-242 */
void sample_poly_cbd_generic_prng_16(PolySizeType n, SE_PRNG *prng, int8_t *poly);      /* sample.c:311-321 */
void sample_add_poly_cbd_generic_inpl_prng_16(int64_t *poly, PolySizeType n, SE_PRNG *prng); /* :347-356 */

/* ---- ckks_sym.h:41-123 --------------------------------------------------------------------- */
/* Pool carving of the reference's DEFAULT configuration (ckks_sym.c:78-160), in ZZ units:
 * conj_vals [0,4n) with c1 = ntt_pte at 2n and c0 at 3n (they live in the imaginary half that
 * ckks_encode_base no longer needs), ntt_roots at 4n, index_map at 5n, ternary (n/16) and values
 * (n/2) behind it.  c1_ptr == ntt_pte_ptr as in the reference. */
size_t ckks_get_mempool_size_sym(size_t degree);
ZZ *ckks_mempool_setup_sym(size_t degree);
void ckks_set_ptrs_sym(size_t degree, ZZ *mempool, SE_PTRS *se_ptrs);
/* sample_s: PRNG(seed) -> ternary s (2-bit packed); otherwise <SE_DATA_PATH>/sk_<n>.dat */
void ckks_setup_s(const Parms *parms, uint8_t *seed, SE_PRNG *prng, ZZ *s);
void ckks_sym_init(const Parms *parms, uint8_t *share_seed_in, uint8_t *seed_in, SE_PRNG *shareable_prng,
                   SE_PRNG *prng, int64_t *conj_vals_int);
/* One prime (parms->curr_modulus): c1 <- a from shareable_prng (counter advanced exactly as the
 * reference's rejection loop does); c0_s <- -(NTT(s) . a) + NTT(pte or ep_small); ntt_pte <-
 * NTT(pte); written in the reference's order, so aliased c1 / ntt_pte buffers end up holding
 * ntt_pte exactly as there (ckks_sym.c:86-88).  `ntt_roots` (scratch in the reference) receives
 * the one-shot root table of the prime. */
void ckks_encode_encrypt_sym(const Parms *parms, const int64_t *conj_vals_int, const int8_t *ep_small,
                             SE_PRNG *shareable_prng, ZZ *s_small, ZZ *ntt_pte, ZZ *ntt_roots, ZZ *c0_s,
                             ZZ *c1, ZZ *s_save, ZZ *c1_save);
bool ckks_next_prime_sym(Parms *parms, ZZ *s);

/* ---- ckks_asym.h:38-123 -------------------------------------------------------------------- */
size_t ckks_get_mempool_size_asym(size_t degree);
ZZ *ckks_mempool_setup_asym(size_t degree);
void ckks_set_ptrs_asym(size_t degree, ZZ *mempool, SE_PTRS *se_ptrs);
void gen_pk(const Parms *parms, ZZ *s_small, ZZ *ntt_roots, uint8_t *seed, SE_PRNG *shareable_prng,
            ZZ *s_save, int8_t *ep_small, ZZ *ntt_ep, ZZ *pk_c0, ZZ *pk_c1);
void ckks_asym_init(const Parms *parms, uint8_t *seed, SE_PRNG *prng, int64_t *conj_vals_int, ZZ *u,
                    int8_t *e1);
/* pk_c0 / pk_c1: in = public key of the current prime (read from pk{0,1}_ntt_<n>_<q>.dat when
 * parms->pk_from_file), out = ciphertext components. */
void ckks_encode_encrypt_asym(const Parms *parms, const int64_t *conj_vals_int, const ZZ *u,
                              const int8_t *e1, ZZ *ntt_roots, ZZ *ntt_u_e1_pte, ZZ *ntt_u_save,
                              ZZ *ntt_e1_save, ZZ *pk_c0, ZZ *pk_c1);
bool ckks_next_prime_asym(Parms *parms, ZZ *u);

/* ---- fileops.h (device-side key files, fileops.c:140-204) ----------------------------------- */
void load_sk(const Parms *parms, ZZ *s);
void load_pki(size_t i, const Parms *parms, ZZ *pki);

#ifdef __cplusplus
}
#endif
#endif /* SEAL_EMBEDDED_AMD_LOWER_H */
