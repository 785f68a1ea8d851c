// kernel_args.h -- argument blocks and launcher prototypes shared by the kernels and the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../se_types.h"

namespace seamd {

struct EncArgs
{
    const float *values;
    const int8_t *err;
    const int8_t *ucodes;
    uint32_t *c0;
    uint32_t *c1;
    uint32_t *ntt_pte;
    int64_t *pte;
    uint8_t *status;
    uint32_t *general;   // fused kernel only: [1 + B] count + indices of the plaintexts the fast form
                         // declined (not "small"), processed by k_encode_encrypt_general
    uint8_t *compact;    // split kernels only (optional): [B] k_encode_rns -> k_ntt_fuse, 1 = the plaintext
                         // was small and travels as ONE int32 row (in c0's last prime row) instead of np
                         // residue rows
    size_t count;        // fused kernel: plaintexts of the launch (set by the launcher: n = 4096 symmetric /
                         // encode-only workgroups take two plaintexts each)
};
struct UniformArgs
{
    const uint8_t *seeds;
    const uint64_t *ctr_in;
    uint64_t *ctr_out;
    uint32_t *out;
    uint32_t *rej_list;
    uint32_t rej_cap;
    uint32_t B;
    uint32_t prime_lo, prime_hi;
    uint32_t out_primes;
    uint32_t *spec;        // [B][spec_cap] scratch: candidates precomputed by helper waves
    uint32_t spec_cap;
    uint32_t master_waves; // waves of a workgroup that own ciphertexts (the rest are redraw helpers)
    uint32_t debug_flags;  // form selection for tests / A/B runs (results are bit-identical): 8 = no helper waves,
                           // 16 = helpers without speculation, 32 / 64 = always the lane / the wave form
    const uint32_t *only_from;  // optional [B]: ciphertext b takes part only if only_from[b] != 0 and
                                // prime_lo >= only_from[b] (redo of speculation misses)
    uint32_t out_prime_base;  // output row of prime j is (b * out_primes + j - out_prime_base): lets a
                              // single-prime launch (prime_lo = j) write one row per ciphertext
    uint32_t helper_fill;  // waves per workgroup that small batches are filled up to with helpers
                           // (0 = default 8 = two per SIMD; 4 leaves room for a co-resident
                           // 1024-thread transform workgroup at n = 16384)
    const uint8_t *prime_of;  // optional [B]: ciphertext b samples ONLY prime prime_of[b], into output row b
                              // (out_primes = 1; prime_lo / prime_hi / out_prime_base are ignored): the virtual
                              // ciphertexts of ALL guessed primes of the prime speculation in ONE launch
    uint32_t *nrej;           // staged form only: [B] rejected coefficients of the polynomial (k_bulk_pair ->
                              // k_resolve_wave)
    // staged forms: [1 + B] count + indices of the ciphertexts k_resolve_light could not finish (candidate row too
    // short, reject list overflowed); zeroed by the bulk kernel of the same prime, walked by a SMALL grid of
    // k_resolve_wave (a full grid of waves that read one word and leave cost 0.35-1.0 ms beside the throughput
    // kernels at B = 65 536).  NULL: k_resolve_wave is launched over every ciphertext and looks at the flag bit.
    uint32_t *flagged;
};
// Launches of at most this many ciphertexts take the wave-per-ciphertext kernel (k_sample_uniform_wave):
// 16 waves per CU = 4 per SIMD, where its ~7.8 us per permutation still beats the lane form's 8.6-10.7 us
// (tools/ubench5).
inline size_t uniform_wave_limit(unsigned num_cus) { return (size_t)16 * (num_cus ? num_cus : 256u); }
struct CbdArgs
{
    const uint8_t *seeds;
    const uint64_t *ctr_base;
    int8_t *out;
    uint32_t blocks_per_ct;
    uint32_t B;
};
struct TernaryArgs
{
    const uint8_t *seeds;
    int8_t *codes;
    uint64_t *ctr_out;
    uint32_t n;
    uint32_t B;
    const uint64_t *ctr_in;  // optional [B]: start counters (NULL = 0, the encrypt path)
    uint32_t num_cus;        // compute units of the device (0 = 256)
    uint32_t debug_flags;    // 32 = always the lane-per-ciphertext kernel (tests)
};

hipError_t launch_encode_encrypt(const DevParams &, const DevTables &, const EncArgs &, int mode,
                                 size_t B, hipStream_t);
hipError_t launch_encode_rns(const DevParams &, const DevTables &, const EncArgs &, bool add_err,
                             size_t B, hipStream_t);
hipError_t launch_ntt_fuse(const DevParams &, const DevTables &, const EncArgs &, int mode, int j,
                           size_t B, hipStream_t);
hipError_t launch_decrypt_decode(const DevParams &, const DevTables &, const uint32_t *c0,
                                 const uint32_t *c1, uint32_t in_primes, int j, uint32_t *dec_ntt,
                                 uint32_t *pt, float *values, size_t B, hipStream_t);
hipError_t launch_reduce_small(const DevParams &, const int8_t *e, uint32_t *out, size_t count, hipStream_t);
hipError_t launch_ntt_polys(const DevParams &, const DevTables &, int j, uint32_t *polys,
                            uint32_t *pairs, size_t count, hipStream_t);
hipError_t launch_make_pairs(const uint32_t *vals, uint32_t *pairs, uint32_t q, size_t count,
                             hipStream_t);
hipError_t launch_sample_uniform(const DevParams &, const UniformArgs &, hipStream_t);
// staged form, one prime per launch (kernels/samplers.hip: k_bulk_pair, k_candidates, k_resolve_wave)
hipError_t launch_uniform_bulk_pair(const DevParams &, const UniformArgs &, hipStream_t);
hipError_t launch_uniform_candidates(const UniformArgs &, hipStream_t);
hipError_t launch_uniform_resolve(const DevParams &, const UniformArgs &, hipStream_t);
// Small-batch prime speculation (se_context.cpp, encrypt_sym_small): the uniform sampler of prime
// j >= 1 is run for every plausible start counter of a window at once ("virtual ciphertexts"), so
// the primes of one ciphertext no longer wait for each other.
struct SpecPlan
{
    uint32_t nprimes;              // primes of the chain (speculated: 1 .. nprimes-1)
    uint32_t B;                    // real ciphertexts
    uint64_t base[kMaxPrimes];     // first guessed start counter of prime j
    uint32_t count[kMaxPrimes];    // guesses per ciphertext for prime j (0 for j = 0)
    uint32_t offset[kMaxPrimes];   // first virtual ciphertext of prime j
    uint32_t total;                // virtual ciphertexts in all
};
hipError_t launch_spec_setup(const SpecPlan &, const uint8_t *seeds, uint8_t *seeds_v, uint64_t *ctr_v,
                             uint8_t *prime_v, hipStream_t);
hipError_t launch_spec_select(const SpecPlan &, uint32_t n, uint64_t *ctr0, const uint64_t *ctrout_v,
                              const uint32_t *rows, uint32_t *c1, uint32_t *fail, hipStream_t);

// ---- explicit-operand stage kernels behind the reference-named lower surface (stage_ops.hip) ----
struct FftArgs
{
    const double *in;    // [count][n][2] interleaved complex128
    double *out_cplx;    // [count][n][2] (optional in mode 1)
    int64_t *out_int;    // mode 1: [count][n]
    uint32_t *fail_idx;  // mode 1: [count], preset to 0xFFFFFFFF
    int mode;            // 0 ifft_inpl, 1 ifft + round (ckks_encode_base tail), 2 fft_inpl
};
struct LowerSymArgs
{
    const uint8_t *s_small;  // 2-bit packed secret key(s): polynomial b uses s_small + b * s_stride bytes
    const int64_t *pte;      // [count][n] m + e, or NULL ...
    const int8_t *ep;        // ... then [count][n] small error (gen_pk)
    const uint32_t *a;       // uniform polynomial of this prime (NTT domain): a + b * a_stride
    uint32_t *c0, *ntt_pte;  // c0 + b * c0_stride; ntt_pte [count][n]
    uint32_t *s_save;        // optional [count][n]
    int j;
    uint32_t s_stride;       // bytes between the keys of consecutive polynomials (0 = one shared key)
    uint32_t a_stride;       // elements between consecutive polynomials of `a`  (0 = n)
    uint32_t c0_stride;      // elements between consecutive polynomials of `c0` (0 = n)
};
struct LowerAsymArgs
{
    const uint8_t *u_small;  // [count][n/4] 2-bit packed u
    const int8_t *e1;        // [count][n]
    const int64_t *pte;      // [count][n] m + e0
    const uint32_t *pk0, *pk1;  // [count][n] public key of this prime (in)
    uint32_t *c0, *c1, *ntt_pte;  // [count][n] (out)
    uint32_t *ntt_u_save, *ntt_e1_save;  // optional
    int j;
};
hipError_t launch_fft_polys(const DevParams &, const DevTables &, const FftArgs &, size_t count, hipStream_t);
hipError_t launch_reduce_poly(const DevParams &, int j, const int64_t *pte, const int8_t *e, uint32_t *out,
                              bool add, size_t total, hipStream_t);
hipError_t launch_word_ops(const DevParams &, int j, int op, const uint64_t *a, const uint64_t *b,
                           const uint64_t *c, uint32_t *out, size_t count, hipStream_t);
hipError_t launch_add_small(int64_t *m, const int8_t *e, size_t total, hipStream_t);
hipError_t launch_pack_ternary(const int8_t *codes, uint8_t *packed, size_t total_bytes, hipStream_t);
hipError_t launch_ternary_words(const uint32_t *in, uint32_t *out, uint32_t *nrej, uint32_t q, uint32_t n, int op,
                                hipStream_t);
hipError_t launch_expand_ternary(const uint8_t *packed, uint32_t *out, uint32_t q, uint32_t n, hipStream_t);
hipError_t launch_lower_sym_prime(const DevParams &, const DevTables &, const LowerSymArgs &, size_t count,
                                  hipStream_t);
hipError_t launch_lower_asym_prime(const DevParams &, const DevTables &, const LowerAsymArgs &, size_t count,
                                   hipStream_t);

hipError_t launch_sample_cbd(const CbdArgs &, hipStream_t);
hipError_t launch_sample_ternary(const TernaryArgs &, hipStream_t);
hipError_t launch_prng_blocks(const uint8_t *seeds, const uint64_t *ctrs, uint8_t *out,
                              uint32_t outlen, uint32_t count, hipStream_t);


}  // namespace seamd
