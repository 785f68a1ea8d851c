// encode_encrypt.hip -- one workgroup per plaintext: CKKS encode (FP64 inverse FFT) -> add the
// sampled error -> per RNS prime {signed reduction -> forward NTT -> fused ciphertext arithmetic}.
//
// Replaces, for a whole batch at once:
//   ckks_encode_base          /root/reference/device/lib/ckks_common.c:105-215
//   ifft_inpl                 /root/reference/device/lib/fft.c:69-144
//   reduce_set_pte / _e_small /root/reference/device/lib/ckks_common.c:224-265
//   ntt_inpl                  /root/reference/device/lib/ntt.c:124-189
//   the per-prime bodies of ckks_encode_encrypt_sym (ckks_sym.c:199-301) and
//   ckks_encode_encrypt_asym (ckks_asym.c:205-286) incl. poly_*_mod_inpl (polymodarith.h:39-101)
//
// Data never leaves the CU between encode and the final store: the plaintext stays in VGPRs
// (int32 x 16 per thread in the fast form, int64 in the general form -- see encrypt_one) across all
// primes; LDS is only the re-deal buffer of the transforms.
// HBM traffic per ciphertext = values (2n B) + error bytes + a (read back, 4n*np) + c0 (4n*np).
//
// NTT(s) is a per-key constant: it is computed once when the key is set (k_ntt_polys below)
// instead of once per ciphertext as the reference does -- same values, 1/2 of the NTT work.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../se_types.h"
#include "kernel_args.h"
#include "transform.cuh"

namespace seamd {

__device__ __forceinline__ void load16(uint32_t (&v)[16], const uint32_t *p)
{
    const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        uint4 w      = p4[i];
        v[4 * i]     = w.x;
        v[4 * i + 1] = w.y;
        v[4 * i + 2] = w.z;
        v[4 * i + 3] = w.w;
    }
}

__device__ __forceinline__ void store16(uint32_t *p, const uint32_t (&v)[16])
{
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) p4[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// 16 interleaved (value, shoup) pairs starting at pair index `first`
__device__ __forceinline__ void load16_pairs(uint32_t (&w)[16], uint32_t (&wp)[16],
                                             const uint32_t *tab, size_t first)
{
    const uint4 *p4 = reinterpret_cast<const uint4 *>(tab + 2 * first);
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
        uint4 v       = p4[i];
        w[2 * i]      = v.x;
        wp[2 * i]     = v.y;
        w[2 * i + 1]  = v.z;
        wp[2 * i + 1] = v.w;
    }
}

// quad-layout (transform.cuh, tile_to_quads) accessors: every instruction of a wave covers 1 KiB contiguous.
// One base pointer per access (the thread's first quad) and compile-time offsets that land in the instructions'
// immediate fields.
//
// opaque_index(): a thread index the compiler must treat as freshly computed where it is taken.  The fused
// kernels address the same per-thread pieces (a, key pairs, c0 / c1, u, e1) again for every prime; seen as
// loop-invariant, the 64-bit ADDRESS of every piece was formed ahead of the prime loop and carried -- or
// spilled -- across it (public-key form: 206 VGPRs, of which 44 were such addresses; the one-transform-at-a-
// time form spilled 22 of them).  Global addresses inside the prime loop are formed from an opaque copy of the
// thread index taken per iteration: a few 64-bit adds per prime, and the registers are free.
__device__ __forceinline__ int opaque_index(int t)
{
    __asm__ volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ void load_quads(uint32_t (&v)[16], const uint32_t *poly, int t)
{
    const uint32_t *base = poly + quad_index(t, 0);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const uint4 w = *reinterpret_cast<const uint4 *>(base + (i << 8));
        v[4 * i] = w.x, v[4 * i + 1] = w.y, v[4 * i + 2] = w.z, v[4 * i + 3] = w.w;
    }
}

__device__ __forceinline__ void store_quads(uint32_t *poly, const uint32_t (&v)[16], int t)
{
    uint32_t *base = poly + quad_index(t, 0);
#pragma unroll
    for (int i = 0; i < 4; i++)
        *reinterpret_cast<uint4 *>(base + (i << 8)) = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// interleaved (value, shoup) pairs of one polynomial's table, quad layout
__device__ __forceinline__ void load_quads_pairs(uint32_t (&w)[16], uint32_t (&wp)[16], const uint32_t *tab,
                                                 int t)
{
    const uint32_t *base = tab + 2 * (size_t)quad_index(t, 0);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const uint4 *p4 = reinterpret_cast<const uint4 *>(base + (i << 9));
        const uint4 a = p4[0], b = p4[1];
        w[4 * i] = a.x, wp[4 * i] = a.y, w[4 * i + 1] = a.z, wp[4 * i + 1] = a.w;
        w[4 * i + 2] = b.x, wp[4 * i + 2] = b.y, w[4 * i + 3] = b.z, wp[4 * i + 3] = b.w;
    }
}

// epilogue accessors in whichever layout the kernel's outputs are in: quad layout after tile_to_quads,
// else the thread's 16 consecutive coefficients
template <bool QUADS>
__device__ __forceinline__ void ld_poly(uint32_t (&v)[16], const uint32_t *poly, int t)
{
    if constexpr (QUADS) load_quads(v, poly, t); else load16(v, poly + 16 * t);
}
template <bool QUADS>
__device__ __forceinline__ void st_poly(uint32_t *poly, const uint32_t (&v)[16], int t)
{
    if constexpr (QUADS) store_quads(poly, v, t); else store16(poly + 16 * t, v);
}
template <bool QUADS>
__device__ __forceinline__ void ld_pairs(uint32_t (&w)[16], uint32_t (&wp)[16], const uint32_t *tab, int t)
{
    if constexpr (QUADS) load_quads_pairs(w, wp, tab, t); else load16_pairs(w, wp, tab, (size_t)16 * t);
}

// ------------------------------------------------------------------------------------------
// Encode front end shared by the fused and the split kernels: values -> LDS -> gather through the
// inverse index map -> inverse FFT -> round to int64 (+ overflow status).  On return thread t owns
// the plaintext coefficients k = t + (n/16)*e, e = 0..15, and the workgroup is synchronised.
// ------------------------------------------------------------------------------------------
//
// MT = int64_t: the general form.  MT = int32_t: the FAST form of the fused kernel -- the coefficients are
// only meaningful when `small` comes back true for every wave of the workgroup (|m| < 2 q_min - 64 < 2^31);
// the caller hands every other plaintext to the general kernel (k_encode_encrypt_general).
// The general kernel walks a list in a loop; seen as loop-invariant, the thread's table loads (twiddles,
// gather map, roots, key rows) would be hoisted out of that loop and the kernel spills.  Its thread index
// therefore passes through an opaque asm (a fresh value per iteration as far as the compiler knows).
template <bool OPAQUE>
__device__ __forceinline__ int thread_index()
{
    if constexpr (OPAQUE)
    {
        int t = threadIdx.x;
        __asm__ volatile("" : "+v"(t));
        return t;
    }
    else
        return threadIdx.x;
}

// C round() (half away from zero; ckks_common.c:183,192) as trunc(x + copysign(0.5 - 2^-54, x)): 3 operations
// (v_bfi, v_add_f64, v_trunc_f64) instead of the 6-7 of the library form (trunc, subtract, compare, select,
// copysign, add).  Exact for every double: with c = pred(0.5), a fraction below 0.5 can never be carried to the
// next integer (x + c stays below it by more than half an ulp), a fraction of 0.5 or more always is (x + c is
// within 2^-54 of the next integer, less than half a spacing -- and for x = 0.5 the tie 1 - 2^-54 rounds to even =
// 1.0); |x| >= 2^52 is an integer already and absorbs c; NaN / infinity pass through.  Checked exhaustively
// through the parity suite (every record of C2 / C3 / C4 / C5 against the oracle's round()) and on the host for
// the boundary cases (tests/test_oracle.py::test_round_half_away_form).
__device__ __forceinline__ double round_half_away(double x)
{
    const double c = 0.49999999999999994;   // 0.5 - 2^-54
    return trunc(__dadd_rn(x, dev_copysign(c, x)));
}

// Workgroup-wide outcome of the encoder (one OR-reduction over the workgroup, returned in `wg`):
constexpr int kWgOverflow  = 1;   // a coefficient fails the reference's overflow test (ckks_common.c:195)
constexpr int kWgNotSmall  = 2;   // some |m| >= 2 q_min - 64 (or a non-finite value in the fast form)
constexpr int kWgNonfinite = 4;   // int64 form: the plaintext holds a NaN or an infinite value
// BRANCH_EXACT (int64 form only): a plaintext with a NaN / infinite value takes the EXACT transform
// (transform.cuh, cmul_annexg), chosen by a workgroup-uniform branch -- the general kernels.  Without it the
// int64 form (k_encode_rns, whose register budget the second transform would cost 30 VGPRs) only REPORTS
// kWgNonfinite: its outputs are then meaningless, no status is written and the caller hands the plaintext
// to its general kernel.
template <int LOGN, typename MT, bool BRANCH_EXACT>
__device__ __forceinline__ void encode_plaintext(const DevParams &P, const DevTables &T,
                                                 const float *values, uint8_t *status, size_t b,
                                                 unsigned char *smem, MT (&m)[16], bool &small, int &wg)
{
    static_assert(sizeof(MT) == 8 || !BRANCH_EXACT, "the fast form never sees a non-finite plaintext through");
    const int t = thread_index<sizeof(MT) == 8>();
    using G          = XformGeom<LOGN>;
    constexpr int N  = G::N;
    constexpr int TH = G::THREADS;
    double *plane    = reinterpret_cast<double *>(smem);
    float *sv        = reinterpret_cast<float *>(smem);

    // ckks_common.c:139-153 scatters values[i] to both conjugate slots; the map is a bijection
    // onto [0,n), so slot k is filled from values[inv_map[k] mod n/2].  The staging array is laid out
    // through sv_slot() (se_types.h) so that the gather is bank-conflict-free; gather_map holds the
    // LDS position directly.
    //
    // NaN and infinite values are legal inputs of the reference (a NaN coefficient passes its overflow
    // test and is stored as INT64_MIN, ckks_common.c:195-206; an infinite one is the `return false`), and
    // they are the ONLY way a non-finite number enters the transform (|value| <= FLT_MAX keeps every
    // intermediate below 2^143).  Each thread folds the values it stages into v * 0 + acc, which is NaN
    // exactly when one of them is NaN or infinite (four v_pk_fma_f32): the fast form declines such a
    // plaintext (`small` comes back false), the general form takes the EXACT transform (transform.cuh,
    // cmul_annexg) for it.
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f nfacc = {0.0f, 0.0f};
    {
        const float4 *src = reinterpret_cast<const float4 *>(values + b * (N / 2));
        static_assert((N / 8) % TH == 0, "every thread stages the same number of float4 pieces");
#pragma unroll
        for (int k = 0; k < (N / 8) / TH; k++)
        {
            const int i    = t + k * TH;
            const float4 v = src[i];
            nfacc          = __builtin_elementwise_fma(v2f{v.x, v.y}, v2f{0.0f, 0.0f}, nfacc);
            nfacc          = __builtin_elementwise_fma(v2f{v.z, v.w}, v2f{0.0f, 0.0f}, nfacc);
            sv[sv_slot(4u * i, LOGN)]      = v.x;
            sv[sv_slot(4u * i + 1u, LOGN)] = v.y;
            sv[sv_slot(4u * i + 2u, LOGN)] = v.z;
            sv[sv_slot(4u * i + 3u, LOGN)] = v.w;
        }
    }
    const float nfsum    = nfacc.x + nfacc.y;
    const bool nonfinite = nfsum != nfsum;   // this thread staged a NaN or an infinity
    bool wg_nonfinite    = false;            // workgroup-uniform
    if constexpr (BRANCH_EXACT)
        wg_nonfinite = __syncthreads_or(nonfinite) != 0;
    else
        __syncthreads();
    double re[16], im[16];
    {
        // [2][n/16] uint4: half h of thread t's 16 entries at row h -- a wave load covers 1 KiB contiguous
        const uint4 *mp = reinterpret_cast<const uint4 *>(T.gather_map);
        uint4 m0 = mp[t], m1 = mp[TH + t];
        uint32_t packed[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            uint32_t idx = (packed[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
            re[e]        = (double)sv[idx];
            im[e]        = 0.0;
        }
    }
    __syncthreads();

    // inverse FFT (no 1/n: folded into n_inv, ckks_common.c:183); real input: short butterflies in pass 0
    if constexpr (BRANCH_EXACT)
    {
        if (wg_nonfinite)
            ifft_tiles<LOGN, false, true>(re, im, T.ifft_w, plane, t);
        else
            ifft_tiles<LOGN, true>(re, im, T.ifft_w, plane, t);
    }
    else
        ifft_tiles<LOGN, true>(re, im, T.ifft_w, plane, t);

    // round to int64, overflow check (ckks_common.c:183-206).  The largest magnitude of the thread
    // serves both the overflow test and the wave-uniform "small" flag: when every coefficient of
    // the wave stays below 2 q_min - 64 (< 2^31; the normal case, |m| ~ scale * |value|; the margin
    // covers the error term added later, |e| <= 21) the int64 conversion is one v_cvt_i32_f64 plus a
    // sign extension, and the per-prime "reduction" is the single add m + 2q (modarith.cuh).
    double amax = 0.0;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        re[e] = round_half_away(__dmul_rn(re[e], P.n_inv));
        amax  = fmax(amax, fabs(re[e]));
    }
    // fmax() skips NaNs: a NaN coefficient is not an overflow for the reference (ckks_common.c:195)
    const int ok = !(amax > 9223372036854775808.0);
    if constexpr (sizeof(MT) == 4)
    {
        small = __all(amax < P.small_bound && !nonfinite);
#pragma unroll
        for (int e = 0; e < 16; e++) m[e] = (int32_t)re[e];
    }
    else
    {
        small = __all(amax < P.small_bound && !nonfinite) && !wg_nonfinite;
        if (small)
        {
#pragma unroll
            for (int e = 0; e < 16; e++) m[e] = (int64_t)(int32_t)re[e];
        }
        else
        {
            // the reference's x86-64 build converts with cvttsd2si: NaN and +2^63 (which passes its check: it
            // rejects only > 2^63) come out as the "integer indefinite" INT64_MIN; the device conversion
            // saturates and maps NaN to 0
#pragma unroll
            for (int e = 0; e < 16; e++)
                m[e] = (fabs(re[e]) < 9223372036854775808.0) ? (int64_t)re[e] : INT64_MIN;
        }
    }
    // ONE reduction over the workgroup carries everything the callers need
    wg = __ockl_wgred_or_i32((ok ? 0 : kWgOverflow) | (small ? 0 : kWgNotSmall) | (nonfinite ? kWgNonfinite : 0));
    if constexpr (sizeof(MT) == 4)
    {
        // fast form: `small` for the whole workgroup (it implies "no overflow"); a plaintext that is not
        // small gets its status from the general kernel
        small = !(wg & kWgNotSmall);
        if (status && t == 0 && small) status[b] = 1;
    }
    else
    {
        const bool declined = !BRANCH_EXACT && (wg & kWgNonfinite);
        if (status && t == 0 && !declined) status[b] = (wg & kWgOverflow) ? 0 : 1;
    }
}

// ------------------------------------------------------------------------------------------
// The fused kernel comes in two forms.
//   FAST (GENERAL = false): the plaintext is carried as int32 -- every |m + e| of a normal plaintext is
//     below 2 q_min (|m| ~ scale * |value| ~ 2^30) -- which frees the 16 VGPRs of the high words and the
//     code of the 64-bit reduction: 164 -> 126 VGPRs (n = 4096 symmetric: 4 workgroups per CU instead of
//     3), the public-key form fits the quad-layout epilogue without spills.  A workgroup whose plaintext
//     is NOT small (any wave) appends its index to A.general and leaves without writing outputs.
//   GENERAL: int64 plaintext, the exact signed 64-bit reduction (reduce_pte_core, ckks_common.c:224-237);
//     k_encode_encrypt_general walks the list the fast launch produced (normally empty: the launch costs
//     a few microseconds).
// ------------------------------------------------------------------------------------------
template <int MODE>
constexpr int enc_quad_stride()
{
    return MODE == kModeAsym ? 28 : 20;
}

// pair form of the fast kernels (encrypt_pair below)
constexpr int kWgRisky          = 8;   // a coefficient within the guard band of a half-integer
constexpr double kHalfDelta     = 3.0e-14 * 1.001;
constexpr size_t kPairParkWords = 5120;   // parked coefficients start here (32-bit words into the dynamic LDS)

// Everything behind the encoder for ONE plaintext whose coefficients k = t + (n/16) e sit in m[]: + error, per prime
// {representative, NTT, fused ciphertext arithmetic}.  PAIR: the caller transforms two plaintexts per workgroup
// (encrypt_pair) and parks the second one's coefficients in LDS behind the first 9 216 words -- the transpose region
// then lives inside the NTT plane (one workgroup barrier more per prime: measured neutral, profiles/r05_ab_qalias28.log).
template <int LOGN, int MODE, bool GENERAL, bool PAIR, typename MT>
__device__ __forceinline__ void encrypt_tail(const DevParams &P, const DevTables &T, const EncArgs &A,
                                             const size_t b, unsigned char *smem, MT (&m)[16], const bool small)
{
    const int t = thread_index<GENERAL>();
    using G            = XformGeom<LOGN>;
    constexpr bool ASYM3 = LOGN <= 12;  // three-way NTT per prime (three LDS planes; spills at n = 8192)
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(smem);

    const int np   = P.nprimes;

    // thread t owns points k = t + (n/16)*e
    if constexpr (MODE == kModeSym)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) m[e] += A.err[b * N + (e << CTOP) + t];
    }
    else if constexpr (MODE == kModeAsym)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) m[e] += A.err[b * 2 * N + (e << CTOP) + t];
    }
    if (A.pte)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) A.pte[b * N + (e << CTOP) + t] = m[e];
    }

    // ---- 4. per prime ----------------------------------------------------------------------
    if constexpr (MODE == kModeEncodeOnly)
    {
        if (!A.c0) return;  // plain ckks_encode_base: only the int64 plaintext was requested
    }
    // Outputs leave the kernel in quad layout (n <= 4096): the wave-local transpose region sits behind the
    // NTT plane(s), so it needs no barrier against the next prime's exchanges.  Symmetric / encode-only:
    // rows of 20 words (32 conflict cycles per transpose instead of the 16 of 28-word rows,
    // tools/lds_conflicts.py) keep plane + region at 37 KiB -- 4 workgroups per CU.  Public-key form: rows
    // of 28 INSIDE the three planes (QALIAS below: 51 KiB, 3 workgroups per CU at 150 VGPRs); its general form
    // keeps the tile layout (three transposes beside the int64 plaintext spill).
    constexpr bool QUADS   = LOGN <= 12 && (MODE != kModeAsym || !GENERAL);
    constexpr int QSTRIDE  = enc_quad_stride<MODE>();
    // The public-key form's transpose region lives INSIDE its three planes (free after the last exchange's
    // trailing barrier; one workgroup barrier per prime keeps the next prime's first exchange off it): 51 KiB per
    // workgroup instead of 79, THREE workgroups per CU -- possible since the kernel needs 150 VGPRs (opaque_index
    // above; 206 before).  Fused stage 5.09 -> 4.83 ms per 65 536 (profiles/r04_ab_transform.log).  (The same alias
    // for the symmetric / encode-only forms, 28-word rows: no gain, profiles/r05_ab_qalias28.log.)
    constexpr bool QALIAS  = (MODE == kModeAsym && ASYM3 && !GENERAL) || PAIR;
    static_assert(!PAIR || (size_t)(G::N / 16) * QSTRIDE <= kPairParkWords, "the transpose region ends below the parked plaintext");
    uint32_t *qlds         = lds32 + (QALIAS ? 0 : (MODE == kModeAsym && ASYM3 ? 3 : 1)) * G::SLOTS;
    auto to_quads = [&](uint32_t (&v)[16]) {
        if constexpr (QUADS) tile_to_quads<QSTRIDE>(v, qlds, t);
    };
    for (int j = 0; j < np; j++)
    {
        const uint32_t q = P.q[j], two_q = q << 1;
        const uint32_t crh = P.cr_hi[j], crl = P.cr_lo[j];
        const uint32_t *RW = T.ntt_rw + 2 * xform_table_len(N) * j;
        const size_t pb    = (b * np + j) * N;            // this polynomial in the [ct][prime][coeff] slabs
        const size_t kb    = (size_t)2 * N * j;           // this prime's rows of the (value, shoup) key tables
        // thread index for this prime's GLOBAL addresses (opaque_index above; the encode-only form has one store
        // per prime and nothing to hoist: measured 1 % slower with it)
        // (the transforms keep the plain index: with the opaque one the symmetric form needs 100 VGPRs instead of
        // 120 but recomputes its LDS and root addresses per prime and runs 12 % slower, profiles/r04_ab_transform.log)
        const int tg       = MODE == kModeEncodeOnly ? t : opaque_index(t);
        uint32_t x[16];

        if constexpr (MODE == kModeAsym && ASYM3)
        {
            // the three transforms of this prime side by side (ntt_tiles3): u_hat = NTT(expand(u))
            // (ckks_asym.c:235-241; code 0 -> q-1, 1 -> 0, 2 -> 1), NTT(e1) (:263-272) and
            // NTT(m + e0) (:280-284)
            uint32_t uh[16], y[16];
            const int8_t *up = A.ucodes + b * N + tg, *ep = A.err + b * 2 * N + N + tg;
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                uint32_t code = (uint32_t)up[e << CTOP];
                uh[e]         = code + q - 1u;            // q-1, q, q+1 == -1, 0, 1 (sample.c:98-111 mod q)
                int32_t e1    = ep[e << CTOP];
                y[e]          = q + (uint32_t)e1;          // == reduce_set_e_small (ckks_common.c:259-265) mod q
            }
            reduce_signed16(m, x, q, crh, crl, small);
            ntt_tiles3<LOGN>(uh, y, x, RW, q, lds32, t);
            to_quads(uh);
            {
                // c1 = pk1 . u_hat + NTT(e1)   (:251): the lazy transform outputs go through the transpose as they
                // are; one canonicalisation at the end (modarith.cuh, add_mul_canon)
                uint32_t w[16], wp[16];
                to_quads(y);
                ld_pairs<QUADS>(w, wp, T.pk1 + kb, tg);
#pragma unroll
                for (int e = 0; e < 16; e++) y[e] = add_mul_canon(y[e], uh[e], w[e], wp[e], q, two_q);
                st_poly<QUADS>(A.c1 + pb, y, tg);
            }
            to_quads(x);
            if (A.ntt_pte)
            {
                uint32_t cx[16];
#pragma unroll
                for (int e = 0; e < 16; e++) cx[e] = canon4(x[e], q, two_q);
                st_poly<QUADS>(A.ntt_pte + pb, cx, tg);
            }
            {
                // c0 = pk0 . u_hat + NTT(m + e0)   (:255)
                uint32_t w[16], wp[16], out[16];
                ld_pairs<QUADS>(w, wp, T.pk0 + kb, tg);
#pragma unroll
                for (int e = 0; e < 16; e++) out[e] = add_mul_canon(x[e], uh[e], w[e], wp[e], q, two_q);
                st_poly<QUADS>(A.c0 + pb, out, tg);
            }
            if constexpr (QALIAS) __syncthreads();
        }
        else if constexpr (MODE == kModeAsym)
        {
            // one transform at a time (n >= 8192)
            // u_hat = NTT(expand(u))   (ckks_asym.c:235-241; code 0 -> q-1, 1 -> 0, 2 -> 1)
            uint32_t uh[16];
            const int8_t *up = A.ucodes + b * N + tg;
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                uint32_t code = (uint32_t)up[e << CTOP];
                uh[e]         = code + q - 1u;
            }
            ntt_tiles<LOGN>(uh, RW, q, lds32, t);
            to_quads(uh);
            // n = 16384 (1 024 threads: 128 VGPRs): u_hat waits in LDS while the other two transforms run -- the
            // encoder's FP64 plane is twice the NTT plane, so the upper half of the allocation is free after the
            // encode; thread-private slots ([e][t]: conflict-free, no barrier).  76 B of scratch per lane without it, 28 B with (three
            // root-table addresses the compiler still carries across the prime loop).
            constexpr bool PARK = LOGN == 14 && !GENERAL;
            uint32_t *park = lds32 + G::SLOTS + t;
            static_assert(!PARK || (size_t)G::SLOTS * sizeof(double) >= ((size_t)G::SLOTS + 16 * G::THREADS) * sizeof(uint32_t),
                          "the parked polynomial fits behind the NTT plane inside the encoder's plane");
            auto park_store = [&]() {
                if constexpr (PARK)
                {
#pragma unroll
                    for (int e = 0; e < 16; e++) park[e * G::THREADS] = uh[e];
                }
            };
            auto park_load = [&]() {
                if constexpr (PARK)
                {
#pragma unroll
                    for (int e = 0; e < 16; e++) uh[e] = park[e * G::THREADS];
                }
            };
            park_store();
            // c1 = pk1 . u_hat + NTT(e1)   (:251, :263-272)
            const int8_t *ep = A.err + b * 2 * N + N + tg;
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                int32_t e1 = ep[e << CTOP];
                x[e]       = q + (uint32_t)e1;
            }
            ntt_tiles<LOGN>(x, RW, q, lds32, t);
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = canon4(x[e], q, two_q);
            to_quads(x);
            park_load();
            {
                uint32_t w[16], wp[16], out[16];
                ld_pairs<QUADS>(w, wp, T.pk1 + kb, tg);
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    uint32_t pr = csub(mul_shoup_lazy(uh[e], w[e], wp[e], q), q);
                    out[e]      = csub(pr + x[e], q);
                }
                st_poly<QUADS>(A.c1 + pb, out, tg);
            }
            // c0 = pk0 . u_hat + NTT(m + e0)   (:255, :280-284)
            reduce_signed16(m, x, q, crh, crl, small);
            ntt_tiles<LOGN>(x, RW, q, lds32, t);
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = canon4(x[e], q, two_q);
            to_quads(x);
            if (A.ntt_pte) st_poly<QUADS>(A.ntt_pte + pb, x, tg);
            park_load();
            {
                uint32_t w[16], wp[16], out[16];
                ld_pairs<QUADS>(w, wp, T.pk0 + kb, tg);
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    uint32_t pr = csub(mul_shoup_lazy(uh[e], w[e], wp[e], q), q);
                    out[e]      = csub(pr + x[e], q);
                }
                st_poly<QUADS>(A.c0 + pb, out, tg);
            }
        }
        else
        {
            // a_j comes from HBM: requested before the transform, it lands while the NTT runs (fused stage
            // 2.60 -> 2.38 ms per 65 536, profiles/r04_ab_transform.log; 16 VGPRs the NTT phase has to spare)
            uint32_t a_pre[16];
            if constexpr (MODE == kModeSym && QUADS) ld_poly<QUADS>(a_pre, A.c1 + pb, tg);
            // NTT(m + e mod q_j)   (ckks_sym.c:286-292)
            reduce_signed16(m, x, q, crh, crl, small);  // see modarith.cuh: the fused
                                                                          // symmetric kernel keeps the exact form
            ntt_tiles<LOGN>(x, RW, q, lds32, t);
            if constexpr (MODE != kModeSym)
            {
#pragma unroll
                for (int e = 0; e < 16; e++) x[e] = canon4(x[e], q, two_q);
            }
            to_quads(x);   // symmetric: the lazy values in [0,4q); sub_mul_canon canonicalises once at the end
            if (A.ntt_pte)
            {
                uint32_t cx[16];
#pragma unroll
                for (int e = 0; e < 16; e++) cx[e] = canon4(x[e], q, two_q);
                st_poly<QUADS>(A.ntt_pte + pb, cx, tg);
            }
            if constexpr (MODE == kModeSym)
            {
                // c0 = -(s_hat . a) + NTT(m+e)   (ckks_sym.c:273-300); a was written to c1
                uint32_t a[16], w[16], wp[16], out[16];
                if constexpr (QUADS)
                {
#pragma unroll
                    for (int e = 0; e < 16; e++) a[e] = a_pre[e];
                }
                else
                    ld_poly<QUADS>(a, A.c1 + pb, tg);
                ld_pairs<QUADS>(w, wp, T.s_hat + kb, tg);
#pragma unroll
                for (int e = 0; e < 16; e++) out[e] = sub_mul_canon(x[e], a[e], w[e], wp[e], q, two_q);
                st_poly<QUADS>(A.c0 + pb, out, tg);
            }
            else
            {
                st_poly<QUADS>(A.c0 + pb, x, tg);
            }
            if constexpr (QALIAS) __syncthreads();   // the transpose region is the next prime's exchange plane
        }
    }
}

template <int LOGN, int MODE, bool GENERAL>
__device__ __forceinline__ void encrypt_one(const DevParams &P, const DevTables &T, const EncArgs &A,
                                            const size_t b, unsigned char *smem)
{
    const int t = thread_index<GENERAL>();
    using MT = typename std::conditional<GENERAL, int64_t, int32_t>::type;
    MT m[16];
    bool small;  // wave-uniform: every |m + e| of this wave is below 2 q_min
    int wg;
    encode_plaintext<LOGN, MT, GENERAL>(P, T, A.values, A.status, b, smem, m, small, wg);
    if constexpr (!GENERAL)
    {
        if (!small)   // workgroup-uniform in the fast form (encode_plaintext)
        {
            if (t == 0) A.general[1 + atomicAdd(A.general, 1u)] = (uint32_t)b;
            return;
        }
    }
    encrypt_tail<LOGN, MODE, GENERAL, false>(P, T, A, b, smem, m, small);
}

// ------------------------------------------------------------------------------------------
// Round 6: the fast (int32) symmetric / encode-only forms at n = 4096 take TWO plaintexts per workgroup and encode
// them through the half-size transform (transform.cuh, ifft_pair_real_half), keeping the reference's bits by a guard:
//
//   * Through stage 10 the lower half's values are the reference's own, bit for bit.  The reference's last stage uses
//     v, its COMPUTED upper-half value, where we use conj(u); in exact arithmetic they are equal, so the two outputs
//     of a pair differ by at most |Re v - Re u| + |Im v + Im u| (+ 3 roundings) <= sqrt(2) |v - conj(u)|, and
//     |v - conj(u)| <= the forward errors of v and of u after 11 stages.  A butterfly row carries a relative error
//     <= 6u on (|u| + |v|) (sum: u; difference x root: u + 2 sqrt(2) u for the Annex-G product + 2u for libm's
//     root, u = 2^-53); a stage multiplies the 2-norm by sqrt(2) and the norm of the absolute-value matrix is 2, so
//     11 stages leave ||x^ - x||_2 <= 11 sqrt(2) 6u ||x||_2 = 1.04e-14 ||x||_2 with ||x||_2 = ||y||_2 / sqrt(2), y =
//     the exact output vector.  Every output therefore differs from the reference's by less than 2.2e-14 ||y||_2 (incl.
//     the roundings of the last stage and of the scaling); ||y||_2 = sqrt(n) ||A||_2 = sqrt(2 n sum values^2) exactly.
//   * kHalfDelta = 3e-14 (x 1.001 for the float accumulation of the sum) is that bound with 36 % of slack -- and the
//     bound itself assumes every rounding error aligned: the measured maximum deviation is 1.9e-17 ||m||_2, 1 300 times
//     smaller.  A coefficient whose scaled value lies within delta of a half-integer could round differently from the
//     reference's: the workgroup then REDOES that plaintext with the full transform (encode_plaintext; about one
//     plaintext in nine at the bench distribution).  Everything else about the fast form is unchanged: a plaintext
//     that is not small, or holds a non-finite value, is declined to the general kernel.
// The second plaintext's coefficients wait in LDS (thread-private slots behind the NTT plane + transpose region)
// while the first one's primes run.
// ------------------------------------------------------------------------------------------

// flags of plaintext A in bits 0..7, of plaintext B in bits 8..15 (kWgNotSmall | kWgNonfinite | kWgRisky)
template <int LOGN>
__device__ __forceinline__ int encode_pair_half(const DevParams &P, const DevTables &T, const float *values,
                                                const size_t bA, const size_t bB, unsigned char *smem,
                                                int32_t (&mA)[16], int32_t (&mB)[16])
{
    using G          = XformGeom<LOGN>;
    constexpr int N  = G::N;
    constexpr int TH = G::THREADS;
    static_assert((N / 8) / TH == 2, "two float4 pieces per thread and plaintext");
    const int t      = threadIdx.x;
    double *plane    = reinterpret_cast<double *>(smem);
    float *sv        = reinterpret_cast<float *>(smem);   // A's values at [0, n/2), B's at [n/2, n)
    float *part      = sv + N;                            // [wave][plaintext] sums of squares

    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f nf[2]   = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    float ss[2] = {0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < 2; p++)
    {
        const float4 *src = reinterpret_cast<const float4 *>(values + (p ? bB : bA) * (N / 2));
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const int i    = t + k * TH;
            const float4 v = src[i];
            nf[p]          = __builtin_elementwise_fma(v2f{v.x, v.y}, v2f{0.0f, 0.0f}, nf[p]);
            nf[p]          = __builtin_elementwise_fma(v2f{v.z, v.w}, v2f{0.0f, 0.0f}, nf[p]);
            ss[p]          = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss[p]))));
            float *dst     = sv + p * (N / 2);
            dst[sv_slot(4u * i, LOGN)]      = v.x;
            dst[sv_slot(4u * i + 1u, LOGN)] = v.y;
            dst[sv_slot(4u * i + 2u, LOGN)] = v.z;
            dst[sv_slot(4u * i + 3u, LOGN)] = v.w;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss[0] += __shfl_xor(ss[0], off), ss[1] += __shfl_xor(ss[1], off);
    if ((t & 63) == 0) part[2 * (t >> 6)] = ss[0], part[2 * (t >> 6) + 1] = ss[1];
    bool nonfinite[2];
#pragma unroll
    for (int p = 0; p < 2; p++)
    {
        const float s = nf[p].x + nf[p].y;
        nonfinite[p]  = s != s;   // this thread staged a NaN or an infinity of plaintext p
    }
    __syncthreads();
    const float ss_all[2] = {(part[0] + part[2]) + (part[4] + part[6]), (part[1] + part[3]) + (part[5] + part[7])};
    double re[16], im[16];
    {
        const int Tl    = t & 127;
        const float *my = sv + (t >> 7) * (N / 2);
        const uint4 *mp = reinterpret_cast<const uint4 *>(T.gather_map);
        uint4 m0 = mp[Tl], m1 = mp[TH + Tl];
        uint32_t packed[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            uint32_t idx = (packed[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
            re[e]        = (double)my[idx];
            im[e]        = 0.0;
        }
    }
    __syncthreads();
    ifft_pair_real_half<LOGN>(re, im, T.ifft_w, plane, t);

    int flags = 0;
#pragma unroll
    for (int p = 0; p < 2; p++)
    {
        // |fast - reference| < delta for every scaled coefficient (see above); ||m||_2 = n_inv sqrt(2 n sum values^2)
        const double delta = kHalfDelta * P.n_inv * sqrt(2.0 * (double)N * (double)ss_all[p]);
        double amax = 0.0;
        bool risky  = false;
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            const double x = e < 8 ? re[8 * p + e] : im[8 * p + e - 8];
            const double v = __dmul_rn(x, P.n_inv);
            const double a = fabs(v);
            const double g = __dsub_rn(a, trunc(a));        // exact; distance to the rounding boundary = |g - 0.5|
            risky          = risky || (fabs(__dsub_rn(g, 0.5)) < delta);
            const double r = round_half_away(v);
            amax           = fmax(amax, fabs(r));
            if (p == 0) mA[e] = (int32_t)r; else mB[e] = (int32_t)r;
        }
        const bool sm = amax < P.small_bound && !nonfinite[p];
        flags |= ((sm ? 0 : kWgNotSmall) | (nonfinite[p] ? kWgNonfinite : 0) | (risky ? kWgRisky : 0)) << (8 * p);
    }
    return __ockl_wgred_or_i32(flags);
}

template <int LOGN, int MODE>
__device__ __forceinline__ void encrypt_pair(const DevParams &P, const DevTables &T, const EncArgs &A,
                                             const size_t bA, const bool haveB, unsigned char *smem)
{
    const int t     = threadIdx.x;
    const size_t bB = haveB ? bA + 1 : bA;   // an odd batch ends with a workgroup that carries its plaintext twice
    int32_t m[16], mo[16];                   // A's coefficients / B's
    const int flags = encode_pair_half<LOGN>(P, T, A.values, bA, bB, smem, m, mo);
    // Per plaintext: declined (general kernel), or coefficients valid -- after an exact redo when one of them sat in the
    // guard band (two call sites of the full encoder: a loop over the two register arrays would put them in scratch).
    int run = 0;
    auto settle = [&](int32_t (&mm)[16], const int f, const size_t b, const bool live, const int bit) {
        bool small = !(f & (kWgNotSmall | kWgNonfinite));
        if (small && (f & kWgRisky) && live)
        {
            int wg;   // the full transform decides (it writes the status and may still decline)
            encode_plaintext<LOGN, int32_t, false>(P, T, A.values, A.status, b, smem, mm, small, wg);
        }
        else if (small && live && A.status && t == 0)
            A.status[b] = 1;
        if (!small && live && t == 0) A.general[1 + atomicAdd(A.general, 1u)] = (uint32_t)b;
        run |= (small && live) ? bit : 0;
    };
    settle(m, flags & 0xFF, bA, true, 1);
    settle(mo, (flags >> 8) & 0xFF, bB, haveB, 2);
    // B's coefficients wait in thread-private LDS slots behind the NTT plane and the transpose region while A's
    // primes run (ONE call site of the tail: the loop below is not unrolled)
    uint32_t *park = reinterpret_cast<uint32_t *>(smem) + kPairParkWords + t;
    __syncthreads();   // an exact redo may still be reading the plane the slots lie in
    if (run & 2)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) park[e * XformGeom<LOGN>::THREADS] = (uint32_t)mo[e];
    }
#pragma nounroll
    for (int p = 0; p < 2; p++)
    {
        if (run & (1 << p)) encrypt_tail<LOGN, MODE, false, true>(P, T, A, p ? bB : bA, smem, m, true);
        if (p == 0 && (run & 2))
        {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; e++) m[e] = (int32_t)park[e * XformGeom<LOGN>::THREADS];
        }
    }
}

// two plaintexts per workgroup (encrypt_pair)
template <int LOGN, int MODE>
constexpr bool enc_pairs()
{
    return LOGN == 12 && MODE != kModeAsym;
}

// workgroups per CU the register budget is set for (n <= 4096: 256 threads = one wave per SIMD each)
template <int LOGN, int MODE, bool GENERAL>
constexpr int enc_blocks()
{
    if (LOGN > 12) return 1;
    if (MODE == kModeAsym && !GENERAL) return 3;   // transpose region inside the planes (encrypt_one, QALIAS)
    if (MODE == kModeAsym) return 2;   // 3 (with the transpose region aliased) spills: 5.45 -> 7.07 ms
    return GENERAL ? 3 : 4;
}

template <int LOGN, int MODE>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS, (enc_blocks<LOGN, MODE, false>()))
void k_encode_encrypt(DevParams P, DevTables T, EncArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (enc_pairs<LOGN, MODE>())
        encrypt_pair<LOGN, MODE>(P, T, A, (size_t)2 * blockIdx.x, (size_t)2 * blockIdx.x + 1 < A.count, smem);
    else
        encrypt_one<LOGN, MODE, false>(P, T, A, blockIdx.x, smem);
}

template <int LOGN, int MODE>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS, (enc_blocks<LOGN, MODE, true>()))
void k_encode_encrypt_general(DevParams P, DevTables T, EncArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t count = A.general[0];
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x)
    {
        encrypt_one<LOGN, MODE, true>(P, T, A, A.general[1 + i], smem);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Split form of the symmetric path, used so that everything that does not need `a` overlaps with
// the (long, one-wave-per-SIMD) uniform sampler:
//   k_encode_rns : encode -> + e -> per prime signed reduction, residues stored (natural order)
//                  into the c0 slab itself -- no extra scratch.
//   k_ntt_fuse   : prime j of every ciphertext: load residues from c0_j, forward NTT, then
//                  c0_j = NTT(m+e) - s_hat . a_j  with a_j read from c1_j (ckks_sym.c:273-300).
// ------------------------------------------------------------------------------------------
// Like the fused kernel it comes as a fast / general pair: the fast launch declines a plaintext that holds
// a NaN or an infinite value (it appends the index to A.general and writes nothing); k_encode_rns_general
// walks that list with the EXACT transform.  Keeping the second transform out of the fast kernel is what
// keeps its registers (120 VGPRs, 104 at n = 16384; with a branch to the exact form inside: 150, and 128 +
// 92 B of spills at n = 16384).
template <int LOGN, bool ADD_ERR, bool GENERAL>
__device__ __forceinline__ void encode_rns_one(const DevParams &P, const DevTables &T, const EncArgs &A,
                                               const size_t b, unsigned char *smem)
{
    using G            = XformGeom<LOGN>;
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    const int t    = thread_index<GENERAL>();
    const int np   = P.nprimes;

    int64_t m[16];
    bool small;  // wave-uniform: every |m + e| of this wave is below 2 q_min
    int wg;
    encode_plaintext<LOGN, int64_t, GENERAL>(P, T, A.values, A.status, b, smem, m, small, wg);
    if constexpr (!GENERAL)
    {
        if (wg & kWgNonfinite)
        {
            if (t == 0) A.general[1 + atomicAdd(A.general, 1u)] = (uint32_t)b;
            return;
        }
    }
    if constexpr (ADD_ERR)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) m[e] += A.err[b * N + (e << CTOP) + t];
    }
    if (A.pte)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) A.pte[b * N + (e << CTOP) + t] = m[e];
    }
    // A small plaintext (every |m + e| < 2 q_min: the normal case) crosses to k_ntt_fuse as ONE row of
    // int32 -- parked in the c0 row of the LAST prime, which its own launch reads before it overwrites it --
    // instead of np rows of residues: k_ntt_fuse forms the representative m + 2 q_j itself.  At n = 16384
    // this kernel is HBM-bound (6 x 64 KiB of residues per plaintext against 48 KiB of inputs).
    if (A.compact)
    {
        const bool compact = !(wg & kWgNotSmall);
        if (t == 0) A.compact[b] = compact ? 1 : 0;
        if (compact)
        {
            uint32_t *dst = A.c0 + (b * np + (np - 1)) * N;
#pragma unroll
            for (int e = 0; e < 16; e++) dst[(e << CTOP) + t] = (uint32_t)(int32_t)m[e];
            return;
        }
    }
    for (int j = 0; j < np; j++)
    {
        uint32_t x[16];
        reduce_signed16(m, x, P.q[j], P.cr_hi[j], P.cr_lo[j], small);
        uint32_t *dst = A.c0 + (b * np + j) * N;
#pragma unroll
        for (int e = 0; e < 16; e++) dst[(e << CTOP) + t] = x[e];
    }
}

template <int LOGN, bool ADD_ERR>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS) void k_encode_rns(DevParams P, DevTables T,
                                                                      EncArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    encode_rns_one<LOGN, ADD_ERR, false>(P, T, A, blockIdx.x, smem);
}

template <int LOGN, bool ADD_ERR>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS) void k_encode_rns_general(DevParams P, DevTables T,
                                                                              EncArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t count = A.general[0];
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x)
    {
        encode_rns_one<LOGN, ADD_ERR, true>(P, T, A, A.general[1 + i], smem);
        __syncthreads();
    }
}

// n = 16384: at most 96 VGPRs (5 waves per SIMD instead of the 4 the 1024-thread workgroup needs) so
// that a workgroup fits beside the uniform sampler's chain waves (128 VGPRs, one wave per SIMD).
template <int LOGN, int MODE>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS)
__attribute__((amdgpu_waves_per_eu(LOGN == 14 ? 5 : (XformGeom<LOGN>::THREADS + 255) / 256))) void k_ntt_fuse(DevParams P, DevTables T, EncArgs A,
                                                                    int j)
{
    using G            = XformGeom<LOGN>;
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds32  = reinterpret_cast<uint32_t *>(smem);
    const int t      = threadIdx.x;
    const size_t b   = blockIdx.x;
    const int np     = P.nprimes;
    const uint32_t q = P.q[j], two_q = q << 1;
    uint32_t *poly   = A.c0 + (b * np + j) * N;

    // residues of this prime, or the compact int32 plaintext from the last prime's row (k_encode_rns):
    // m + 2 q_j is a representative in (0, 4q) the NTT accepts (modarith.cuh, reduce_signed16)
    const bool compact  = A.compact != nullptr && A.compact[b] != 0;
    const uint32_t *src = compact ? A.c0 + (b * np + (np - 1)) * N : poly;
    const uint32_t bias = compact ? two_q : 0u;
    uint32_t x[16];
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = src[(e << CTOP) + t] + bias;
    // issue the epilogue operands now; they land while the NTT runs.  At n = 16384 only `a` (HBM) is
    // prefetched; the L2-resident s_hat pairs are fetched after the NTT to stay within 96 VGPRs.
    // All epilogue accesses are in quad layout (transform.cuh, tile_to_quads): 1 KiB contiguous per wave
    // instruction instead of 16-byte pieces at a 64 / 128-byte lane stride.
    constexpr bool LATE_KEY = LOGN == 14;
    uint32_t a[16], w[16], wp[16];
    if constexpr (MODE == kModeSym)
    {
        load_quads(a, A.c1 + (b * np + j) * N, t);
        if constexpr (!LATE_KEY) load_quads_pairs(w, wp, T.s_hat + (size_t)2 * N * j, t);
    }
    ntt_tiles<LOGN>(x, T.ntt_rw + 2 * xform_table_len(N) * j, q, lds32, t);
    if constexpr (MODE == kModeSym && LATE_KEY) load_quads_pairs(w, wp, T.s_hat + (size_t)2 * N * j, t);
    // n <= 4096: the symmetric epilogue takes the LAZY transform output and canonicalises once (modarith.cuh,
    // sub_mul_canon); at n >= 8192 that form costs registers the kernel does not have (96-VGPR cap at n = 16384:
    // 36 B of spills) and the separate canonicalisation stays
    constexpr bool LAZY_EPI = MODE == kModeSym && LOGN <= 12;
    if constexpr (!LAZY_EPI)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) x[e] = canon4(x[e], q, two_q);
    }
    // one prime per launch: the exchange plane is free after the NTT's last barrier, the wave-local
    // transpose runs inside it (unpadded rows: the plane has no room for more at n = 16384 beside the chains)
    tile_to_quads<16>(x, lds32, t);
    if (A.ntt_pte)
    {
        uint32_t cx[16];
#pragma unroll
        for (int e = 0; e < 16; e++) cx[e] = LAZY_EPI ? canon4(x[e], q, two_q) : x[e];
        store_quads(A.ntt_pte + (b * np + j) * N, cx, t);
    }
    if constexpr (MODE == kModeSym)
    {
        uint32_t out[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            if constexpr (LAZY_EPI)
                out[e] = sub_mul_canon(x[e], a[e], w[e], wp[e], q, two_q);
            else
                out[e] = csub(x[e] + q - csub(mul_shoup_lazy(a[e], w[e], wp[e], q), q), q);
        }
        store_quads(poly, out, t);
    }
    else
    {
        store_quads(poly, x, t);
    }
}

// Batched stand-alone forward NTT (ntt_inpl, ntt.c:168-189) of `count` polynomials mod q_j,
// in place, natural-order in, bit-reversed-order canonical out.  Also emits the Shoup companion
// table when `pairs_out` is given (used once per key for NTT(s) and the public key).
template <int LOGN>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS) void k_ntt_polys(DevParams P, DevTables T, int j,
                                                                     uint32_t *polys,
                                                                     uint32_t *pairs_out)
{
    using G            = XformGeom<LOGN>;
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(smem);
    const int t     = threadIdx.x;
    uint32_t *poly  = polys + (size_t)blockIdx.x * N;
    const uint32_t q = P.q[j], two_q = q << 1;
    uint32_t x[16];
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = poly[(e << CTOP) + t];
    ntt_tiles<LOGN>(x, T.ntt_rw + 2 * xform_table_len(N) * j, q, lds32, t);
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = canon4(x[e], q, two_q);
    store16(poly + 16 * t, x);
    if (pairs_out)
    {
        uint32_t *po = pairs_out + ((size_t)blockIdx.x * N + 16 * t) * 2;
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            po[2 * e]     = x[e];
            po[2 * e + 1] = (uint32_t)((((uint64_t)x[e]) << 32) / q);
        }
    }
}

// Shoup companions for a table that is already in NTT form (public-key slabs): pairs[i] =
// (v[i], floor(v[i] * 2^32 / q_j)).  (A tile-major layout that makes the public-key kernel's pair loads
// contiguous over the wave was measured 4 % SLOWER on that kernel, gpurun_out/ab_pk_tiled.log: at 256
// VGPRs / 2 waves it is not the access pattern that limits it.)
__global__ void k_make_pairs(const uint32_t *vals, uint32_t *pairs, uint32_t q, int count)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
    {
        uint32_t v       = vals[i];
        pairs[2 * i]     = v;
        pairs[2 * i + 1] = (uint32_t)((((uint64_t)v) << 32) / q);
    }
}

// ------------------------------------------------------------------------------------------
// Verification side: the reference's round-trip check, batched.
//   ckks_decrypt   device/test/ckks_tests_common.c:136-153   d = c0 + c1 . NTT(s)
//   intt_inpl      device/lib/intt.c:144-222 (+ n^-1)        pt = INTT(d)
//   ckks_decode    device/test/ckks_tests_common.c:59-115    centred lift, / scale, fft_inpl
//                  (device/lib/fft.c:146-213), slot pick through the index map
// One workgroup per ciphertext, one prime per launch.  Decode is bit-exact with the reference's
// (same butterflies, same root table, IEEE division and float conversion).
// ------------------------------------------------------------------------------------------
struct VerifyArgs
{
    const uint32_t *c0;   // [B][np][n]
    const uint32_t *c1;   // [B][np][n]   (NULL: c0 slab already holds the value to invert)
    uint32_t *dec_ntt;    // optional [B][n]: c0 + c1 . NTT(s) mod q_j   (NTT form)
    uint32_t *pt;         // optional [B][n]: INTT of it, canonical, natural order
    float *values;        // optional [B][n/2]: decoded slots
    uint32_t in_primes;   // polynomials per record in c0/c1 (np, or 1 for a bare polynomial batch)
    int j;                // prime
};

template <int LOGN>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS) void k_decrypt_decode(DevParams P, DevTables T,
                                                                          VerifyArgs A)
{
    using G            = XformGeom<LOGN>;
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds32  = reinterpret_cast<uint32_t *>(smem);
    double *plane    = reinterpret_cast<double *>(smem);
    const int t      = threadIdx.x;
    const size_t b   = blockIdx.x;
    const int j      = A.j;
    const uint32_t q = P.q[j];
    const size_t rec = (b * A.in_primes + (A.in_primes > 1 ? j : 0)) * N + 16 * t;

    uint32_t x[16];
    load16(x, A.c0 + rec);
    if (A.c1)
    {
        uint32_t a[16], w[16], wp[16];
        load16(a, A.c1 + rec);
        load16_pairs(w, wp, T.s_hat, (size_t)j * N + 16 * t);
#pragma unroll
        for (int e = 0; e < 16; e++)
            x[e] = csub(csub(mul_shoup_lazy(a[e], w[e], wp[e], q), q) + x[e], q);
    }
    if (A.dec_ntt) store16(A.dec_ntt + b * N + 16 * t, x);
    if (!A.pt && !A.values) return;

    intt_tiles<LOGN>(x, T.intt_rw + (size_t)2 * N * j, q, lds32, t);
    const uint32_t inv_n = P.inv_n[j], inv_n_sh = P.inv_n_sh[j];
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = csub(mul_shoup_lazy(x[e], inv_n, inv_n_sh, q), q);
    if (A.pt)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) A.pt[b * N + (e << CTOP) + t] = x[e];
    }
    if (!A.values) return;

    double re[16], im[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        // representative in (-q/2, q/2], then 1/scale (ckks_tests_common.c:74-83)
        double dval = (x[e] > q / 2) ? -(double)(q - x[e]) : (double)x[e];
        re[e]       = __ddiv_rn(dval, P.scale);
        im[e]       = 0.0;
    }
    fft_tiles<LOGN>(re, im, T.ifft_w, plane, t);
    // values_decoded[i] = (flpt) Re res[index_map[i]]: this thread holds res[16t + e]; the slot
    // that reads it is inv_map[16t + e] when that is < n/2
    const uint4 *mp = reinterpret_cast<const uint4 *>(T.inv_map + 16 * t);
    uint4 m0 = mp[0], m1 = mp[1];
    uint32_t packed[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        uint32_t i = (packed[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
        if (i < (uint32_t)(N / 2)) A.values[b * (N / 2) + i] = (float)re[e];
    }
}

template <int LOGN>
static hipError_t launch_vfy(const DevParams &P, const DevTables &T, const VerifyArgs &A, size_t B,
                             hipStream_t st)
{
    using G      = XformGeom<LOGN>;
    size_t shmem = (size_t)G::SLOTS * sizeof(double);
    (void)hipFuncSetAttribute((const void *)k_decrypt_decode<LOGN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL((k_decrypt_decode<LOGN>), dim3((unsigned)B), dim3(G::THREADS), shmem, st, P, T, A);
    return hipGetLastError();
}

hipError_t launch_decrypt_decode(const DevParams &P, const DevTables &T, const uint32_t *c0,
                                 const uint32_t *c1, uint32_t in_primes, int j, uint32_t *dec_ntt,
                                 uint32_t *pt, float *values, size_t B, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    VerifyArgs A{c0, c1, dec_ntt, pt, values, in_primes, j};
    switch (P.logn)
    {
        case 10: return launch_vfy<10>(P, T, A, B, st);
        case 11: return launch_vfy<11>(P, T, A, B, st);
        case 12: return launch_vfy<12>(P, T, A, B, st);
        case 13: return launch_vfy<13>(P, T, A, B, st);
        case 14: return launch_vfy<14>(P, T, A, B, st);
        default: return hipErrorInvalidValue;
    }
}

// reduce_set_e_small (ckks_common.c:259-265) for every prime: int8 error -> residues, natural order,
// into a [count][np][n] slab (public-key generation feeds them to k_ntt_fuse).
__global__ void k_reduce_small(DevParams P, const int8_t *e, uint32_t *out, int count)
{
    const int n = P.n, np = P.nprimes;
    size_t i    = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)count * n) return;
    size_t b = i / n, k = i - b * n;
    int32_t v = e[i];
    for (int j = 0; j < np; j++) out[(b * np + j) * n + k] = (v < 0 ? P.q[j] : 0u) + (uint32_t)v;
}

hipError_t launch_reduce_small(const DevParams &P, const int8_t *e, uint32_t *out, size_t count,
                               hipStream_t st)
{
    if (count == 0) return hipSuccess;
    size_t total = count * P.n;
    hipLaunchKernelGGL(k_reduce_small, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, e, out,
                       (int)count);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launchers (called from se_context.cpp)
// ------------------------------------------------------------------------------------------
template <int LOGN, int MODE>
static hipError_t launch_enc_mode(const DevParams &P, const DevTables &T, const EncArgs &A, size_t B,
                                  hipStream_t st)
{
    using G             = XformGeom<LOGN>;
    const size_t planes = (size_t)G::SLOTS * sizeof(double);   // the IFFT's two f64 half-planes
    const size_t quads  = (size_t)(G::N / 16) * enc_quad_stride<MODE>() * sizeof(uint32_t);
    // n <= 4096: one u32 NTT plane (public key: three, for the three-way NTT) + the transpose region
    // behind it (not in the general public-key form)
    size_t shmem_fast = planes, shmem_gen = planes;
    if (LOGN <= 12)
    {
        const size_t ntt_planes = (size_t)(MODE == kModeAsym ? 3 : 1) * G::SLOTS * sizeof(uint32_t);
        // public key: the transpose region is aliased into the three planes (encrypt_one, QALIAS)
        shmem_fast = MODE == kModeAsym ? std::max(planes, std::max(ntt_planes, quads)) : std::max(planes, ntt_planes + quads);
        // (the general forms never alias the region; the general public-key form keeps the tile layout)
        shmem_gen  = MODE == kModeAsym ? std::max(planes, ntt_planes) : std::max(planes, ntt_planes + quads);
    }
    hipError_t e = hipMemsetAsync(A.general, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    EncArgs Af = A;
    Af.count   = B;
    if (enc_pairs<LOGN, MODE>()) shmem_fast = std::max(shmem_fast, (kPairParkWords + (size_t)G::N) * sizeof(uint32_t));
    const unsigned grid_fast = (unsigned)(enc_pairs<LOGN, MODE>() ? (B + 1) / 2 : B);   // two plaintexts per workgroup
    (void)hipFuncSetAttribute((const void *)k_encode_encrypt<LOGN, MODE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_fast);
    hipLaunchKernelGGL((k_encode_encrypt<LOGN, MODE>), dim3(grid_fast), dim3(G::THREADS), shmem_fast, st, P, T,
                       Af);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the plaintexts the fast form declined (normally none: the workgroups read a zero count and leave)
    const unsigned cus  = P.num_cus ? P.num_cus : 256u;
    const unsigned grid = (unsigned)std::min<size_t>(B, (size_t)4 * cus);
    (void)hipFuncSetAttribute((const void *)k_encode_encrypt_general<LOGN, MODE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_gen);
    hipLaunchKernelGGL((k_encode_encrypt_general<LOGN, MODE>), dim3(grid), dim3(G::THREADS), shmem_gen, st, P,
                       T, A);
    return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_enc(const DevParams &P, const DevTables &T, const EncArgs &A, int mode,
                             size_t B, hipStream_t st)
{
    if (!A.general) return hipErrorInvalidValue;   // the context's list of declined plaintexts
    switch (mode)
    {
        case kModeSym: return launch_enc_mode<LOGN, kModeSym>(P, T, A, B, st);
        case kModeAsym: return launch_enc_mode<LOGN, kModeAsym>(P, T, A, B, st);
        default: return launch_enc_mode<LOGN, kModeEncodeOnly>(P, T, A, B, st);
    }
}

hipError_t launch_encode_encrypt(const DevParams &P, const DevTables &T, const EncArgs &A, int mode,
                                 size_t B, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    switch (P.logn)
    {
        case 10: return launch_enc<10>(P, T, A, mode, B, st);
        case 11: return launch_enc<11>(P, T, A, mode, B, st);
        case 12: return launch_enc<12>(P, T, A, mode, B, st);
        case 13: return launch_enc<13>(P, T, A, mode, B, st);
        case 14: return launch_enc<14>(P, T, A, mode, B, st);
        default: return hipErrorInvalidValue;
    }
}

template <int LOGN, bool ADD_ERR>
static hipError_t launch_enc_rns_e(const DevParams &P, const DevTables &T, const EncArgs &A, size_t B,
                                   hipStream_t st)
{
    using G      = XformGeom<LOGN>;
    size_t shmem = (size_t)G::SLOTS * sizeof(double);
    if (!A.general) return hipErrorInvalidValue;   // the context's list of declined plaintexts
    hipError_t e = hipMemsetAsync(A.general, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    (void)hipFuncSetAttribute((const void *)k_encode_rns<LOGN, ADD_ERR>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL((k_encode_rns<LOGN, ADD_ERR>), dim3((unsigned)B), dim3(G::THREADS), shmem, st, P, T, A);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // plaintexts with NaN / infinite values (normally none: the workgroups read a zero count and leave)
    const unsigned cus  = P.num_cus ? P.num_cus : 256u;
    const unsigned grid = (unsigned)std::min<size_t>(B, (size_t)cus);
    (void)hipFuncSetAttribute((const void *)k_encode_rns_general<LOGN, ADD_ERR>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL((k_encode_rns_general<LOGN, ADD_ERR>), dim3(grid), dim3(G::THREADS), shmem, st, P, T, A);
    return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_enc_rns(const DevParams &P, const DevTables &T, const EncArgs &A, bool add_err,
                                 size_t B, hipStream_t st)
{
    return add_err ? launch_enc_rns_e<LOGN, true>(P, T, A, B, st) : launch_enc_rns_e<LOGN, false>(P, T, A, B, st);
}

hipError_t launch_encode_rns(const DevParams &P, const DevTables &T, const EncArgs &A, bool add_err,
                             size_t B, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    switch (P.logn)
    {
        case 10: return launch_enc_rns<10>(P, T, A, add_err, B, st);
        case 11: return launch_enc_rns<11>(P, T, A, add_err, B, st);
        case 12: return launch_enc_rns<12>(P, T, A, add_err, B, st);
        case 13: return launch_enc_rns<13>(P, T, A, add_err, B, st);
        case 14: return launch_enc_rns<14>(P, T, A, add_err, B, st);
        default: return hipErrorInvalidValue;
    }
}

template <int LOGN>
static hipError_t launch_nf(const DevParams &P, const DevTables &T, const EncArgs &A, int mode, int j,
                            size_t B, hipStream_t st)
{
    using G      = XformGeom<LOGN>;
    size_t shmem = (size_t)G::SLOTS * sizeof(uint32_t);
    dim3 grid((unsigned)B), block(G::THREADS);
    if (mode == kModeSym)
    {
        (void)hipFuncSetAttribute((const void *)k_ntt_fuse<LOGN, kModeSym>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL((k_ntt_fuse<LOGN, kModeSym>), grid, block, shmem, st, P, T, A, j);
    }
    else
    {
        (void)hipFuncSetAttribute((const void *)k_ntt_fuse<LOGN, kModeEncodeOnly>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL((k_ntt_fuse<LOGN, kModeEncodeOnly>), grid, block, shmem, st, P, T, A, j);
    }
    return hipGetLastError();
}

hipError_t launch_ntt_fuse(const DevParams &P, const DevTables &T, const EncArgs &A, int mode, int j,
                           size_t B, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    switch (P.logn)
    {
        case 10: return launch_nf<10>(P, T, A, mode, j, B, st);
        case 11: return launch_nf<11>(P, T, A, mode, j, B, st);
        case 12: return launch_nf<12>(P, T, A, mode, j, B, st);
        case 13: return launch_nf<13>(P, T, A, mode, j, B, st);
        case 14: return launch_nf<14>(P, T, A, mode, j, B, st);
        default: return hipErrorInvalidValue;
    }
}

template <int LOGN>
static hipError_t launch_ntt(const DevParams &P, const DevTables &T, int j, uint32_t *polys,
                             uint32_t *pairs, size_t count, hipStream_t st)
{
    using G      = XformGeom<LOGN>;
    size_t shmem = (size_t)G::SLOTS * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void *)k_ntt_polys<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)shmem);
    hipLaunchKernelGGL((k_ntt_polys<LOGN>), dim3((unsigned)count), dim3(G::THREADS), shmem, st, P, T,
                       j, polys, pairs);
    return hipGetLastError();
}

hipError_t launch_ntt_polys(const DevParams &P, const DevTables &T, int j, uint32_t *polys,
                            uint32_t *pairs, size_t count, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    switch (P.logn)
    {
        case 10: return launch_ntt<10>(P, T, j, polys, pairs, count, st);
        case 11: return launch_ntt<11>(P, T, j, polys, pairs, count, st);
        case 12: return launch_ntt<12>(P, T, j, polys, pairs, count, st);
        case 13: return launch_ntt<13>(P, T, j, polys, pairs, count, st);
        case 14: return launch_ntt<14>(P, T, j, polys, pairs, count, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_make_pairs(const uint32_t *vals, uint32_t *pairs, uint32_t q, size_t count,
                             hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_make_pairs, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, vals,
                       pairs, q, (int)count);
    return hipGetLastError();
}

}  // namespace seamd
