// transform.cuh -- register-tiled radix-16 passes for the two length-n transforms of the path.
//
// One workgroup of n/16 threads owns one polynomial; every thread holds 16 points in VGPRs.
// A "pass" runs up to four consecutive radix-2 stages entirely in registers; between passes the
// points are re-dealt through LDS (one write + one read per point) instead of one LDS round trip
// per stage.  n = 4096: 3 passes, 2 exchanges (instead of 12 barrier'd stages).
//
//   * inverse FFT of the CKKS encoder, FP64, DIF, rounds tt = 1,2,..,n/2, butterfly
//     (u, v) -> (u + v, (u - v) * W[h + j])           /root/reference/device/lib/fft.c:69-144
//   * forward negacyclic NTT, 32-bit residues, CT/Harvey, rounds h = 1,2,..,n/2, butterfly
//     (u, v) -> (u + v*R[h + g], u - v*R[h + g])      /root/reference/device/lib/ntt.c:124-165
//
// Both index their root table with  (n >> (i+1)) + (k >> (i+1))  where i is the bit in which the
// two butterfly inputs k, k + 2^i differ; the IFFT walks i upward, the NTT downward.
//
// Bit-exactness of the IFFT: every butterfly is the reference's, operand for operand --
// complex product in C99 Annex-G order (ac - bd, ad + bc), each op individually rounded
// (__dmul_rn/__dadd_rn/__dsub_rn are never contracted into FMAs).  Re-dealing points between
// threads does not change any arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "modarith.cuh"

namespace seamd {

// compile-time loop: f(integral_constant<int, I>) for I in [BEGIN, END)
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (BEGIN < END)
    {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

// Point index held in slot e (0..15) of thread t for a pass whose 16-point tiles span bits
// [C, C+4) of the index.
template <int C>
__device__ __forceinline__ int tile_index(int t, int e)
{
    return ((t >> C) << (C + 4)) | (e << C) | (t & ((1 << C) - 1));
}

// LDS slot of point k for an exchange between tile layouts CA and CB.  The pad is chosen per exchange so
// that BOTH deal patterns are bank-conflict-free under the gfx950 rules (MI355X_MICROARCH.md, LDS:
// ds_read/write_b32 and ds_read_b64 are served in two 32-lane groups, ds_write_b64 / ds_read2_b64 in
// four contiguous 16-lane groups, banks = dword address mod 32 resp. 64) while every thread's 16
// addresses stay base(t) + constant(e), i.e. immediate offsets:
//   both windows <= 4 : lanes walk strides 16 / 256  -> one pad element per 16   (k + (k >> 4))
//   both windows >= 5 : 32 consecutive lanes hold 32 consecutive points -> no pad
//   mixed             : two pad elements per 32                                  (k + 2 (k >> 5))
// (one pad per 16 under a window >= 5 makes lanes 0 and 31 of a group collide: t + (t >> 4) spans 33
// slots -- 1 024 conflict cycles per n = 4096 workgroup in round 1's SQ_LDS_BANK_CONFLICT together with
// the gather of the encoder; verified conflict-free for every exchange of n = 1024 .. 16384 with the
// lane-group model, tools/lds_conflicts.py.)
template <int CA, int CB>
__host__ __device__ constexpr int lds_slot(int k)
{
    constexpr int lo = CA < CB ? CA : CB, hi = CA < CB ? CB : CA;
    if constexpr (hi <= 4)
        return k + (k >> 4);
    else if constexpr (lo >= 5)
        return k;
    else
        return k + ((k >> 5) << 1);
}

template <int LOGN>
struct XformGeom
{
    static constexpr int N       = 1 << LOGN;
    static constexpr int THREADS = N / 16;
    static constexpr int PASSES  = (LOGN + 3) / 4;
    static constexpr int SLOTS   = N + N / 16;  // padded LDS elements
    // window offset of pass p
    static constexpr int ifft_c(int p) { return (4 * p < LOGN - 4) ? 4 * p : LOGN - 4; }
    static constexpr int ntt_c(int p) { return (LOGN - 4 - 4 * p > 0) ? LOGN - 4 - 4 * p : 0; }
};

// Re-deal 16 values per thread from tile layout C_FROM to tile layout C_TO through `lds`: write, barrier, read,
// barrier (leaves the workgroup synchronised and `lds` free for reuse).
//
// lds_slot() is additive over disjoint bit fields (shifts distribute over OR), so the slot of point
// tile_index<C>(t, e) is slot(thread part) + slot(e << C): ONE base register per deal and sixteen
// compile-time offsets that land in the instructions' immediate fields.  Written as slot(tile_index(t, e))
// the compiler does not see this: it computed the 16 addresses of every deal separately and, in the fused
// kernels, kept them in registers across the prime loop (64 VGPRs of addresses for the two NTT exchanges).
// (Measured and not adopted -- experiments/: leading barriers + wave-local first / last exchanges, neutral;
// the window 4 -> 0 exchange as a DPP register transpose, +1.5 %; FP64 exchanges through a half-size plane.)
template <int C_FROM, int C_TO, typename T>
__device__ __forceinline__ void redeal(T (&v)[16], T *lds, int t)
{
    T *wr = lds + lds_slot<C_FROM, C_TO>(tile_index<C_FROM>(t, 0));
    static_for<0, 16>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        wr[lds_slot<C_FROM, C_TO>(e << C_FROM)] = v[e];
    });
    __syncthreads();
    const T *rd = lds + lds_slot<C_FROM, C_TO>(tile_index<C_TO>(t, 0));
    static_for<0, 16>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        v[e] = rd[lds_slot<C_FROM, C_TO>(e << C_TO)];
    });
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Tile layout -> "quad" layout, wave-local.  After the last NTT pass thread t owns the 16 CONSECUTIVE
// coefficients 16t .. 16t+15, so every global access of the epilogue (a, key pairs, c0 / c1 stores) would
// be 16-byte pieces at a 64- or 128-byte lane stride: 64 different cache lines per wave instruction,
// 4-8 instructions per operand.  Measured on the fused symmetric kernel: the `a` and key-pair loads alone
// cost 1.0 of its 3.6 ms (ablation, gpurun_out/abl_sym_loads.log) although the key table is L2-resident.
// A wave's 64 tiles are the 1024 consecutive coefficients 1024 w .. 1024 w + 1023, so one wave-local LDS
// transpose (4 ds_write_b128 + 4 ds_read_b128 per thread, no barrier: LDS executes a wave's accesses in
// order) re-deals them so that in slot group i lane l holds coefficients 1024 w + 256 i + 4 l + (0..3):
// every global instruction then covers 1 KiB contiguous.  Rows of 16 words are padded to STRIDE words
// (multiple of 4): 28 makes the writes conflict-free and the reads cost 16 extra cycles per transpose, 16
// (no pad) the other way round (96 / 0) -- tools/lds_conflicts.py.
// ------------------------------------------------------------------------------------------
template <int STRIDE>
__device__ __forceinline__ void tile_to_quads(uint32_t (&x)[16], uint32_t *lds_region, int t)
{
    static_assert(STRIDE % 4 == 0 && STRIDE >= 16, "rows stay 16-byte aligned");
    const int lane = t & 63;
    uint32_t *w    = lds_region + (t >> 6) * (64 * STRIDE);
    uint4 *row     = reinterpret_cast<uint4 *>(w + STRIDE * lane);
#pragma unroll
    for (int c = 0; c < 4; c++) row[c] = make_uint4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
    // the reads below fetch OTHER lanes' rows: keep the compiler from moving them above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int q   = 64 * i + lane;
        const uint4 v = *reinterpret_cast<const uint4 *>(w + STRIDE * (q >> 2) + 4 * (q & 3));
        x[4 * i] = v.x, x[4 * i + 1] = v.y, x[4 * i + 2] = v.z, x[4 * i + 3] = v.w;
    }
}

// coefficient index (within the polynomial) of slot 4 i of thread t in quad layout
__device__ __forceinline__ int quad_index(int t, int i)
{
    return ((t >> 6) << 10) + (i << 8) + ((t & 63) << 2);
}

// ------------------------------------------------------------------------------------------
// Complex product (a + ib)(c + id) as the reference's build evaluates the C operator `*` on
// `double complex` (fft.c:139, :205): Annex-G form x = ac - bd, y = ad + bc, one rounding per operation.
// EXACT additionally reproduces what gcc's expansion does when the result is NaN + iNaN: libgcc's __muldc3
// "recovers infinities" (C99 G.5.1) -- an infinite factor is boxed to (+-1 | +-0), NaNs of the other factor
// become signed zeros, and the product is INFINITY * (ac - bd), INFINITY * (ad + bc).  That never triggers
// on finite data (the hot kernels use EXACT = false and only ever see finite values: encode_plaintext hands
// a plaintext with a NaN / infinite value to the EXACT path), but it decides which coefficients of such a
// plaintext are NaN (accepted by ckks_common.c:195, stored as INT64_MIN) and which are infinite (`return
// false`).  Restated from the published libgcc algorithm (libgcc2.c, __mulMODE3); the GPU tests pin it
// against outputs of the compiled reference.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double dev_copysign(double mag, double sgn)
{
    return __longlong_as_double((__double_as_longlong(mag) & 0x7FFFFFFFFFFFFFFFll) |
                                (__double_as_longlong(sgn) & (long long)0x8000000000000000ull));
}
__device__ __forceinline__ bool dev_isinf(double v)
{
    return (__double_as_longlong(v) & 0x7FFFFFFFFFFFFFFFll) == 0x7FF0000000000000ll;
}
__device__ __forceinline__ bool dev_isnan(double v) { return v != v; }

__device__ inline void cmul_recover(double a, double b, double c, double d, double ac, double bd, double ad,
                                    double bc, double &x, double &y)
{
    bool recalc = false;
    if (dev_isinf(a) || dev_isinf(b))
    {
        a = dev_copysign(dev_isinf(a) ? 1.0 : 0.0, a);
        b = dev_copysign(dev_isinf(b) ? 1.0 : 0.0, b);
        if (dev_isnan(c)) c = dev_copysign(0.0, c);
        if (dev_isnan(d)) d = dev_copysign(0.0, d);
        recalc = true;
    }
    if (dev_isinf(c) || dev_isinf(d))
    {
        c = dev_copysign(dev_isinf(c) ? 1.0 : 0.0, c);
        d = dev_copysign(dev_isinf(d) ? 1.0 : 0.0, d);
        if (dev_isnan(a)) a = dev_copysign(0.0, a);
        if (dev_isnan(b)) b = dev_copysign(0.0, b);
        recalc = true;
    }
    if (!recalc && (dev_isinf(ac) || dev_isinf(bd) || dev_isinf(ad) || dev_isinf(bc)))
    {
        if (dev_isnan(a)) a = dev_copysign(0.0, a);
        if (dev_isnan(b)) b = dev_copysign(0.0, b);
        if (dev_isnan(c)) c = dev_copysign(0.0, c);
        if (dev_isnan(d)) d = dev_copysign(0.0, d);
        recalc = true;
    }
    if (recalc)
    {
        const double inf = __longlong_as_double(0x7FF0000000000000ll);
        x = __dmul_rn(inf, __dsub_rn(__dmul_rn(a, c), __dmul_rn(b, d)));
        y = __dmul_rn(inf, __dadd_rn(__dmul_rn(a, d), __dmul_rn(b, c)));
    }
}

template <bool EXACT>
__device__ __forceinline__ void cmul_annexg(double a, double b, double c, double d, double &x, double &y)
{
    const double ac = __dmul_rn(a, c), bd = __dmul_rn(b, d), ad = __dmul_rn(a, d), bc = __dmul_rn(b, c);
    x = __dsub_rn(ac, bd);
    y = __dadd_rn(ad, bc);
    if constexpr (EXACT)
    {
        if (dev_isnan(x) && dev_isnan(y)) cmul_recover(a, b, c, d, ac, bd, ad, bc, x, y);
    }
}

// ------------------------------------------------------------------------------------------
// IFFT pass: stages for local bits [B_LO, B_HI) of a tile at window C, ascending.
// ------------------------------------------------------------------------------------------
template <int LOGN, int C, int B_LO, int B_HI, bool EXACT = false>
__device__ __forceinline__ void ifft_pass(double (&re)[16], double (&im)[16],
                                          const double *__restrict__ W, int t)
{
    constexpr int N = 1 << LOGN;
    // top window: t < n/16 = 2^C, so t >> C == 0 and the root indices are compile-time constants
    // (uniform addresses -> scalar loads, no VGPRs)
    const int thi   = (C + 4 >= LOGN) ? 0 : (t >> C);
    static_for<B_LO, B_HI>([&](auto bc) {
        constexpr int b      = decltype(bc)::value;
        constexpr int h      = N >> (C + b + 1);
        constexpr int groups = 1 << (3 - b);  // distinct twiddles this thread needs in this stage
        static_for<0, groups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            // window 0: the thread-major copy behind the table (se_types.h, xform_table_len)
            const int idx   = (C == 0) ? N + ((8 >> b) - 1 + g) * (N / 16) + t : h + ((thi << (3 - b)) | g);
            const double2 w = *reinterpret_cast<const double2 *>(W + 2 * idx);
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int e0 = (g << (b + 1)) | decltype(rc)::value;
                constexpr int e1 = e0 | (1 << b);
                double ar = __dsub_rn(re[e0], re[e1]);
                double ai = __dsub_rn(im[e0], im[e1]);
                re[e0]    = __dadd_rn(re[e0], re[e1]);
                im[e0]    = __dadd_rn(im[e0], im[e1]);
                cmul_annexg<EXACT>(ar, ai, w.x, w.y, re[e1], im[e1]);
            });
        });
    });
}

// First IFFT pass (window 0, stages 0..3) for REAL input (im == +0 everywhere: the CKKS encoder's
// conjugate-symmetric slot vector, ckks_common.c:148-150).  A butterfly whose two inputs are still real
// needs 2 adds + 2 multiplies instead of the 10 operations of the complex form:
//     (u, v) -> (u + v, (u - v) * w)  =  (u + v,  (ar * w.x) + i (ar * w.y)),   ar = u - v
// Position e of a tile is real before stage b iff its low b bits are zero, so 8 + 4 + 2 + 1 = 15 of
// the pass's 32 butterflies take the short form.  The general form would compute ar*w.x - (+0)*w.y and
// ar*w.y + (+0)*w.x: adding or subtracting a zero leaves every NONZERO value bit-identical and can only
// change the sign of a zero result; a zero's sign never reaches a nonzero value through +, -, * and
// vanishes in the encoder's final round-to-int64 (and in its |.| overflow test).  So the int64
// plaintext is bit-identical; the complex-output operators (ifft_inpl) keep the general form.
template <int LOGN>
__device__ __forceinline__ void ifft_pass0_real(double (&re)[16], double (&im)[16],
                                                const double *__restrict__ W, int t)
{
    constexpr int N = 1 << LOGN;
    static_for<0, 4>([&](auto bc) {
        constexpr int b      = decltype(bc)::value;
        constexpr int groups = 1 << (3 - b);
        static_for<0, groups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const int idx   = N + ((8 >> b) - 1 + g) * (N / 16) + t;   // thread-major copy (se_types.h)
            const double2 w = *reinterpret_cast<const double2 *>(W + 2 * idx);
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int r  = decltype(rc)::value;
                constexpr int e0 = (g << (b + 1)) | r;
                constexpr int e1 = e0 | (1 << b);
                if constexpr (r == 0)
                {
                    const double ar = __dsub_rn(re[e0], re[e1]);
                    re[e0]          = __dadd_rn(re[e0], re[e1]);
                    re[e1]          = __dmul_rn(ar, w.x);
                    im[e1]          = __dmul_rn(ar, w.y);
                }
                else
                {
                    double ar = __dsub_rn(re[e0], re[e1]);
                    double ai = __dsub_rn(im[e0], im[e1]);
                    re[e0]    = __dadd_rn(re[e0], re[e1]);
                    im[e0]    = __dadd_rn(im[e0], im[e1]);
                    re[e1]    = __dsub_rn(__dmul_rn(ar, w.x), __dmul_rn(ai, w.y));
                    im[e1]    = __dadd_rn(__dmul_rn(ar, w.y), __dmul_rn(ai, w.x));
                }
            });
        });
    });
}

// Whole IFFT: input in tile layout 0 (thread t holds points 16t..16t+15), output in tile layout
// LOGN-4 (thread t holds points t + (n/16)*e).  `plane` = LDS scratch of XformGeom::SLOTS doubles.
// REAL_IN: the imaginary parts of the input are all +0 (im[] need not be initialised except im[0]).
// EXACT: the Annex-G product with libgcc's infinity recovery (cmul_annexg) -- for plaintexts that hold NaN or
// infinite values and for the stand-alone operators, whose caller may pass anything; im[] fully initialised.
template <int LOGN, bool REAL_IN = false, bool EXACT = false>
__device__ __forceinline__ void ifft_tiles(double (&re)[16], double (&im)[16],
                                           const double *__restrict__ W, double *plane, int t)
{
    using G = XformGeom<LOGN>;
    if constexpr (REAL_IN && !EXACT)
        ifft_pass0_real<LOGN>(re, im, W, t);
    else
        ifft_pass<LOGN, 0, 0, 4, EXACT>(re, im, W, t);
    redeal<0, 4>(re, plane, t);
    redeal<0, 4>(im, plane, t);
    ifft_pass<LOGN, 4, 0, 4, EXACT>(re, im, W, t);
    if constexpr (LOGN <= 12)
    {
        constexpr int C2 = G::ifft_c(2);  // 6, 7 or 8
        redeal<4, C2>(re, plane, t);
        redeal<4, C2>(im, plane, t);
        ifft_pass<LOGN, C2, 8 - C2, 4, EXACT>(re, im, W, t);
    }
    else
    {
        redeal<4, 8>(re, plane, t);
        redeal<4, 8>(im, plane, t);
        ifft_pass<LOGN, 8, 0, 4, EXACT>(re, im, W, t);
        constexpr int C3 = G::ifft_c(3);  // 9 or 10
        redeal<8, C3>(re, plane, t);
        redeal<8, C3>(im, plane, t);
        ifft_pass<LOGN, C3, 12 - C3, 4, EXACT>(re, im, W, t);
    }
}

// ------------------------------------------------------------------------------------------
// HALF-SIZE IFFT of the CKKS slot vector, TWO plaintexts per workgroup (n = 4096, round 6).  The encoder's input is
// real and reverse-symmetric in stored order, A[n-1-k] = A[k] (ckks_common.c:139-153: slot and conjugate slot of the
// index map are bit-reversed complements), and every DIF stage i < logn-1 maps the half k < n/2 onto itself (its
// butterflies pair k with k + 2^i, both below n/2): stages 0 .. logn-2 on the lower half ARE the reference's own
// operations on those points, bit for bit.  Only the last stage pairs u = X[k] with v = X[k + n/2], and in exact
// arithmetic v = conj(u), so
//     out[k]       = u + v             = 2 Re u
//     out[k + n/2] = (u - v) * W[1]    -> real part -(2 Im u) * Im W[1]            (fft.c:118-141, Annex-G product)
// The reference computes v from the upper half with its own roundings (libm's root table is mirror-symmetric only up
// to the last ulp), so its outputs differ from these by a few ulp -- the CALLER detects the coefficients whose rounding
// to integer could differ and redoes those plaintexts with ifft_tiles (encode_encrypt.hip, encode_pair_half: rigorous
// bound, exact redo).  46 % of the butterflies of ifft_tiles and half its LDS traffic per plaintext.  A half-size
// problem fills only 128 register tiles, so a 256-thread workgroup transforms TWO plaintexts side by side (with one
// plaintext half the waves idle through the register passes and the kernel gains 3 % instead of 10 %): waves 0, 1 run
// passes 0 and 1 (windows 0 and 4) on plaintext A's 128 lower tiles, waves 2, 3 on plaintext B's; then every thread
// takes 8 points of A and 8 of B through the three remaining stages.
// In: thread t holds the real points 16T .. 16T+15 of plaintext t >> 7 in re[] (T = t & 127, im[0] = +0).
// Out: the REAL outputs t + 256 e of A are re[e] (e < 8), im[e-8] (e >= 8); of B: re[8+e], im[e].
// `plane` = XformGeom::SLOTS doubles: A's exchange region in the lower half, B's in the upper.
// ------------------------------------------------------------------------------------------
template <int LOGN>
__device__ __forceinline__ void ifft_pair_real_half(double (&re)[16], double (&im)[16], const double *__restrict__ W,
                                                    double *plane, int t)
{
    static_assert(LOGN == 12, "four waves: two per plaintext in the register passes");
    using G             = XformGeom<LOGN>;
    constexpr int N     = 1 << LOGN;
    constexpr int HALF  = G::SLOTS / 2;
    static_assert(lds_slot<0, 4>(N / 2 - 1) < HALF && lds_slot<4, 8>(N / 2 - 1) < HALF, "a plaintext's lower half fits its region");
    const int T   = t & 127;              // tile of this thread in the register passes
    double *mine  = plane + (t >> 7) * HALF;
    ifft_pass0_real<LOGN>(re, im, W, T);
    auto exchange_0_4 = [&](double (&v)[16]) {
        double *wr = mine + lds_slot<0, 4>(tile_index<0>(T, 0));
        static_for<0, 16>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            wr[lds_slot<0, 4>(e << 0)] = v[e];
        });
        __syncthreads();
        const double *rd = mine + lds_slot<0, 4>(tile_index<4>(T, 0));
        static_for<0, 16>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            v[e] = rd[lds_slot<0, 4>(e << 4)];
        });
        __syncthreads();
    };
    exchange_0_4(re);
    exchange_0_4(im);
    ifft_pass<LOGN, 4, 0, 4>(re, im, W, T);
    // window 4 -> 8: every thread takes points (e << 8) | t, e < 8, of BOTH plaintexts (A into slots 0..7, B into 8..15)
    auto exchange_4_8 = [&](double (&v)[16]) {
        double *wr = mine + lds_slot<4, 8>(tile_index<4>(T, 0));
        static_for<0, 16>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            wr[lds_slot<4, 8>(e << 4)] = v[e];
        });
        __syncthreads();
        const double *rd = plane + lds_slot<4, 8>(tile_index<8>(t, 0));
        static_for<0, 8>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            v[e]     = rd[lds_slot<4, 8>(e << 8)];
            v[8 + e] = rd[HALF + lds_slot<4, 8>(e << 8)];
        });
        __syncthreads();
    };
    exchange_4_8(re);
    exchange_4_8(im);
    // stages 8, 9, 10 (bits 0..2 of e): root index (n >> (i+1)) + (k >> (i+1)), uniform over the workgroup
    static_for<0, 3>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        constexpr int h = N >> (8 + b + 1);
        static_for<0, (4 >> b)>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const double2 w = *reinterpret_cast<const double2 *>(W + 2 * (h + g));
            static_for<0, (2 << b)>([&](auto rc) {
                constexpr int r  = decltype(rc)::value;                       // low bits: butterfly; top bit: plaintext
                constexpr int e0 = ((r >> b) << 3) | (g << (b + 1)) | (r & ((1 << b) - 1));
                constexpr int e1 = e0 | (1 << b);
                double ar = __dsub_rn(re[e0], re[e1]);
                double ai = __dsub_rn(im[e0], im[e1]);
                re[e0]    = __dadd_rn(re[e0], re[e1]);
                im[e0]    = __dadd_rn(im[e0], im[e1]);
                cmul_annexg<false>(ar, ai, w.x, w.y, re[e1], im[e1]);
            });
        });
    });
    // stage 11 with v = conj(u)
    const double w1y = W[3];   // Im W[1]
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const double i2 = __dadd_rn(im[e], im[e]);
        im[e]           = -__dmul_rn(i2, w1y);
        re[e]           = __dadd_rn(re[e], re[e]);
    }
}

// ------------------------------------------------------------------------------------------
// NTT pass: stages for local bits [B_LO, B_HI) of a tile at window C, descending.
// RW = interleaved (-root mod 2^32, shoup(root)) table of one prime.
// ------------------------------------------------------------------------------------------
template <int LOGN, int C, int B_LO, int B_HI>
__device__ __forceinline__ void ntt_pass(uint32_t (&x)[16], const uint32_t *__restrict__ RW,
                                         uint32_t q, uint32_t two_q, int t)
{
    constexpr int N = 1 << LOGN;
    const int thi   = (C + 4 >= LOGN) ? 0 : (t >> C);
    static_for<0, B_HI - B_LO>([&](auto sc) {
        constexpr int b      = B_HI - 1 - decltype(sc)::value;  // descending
        constexpr int h      = N >> (C + b + 1);
        constexpr int groups = 1 << (3 - b);
        static_for<0, groups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            // window 0: the thread-major copy behind the table (se_types.h, xform_table_len)
            const int idx   = (C == 0) ? N + ((8 >> b) - 1 + g) * (N / 16) + t : h + ((thi << (3 - b)) | g);
            const uint2 rw  = *reinterpret_cast<const uint2 *>(RW + 2 * idx);
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int e0 = (g << (b + 1)) | decltype(rc)::value;
                constexpr int e1 = e0 | (1 << b);
                ct_butterfly(x[e0], x[e1], rw.x, rw.y, q, two_q);
            });
        });
    });
}

// Whole NTT: input in tile layout LOGN-4 with values in [0, 2q] (anything < 4q), output in tile
// layout 0 (thread t holds 16t..16t+15 of the bit-reversed-order result), values in [0,4q).
template <int LOGN>
__device__ __forceinline__ void ntt_tiles(uint32_t (&x)[16], const uint32_t *__restrict__ RW,
                                          uint32_t q, uint32_t *lds, int t)
{
    using G              = XformGeom<LOGN>;
    const uint32_t two_q = q << 1;
    constexpr int C0     = LOGN - 4;
    constexpr int C1     = G::ntt_c(1);  // LOGN - 8
    ntt_pass<LOGN, C0, 0, 4>(x, RW, q, two_q, t);
    redeal<C0, C1>(x, lds, t);
    ntt_pass<LOGN, C1, 0, 4>(x, RW, q, two_q, t);
    if constexpr (LOGN <= 12)
    {
        redeal<C1, 0>(x, lds, t);
        ntt_pass<LOGN, 0, 0, C1>(x, RW, q, two_q, t);  // remaining bits C1-1 .. 0
    }
    else
    {
        constexpr int C2 = G::ntt_c(2);  // LOGN - 12 (1 or 2)
        redeal<C1, C2>(x, lds, t);
        ntt_pass<LOGN, C2, 0, 4>(x, RW, q, two_q, t);
        redeal<C2, 0>(x, lds, t);
        ntt_pass<LOGN, 0, 0, C2>(x, RW, q, two_q, t);
    }
}

// ------------------------------------------------------------------------------------------
// Three forward NTTs mod the same prime at once (the public-key path: u_hat, NTT(e1), NTT(m+e0),
// ckks_asym.c:235-284).  Every root pair is loaded once for three butterflies, the three point sets
// cross LDS through three planes under ONE barrier pair per exchange, and the three independent
// butterfly streams give the two-waves-per-SIMD kernel the ILP it lacks (7.16 -> 6.71 ms per 65 536 at
// n = 4096; requesting the roots before the exchange on top of this was slower, 6.95 ms).
// ------------------------------------------------------------------------------------------
template <int LOGN, int C_FROM, int C_TO>
__device__ __forceinline__ void redeal3(uint32_t (&a)[16], uint32_t (&b)[16], uint32_t (&c)[16],
                                        uint32_t *lds, int t)
{
    constexpr int SL = XformGeom<LOGN>::SLOTS;
    uint32_t *wr = lds + lds_slot<C_FROM, C_TO>(tile_index<C_FROM>(t, 0));
    static_for<0, 16>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        constexpr int s = lds_slot<C_FROM, C_TO>(e << C_FROM);
        wr[s] = a[e], wr[SL + s] = b[e], wr[2 * SL + s] = c[e];
    });
    __syncthreads();
    const uint32_t *rd = lds + lds_slot<C_FROM, C_TO>(tile_index<C_TO>(t, 0));
    static_for<0, 16>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        constexpr int s = lds_slot<C_FROM, C_TO>(e << C_TO);
        a[e] = rd[s], b[e] = rd[SL + s], c[e] = rd[2 * SL + s];
    });
    __syncthreads();
}

template <int LOGN, int C, int B_LO, int B_HI>
__device__ __forceinline__ void ntt_pass3(uint32_t (&x)[16], uint32_t (&y)[16], uint32_t (&z)[16],
                                          const uint32_t *__restrict__ RW, uint32_t q,
                                          uint32_t two_q, int t)
{
    constexpr int N = 1 << LOGN;
    const int thi   = (C + 4 >= LOGN) ? 0 : (t >> C);
    static_for<0, B_HI - B_LO>([&](auto sc) {
        constexpr int b      = B_HI - 1 - decltype(sc)::value;  // descending
        constexpr int h      = N >> (C + b + 1);
        constexpr int groups = 1 << (3 - b);
        static_for<0, groups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const int idx   = (C == 0) ? N + ((8 >> b) - 1 + g) * (N / 16) + t : h + ((thi << (3 - b)) | g);
            const uint2 rw  = *reinterpret_cast<const uint2 *>(RW + 2 * idx);
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int e0 = (g << (b + 1)) | decltype(rc)::value;
                constexpr int e1 = e0 | (1 << b);
                ct_butterfly(x[e0], x[e1], rw.x, rw.y, q, two_q);
                ct_butterfly(y[e0], y[e1], rw.x, rw.y, q, two_q);
                ct_butterfly(z[e0], z[e1], rw.x, rw.y, q, two_q);
            });
        });
    });
}

template <int LOGN>
__device__ __forceinline__ void ntt_tiles3(uint32_t (&x)[16], uint32_t (&y)[16], uint32_t (&z)[16],
                                           const uint32_t *__restrict__ RW, uint32_t q, uint32_t *lds,
                                           int t)
{
    using G              = XformGeom<LOGN>;
    const uint32_t two_q = q << 1;
    constexpr int C0     = LOGN - 4;
    constexpr int C1     = G::ntt_c(1);
    ntt_pass3<LOGN, C0, 0, 4>(x, y, z, RW, q, two_q, t);
    redeal3<LOGN, C0, C1>(x, y, z, lds, t);
    ntt_pass3<LOGN, C1, 0, 4>(x, y, z, RW, q, two_q, t);
    if constexpr (LOGN <= 12)
    {
        redeal3<LOGN, C1, 0>(x, y, z, lds, t);
        ntt_pass3<LOGN, 0, 0, C1>(x, y, z, RW, q, two_q, t);
    }
    else
    {
        constexpr int C2 = G::ntt_c(2);
        redeal3<LOGN, C1, C2>(x, y, z, lds, t);
        ntt_pass3<LOGN, C2, 0, 4>(x, y, z, RW, q, two_q, t);
        redeal3<LOGN, C2, 0>(x, y, z, lds, t);
        ntt_pass3<LOGN, 0, 0, C2>(x, y, z, RW, q, two_q, t);
    }
}

// ------------------------------------------------------------------------------------------
// Verification side (used by k_decrypt_decode): inverse NTT and forward FFT, same tiling.
// ------------------------------------------------------------------------------------------

// INTT pass: Gentleman-Sande stages for local bits [B_LO, B_HI), ascending (intt.c:144-222;
// the reference merges the 1/n scaling into its last round -- the caller multiplies by n^-1
// afterwards, which is the same exact residue).
template <int LOGN, int C, int B_LO, int B_HI>
__device__ __forceinline__ void intt_pass(uint32_t (&x)[16], const uint32_t *__restrict__ RW,
                                          uint32_t neg_q, uint32_t two_q, int t)
{
    constexpr int N = 1 << LOGN;
    const int thi   = (C + 4 >= LOGN) ? 0 : (t >> C);
    static_for<B_LO, B_HI>([&](auto bc) {
        constexpr int b      = decltype(bc)::value;
        constexpr int h      = N >> (C + b + 1);
        constexpr int groups = 1 << (3 - b);
        static_for<0, groups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const int idx   = h + ((thi << (3 - b)) | g);
            const uint2 rw  = *reinterpret_cast<const uint2 *>(RW + 2 * idx);
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int e0 = (g << (b + 1)) | decltype(rc)::value;
                constexpr int e1 = e0 | (1 << b);
                gs_butterfly(x[e0], x[e1], rw.x, rw.y, neg_q, two_q);
            });
        });
    });
}

// Whole INTT (without the final 1/n): input tile layout 0, values < 2q; output tile layout
// LOGN-4, values in [0,2q).
template <int LOGN>
__device__ __forceinline__ void intt_tiles(uint32_t (&x)[16], const uint32_t *__restrict__ RW,
                                           uint32_t q, uint32_t *lds, int t)
{
    using G              = XformGeom<LOGN>;
    const uint32_t two_q = q << 1, neg_q = 0u - q;
    intt_pass<LOGN, 0, 0, 4>(x, RW, neg_q, two_q, t);
    redeal<0, 4>(x, lds, t);
    intt_pass<LOGN, 4, 0, 4>(x, RW, neg_q, two_q, t);
    if constexpr (LOGN <= 12)
    {
        constexpr int C2 = G::ifft_c(2);
        redeal<4, C2>(x, lds, t);
        intt_pass<LOGN, C2, 8 - C2, 4>(x, RW, neg_q, two_q, t);
    }
    else
    {
        redeal<4, 8>(x, lds, t);
        intt_pass<LOGN, 8, 0, 4>(x, RW, neg_q, two_q, t);
        constexpr int C3 = G::ifft_c(3);
        redeal<8, C3>(x, lds, t);
        intt_pass<LOGN, C3, 12 - C3, 4>(x, RW, neg_q, two_q, t);
    }
}

// Forward FFT pass (fft.c:146-213): DIT stages for local bits [B_LO, B_HI), descending;
// (u, v) -> (u + v*s, u - v*s) with s = conj(W[h + j]) = (W.re, -W.im), product in Annex-G order.
template <int LOGN, int C, int B_LO, int B_HI, bool EXACT = false>
__device__ __forceinline__ void fft_pass(double (&re)[16], double (&im)[16],
                                         const double *__restrict__ W, int t)
{
    constexpr int N = 1 << LOGN;
    const int thi   = (C + 4 >= LOGN) ? 0 : (t >> C);
    static_for<0, B_HI - B_LO>([&](auto sc) {
        constexpr int b      = B_HI - 1 - decltype(sc)::value;
        constexpr int h      = N >> (C + b + 1);
        constexpr int groups = 1 << (3 - b);
        static_for<0, groups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const int idx   = h + ((thi << (3 - b)) | g);
            const double2 w = *reinterpret_cast<const double2 *>(W + 2 * idx);
            const double c = w.x, d = -w.y;
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int e0 = (g << (b + 1)) | decltype(rc)::value;
                constexpr int e1 = e0 | (1 << b);
                double vr, vi;
                cmul_annexg<EXACT>(re[e1], im[e1], c, d, vr, vi);
                re[e1]    = __dsub_rn(re[e0], vr);
                im[e1]    = __dsub_rn(im[e0], vi);
                re[e0]    = __dadd_rn(re[e0], vr);
                im[e0]    = __dadd_rn(im[e0], vi);
            });
        });
    });
}

// Whole forward FFT: input tile layout LOGN-4, output tile layout 0.
template <int LOGN, bool EXACT = false>
__device__ __forceinline__ void fft_tiles(double (&re)[16], double (&im)[16],
                                          const double *__restrict__ W, double *plane, int t)
{
    using G          = XformGeom<LOGN>;
    constexpr int C0 = LOGN - 4;
    fft_pass<LOGN, C0, 0, 4, EXACT>(re, im, W, t);
    constexpr int C1 = G::ntt_c(1);
    redeal<C0, C1>(re, plane, t);
    redeal<C0, C1>(im, plane, t);
    fft_pass<LOGN, C1, 0, 4, EXACT>(re, im, W, t);
    if constexpr (LOGN <= 12)
    {
        redeal<C1, 0>(re, plane, t);
        redeal<C1, 0>(im, plane, t);
        fft_pass<LOGN, 0, 0, C1, EXACT>(re, im, W, t);
    }
    else
    {
        constexpr int C2 = G::ntt_c(2);
        redeal<C1, C2>(re, plane, t);
        redeal<C1, C2>(im, plane, t);
        fft_pass<LOGN, C2, 0, 4, EXACT>(re, im, W, t);
        redeal<C2, 0>(re, plane, t);
        redeal<C2, 0>(im, plane, t);
        fft_pass<LOGN, 0, 0, C2, EXACT>(re, im, W, t);
    }
}

}  // namespace seamd
