// transform8.cuh -- the two length-n transforms of transform.cuh with EIGHT points per thread.
//
// One workgroup of n/8 threads owns one polynomial; a pass runs three radix-2 stages in registers, the points
// are re-dealt through LDS between passes.  n = 4096: 4 passes over the windows 0 / 3 / 6 / 9 of the point
// index, 3 exchanges.  Against the 16-point form (3 passes, 2 exchanges) a plaintext costs +50 % LDS traffic,
// +25 % root loads and twice the waves per workgroup barrier -- and has half the live registers per thread:
// 8 waves per SIMD instead of 4.  Same butterflies, same root tables, same order of the
// floating-point operations per point as transform.cuh (re-dealing points between threads changes no
// arithmetic):
//   inverse FFT, DIF, (u, v) -> (u + v, (u - v) * W[h + j])            /root/reference/device/lib/fft.c:69-144
//   forward NTT, CT/Harvey, (u, v) -> (u + v*R[h + g], u - v*R[h + g]) /root/reference/device/lib/ntt.c:124-165
#pragma once
#include "transform.cuh"

namespace seamd {

template <int LOGN>
struct Xform8Geom
{
    static_assert(LOGN == 12, "four radix-8 passes: windows 0, 3, 6, 9 of a 12-bit index");
    static constexpr int N       = 1 << LOGN;
    static constexpr int THREADS = N / 8;
    static constexpr int SLOTS   = N + N / 8;   // padded LDS elements (largest lds8_slot + 1, rounded)
    static constexpr int CTOP    = LOGN - 3;
};

// point index held in slot e (0..7) of thread t for a pass whose 8-point tiles span bits [C, C+3)
template <int C>
__device__ __forceinline__ int tile8_index(int t, int e)
{
    return ((t >> C) << (C + 3)) | (e << C) | (t & ((1 << C) - 1));
}

// LDS slot of point k for an exchange between the tile layouts CA and CB: both deal patterns conflict-free under
// the lane-group model (tools/lds_conflicts.py --points 8: 0 extra cycles for ds_write/read_b32 and _b64 at
// every exchange of n = 4096) with every thread's 8 addresses base(t) + constant(e):
//   windows 0 and 3 : lanes walk strides 8 / 64  -> one pad element per 8        (k + (k >> 3))
//   windows 3 and 6 : 8 consecutive lanes, then a stride of 64 -> 8 pads per 64  (k + 8 (k >> 6))
//   windows 6 and 9 : 64 consecutive lanes hold 64 consecutive points -> no pad
template <int CA, int CB>
__host__ __device__ constexpr int lds8_slot(int k)
{
    constexpr int lo = CA < CB ? CA : CB, hi = CA < CB ? CB : CA;
    if constexpr (hi <= 3)
        return k + (k >> 3);
    else if constexpr (lo >= 6)
        return k;
    else
        return k + ((k >> 6) << 3);
}

// lds8_slot() is additive over disjoint bit fields, so a thread's 8 slots are one base + compile-time offsets
// (transform.cuh, redeal).  Write, barrier, read, barrier.
template <int C_FROM, int C_TO, typename T>
__device__ __forceinline__ void redeal8(T (&v)[8], T *lds, int t)
{
    T *wr = lds + lds8_slot<C_FROM, C_TO>(tile8_index<C_FROM>(t, 0));
    static_for<0, 8>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        wr[lds8_slot<C_FROM, C_TO>(e << C_FROM)] = v[e];
    });
    __syncthreads();
    const T *rd = lds + lds8_slot<C_FROM, C_TO>(tile8_index<C_TO>(t, 0));
    static_for<0, 8>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        v[e] = rd[lds8_slot<C_FROM, C_TO>(e << C_TO)];
    });
    __syncthreads();
}

// Root index of the butterflies of stage b (bit C + b of the point index) held by thread t: h + (k >> (C+b+1)),
// h = n >> (C+b+1).  Window 0 reads the thread-major copy behind the tables (se_types.h, xform8_offset): row
// (4 >> b) - 1 + g holds the entry of group g for t = 0 .. n/8 - 1, a wave load covers 64 consecutive entries.
template <int LOGN, int C, int B, int G>
__device__ __forceinline__ int root8_index(int t)
{
    constexpr int N = 1 << LOGN;
    if constexpr (C == 0)
        return (int)xform8_offset(N) + ((4 >> B) - 1 + G) * (N / 8) + t;
    else
    {
        // top window: t < n/8 = 2^C, every index is a compile-time constant (uniform address: scalar loads)
        const int thi = (C + 3 >= LOGN) ? 0 : (t >> C);
        return (N >> (C + B + 1)) + ((thi << (2 - B)) | G);
    }
}

// IFFT pass over window C: stages b = 0, 1, 2 ascending (fft.c:118-141, one round per stage)
template <int LOGN, int C>
__device__ __forceinline__ void ifft8_pass(double (&re)[8], double (&im)[8], const double *__restrict__ W, int t)
{
    static_for<0, 3>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        static_for<0, (4 >> b)>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const double2 w = *reinterpret_cast<const double2 *>(W + 2 * root8_index<LOGN, C, b, g>(t));
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int e0 = (g << (b + 1)) | decltype(rc)::value;
                constexpr int e1 = e0 | (1 << b);
                double ar = __dsub_rn(re[e0], re[e1]);
                double ai = __dsub_rn(im[e0], im[e1]);
                re[e0]    = __dadd_rn(re[e0], re[e1]);
                im[e0]    = __dadd_rn(im[e0], im[e1]);
                cmul_annexg<false>(ar, ai, w.x, w.y, re[e1], im[e1]);
            });
        });
    });
}

// First pass for REAL input (transform.cuh, ifft_pass0_real: position e of a tile is real before stage b iff
// its low b bits are zero -- 4 + 2 + 1 = 7 of the pass's 12 butterflies take the short form; the sign of a zero
// is the only thing that can differ from the general form and it never reaches the int64 plaintext).
template <int LOGN>
__device__ __forceinline__ void ifft8_pass0_real(double (&re)[8], double (&im)[8], const double *__restrict__ W,
                                                 int t)
{
    static_for<0, 3>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        static_for<0, (4 >> b)>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const double2 w = *reinterpret_cast<const double2 *>(W + 2 * root8_index<LOGN, 0, b, g>(t));
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

// Whole IFFT of a real input: tile layout 0 in (thread t holds points 8t .. 8t+7; im[] need not be initialised
// except im[0]), tile layout 9 out (thread t holds points t + (n/8) e).  `plane` = Xform8Geom::SLOTS doubles.
template <int LOGN>
__device__ __forceinline__ void ifft8_tiles_real(double (&re)[8], double (&im)[8], const double *__restrict__ W,
                                                 double *plane, int t)
{
    ifft8_pass0_real<LOGN>(re, im, W, t);
    redeal8<0, 3>(re, plane, t);
    redeal8<0, 3>(im, plane, t);
    ifft8_pass<LOGN, 3>(re, im, W, t);
    redeal8<3, 6>(re, plane, t);
    redeal8<3, 6>(im, plane, t);
    ifft8_pass<LOGN, 6>(re, im, W, t);
    redeal8<6, 9>(re, plane, t);
    redeal8<6, 9>(im, plane, t);
    ifft8_pass<LOGN, 9>(re, im, W, t);
}

// NTT pass over window C: stages b = 2, 1, 0 descending (ntt.c:140-164)
template <int LOGN, int C>
__device__ __forceinline__ void ntt8_pass(uint32_t (&x)[8], const uint32_t *__restrict__ RW, uint32_t q,
                                          uint32_t two_q, int t)
{
    static_for<0, 3>([&](auto sc) {
        constexpr int b = 2 - decltype(sc)::value;
        static_for<0, (4 >> b)>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const uint2 rw  = *reinterpret_cast<const uint2 *>(RW + 2 * root8_index<LOGN, C, b, g>(t));
            static_for<0, (1 << b)>([&](auto rc) {
                constexpr int e0 = (g << (b + 1)) | decltype(rc)::value;
                constexpr int e1 = e0 | (1 << b);
                ct_butterfly(x[e0], x[e1], rw.x, rw.y, q, two_q);
            });
        });
    });
}

// Whole NTT: tile layout 9 in, values anywhere in [0, 4q); tile layout 0 out (thread t holds the 8 consecutive
// coefficients 8t .. 8t+7 of the bit-reversed-order result), values in [0, 4q).
template <int LOGN>
__device__ __forceinline__ void ntt8_tiles(uint32_t (&x)[8], const uint32_t *__restrict__ RW, uint32_t q,
                                           uint32_t *lds, int t)
{
    const uint32_t two_q = q << 1;
    ntt8_pass<LOGN, 9>(x, RW, q, two_q, t);
    redeal8<9, 6>(x, lds, t);
    ntt8_pass<LOGN, 6>(x, RW, q, two_q, t);
    redeal8<6, 3>(x, lds, t);
    ntt8_pass<LOGN, 3>(x, RW, q, two_q, t);
    redeal8<3, 0>(x, lds, t);
    ntt8_pass<LOGN, 0>(x, RW, q, two_q, t);
}

// Tile layout 0 -> quad layout, wave-local (transform.cuh, tile_to_quads): a wave's 64 tiles are the 512
// consecutive coefficients 512 w .. 512 w + 511; after the transpose lane l holds, in slot group i = 0, 1,
// coefficients 512 w + 256 i + 4 l + (0..3), so every global instruction of the epilogue covers 1 KiB
// contiguous.  Unpadded rows of 8 words: the reads are linear (conflict-free), the two writes 2-way.
__device__ __forceinline__ void tile8_to_quads(uint32_t (&x)[8], uint32_t *lds_region, int t)
{
    const int lane = t & 63;
    uint32_t *w    = lds_region + (t >> 6) * 512;
    uint4 *row     = reinterpret_cast<uint4 *>(w + 8 * lane);
    row[0]         = make_uint4(x[0], x[1], x[2], x[3]);
    row[1]         = make_uint4(x[4], x[5], x[6], x[7]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 2; i++)
    {
        const uint4 v = *reinterpret_cast<const uint4 *>(w + 256 * i + 4 * lane);
        x[4 * i] = v.x, x[4 * i + 1] = v.y, x[4 * i + 2] = v.z, x[4 * i + 3] = v.w;
    }
    // the next prime's transpose writes this chunk again: keep those writes behind these reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// coefficient index (within the polynomial) of slot 4 i of thread t in the 8-point quad layout
__device__ __forceinline__ int quad8_index(int t, int i)
{
    return ((t >> 6) << 9) + (i << 8) + ((t & 63) << 2);
}

}  // namespace seamd
