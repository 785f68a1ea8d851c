// stage_ops.hip -- stand-alone, explicit-operand kernels behind the reference-named LOWER surface
// (se_lower.cpp): every operand comes from the caller instead of the context's key / table state,
// so the reference's per-prime calling sequence (device/test/ckks_tests_sym.c:103-172,
// device/bench/bench_sym.c:76-143) maps onto one launch per call.
//
//   k_fft_polys         ifft_inpl / fft_inpl                 /root/reference/device/lib/fft.c:69-213
//                       + the rounding tail of ckks_encode_base   ckks_common.c:183-206
//   k_reduce_poly       reduce_set_pte / reduce_add_pte /
//                       reduce_set_e_small / reduce_add_e_small   ckks_common.c:224-274
//   k_expand_ternary    expand_poly_ternary                      sample.c:98-129
//   k_lower_sym_prime   ckks_encode_encrypt_sym after `a`        ckks_sym.c:240-300
//   k_lower_asym_prime  ckks_encode_encrypt_asym                 ckks_asym.c:205-286
//
// Same transforms as the batched hot path (transform.cuh), one workgroup of n/16 threads per
// polynomial.  These are latency-path kernels (batch of one per call); the throughput path is
// encode_encrypt.hip / samplers.hip.
#include <hip/hip_runtime.h>

#include "../se_types.h"
#include "kernel_args.h"
#include "transform.cuh"

namespace seamd {

namespace {

__device__ __forceinline__ uint32_t mulmod64(uint32_t a, uint32_t b, uint32_t q, uint32_t crh, uint32_t crl)
{
    return barrett64((uint64_t)a * (uint64_t)b, q, crh, crl);  // uintmodarith.h:123-128
}

__device__ __forceinline__ uint32_t addmod(uint32_t a, uint32_t b, uint32_t q)
{
    return csub(a + b, q);  // uintmodarith.h:26-47 (operands < q)
}

// 2-bit code of coefficient k, MSB-first within a byte (sample.c:89-96), expanded (sample.c:98-111)
__device__ __forceinline__ uint32_t expand_code(const uint8_t *packed, uint32_t k, uint32_t q)
{
    const uint32_t v = (packed[k >> 2] >> (6 - 2 * (k & 3))) & 3u;
    return v + (v == 0 ? q : 0u) - 1u;
}

__device__ __forceinline__ void store16u(uint32_t *p, const uint32_t (&v)[16])
{
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) p4[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

__device__ __forceinline__ void load16u(uint32_t (&v)[16], const uint32_t *p)
{
    const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        uint4 w = p4[i];
        v[4 * i] = w.x, v[4 * i + 1] = w.y, v[4 * i + 2] = w.z, v[4 * i + 3] = w.w;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// FFT family on interleaved complex128 polynomials [count][n][2].
//   mode 0: ifft_inpl   (DIF, no 1/n; fft.c:69-144)            in -> out_cplx
//   mode 1: ifft_inpl + round(Re * scale/n) -> int64 (ckks_common.c:183-206); out_cplx (optional)
//           receives the IFFT output as well; fail_idx[b] = first index whose |coeff| > 2^63
//           (0xFFFFFFFF when none), which is where the reference returns false
//   mode 2: fft_inpl    (DIT; fft.c:146-213)                    in -> out_cplx
// ------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS) void k_fft_polys(DevParams P, DevTables T, FftArgs A)
{
    using G            = XformGeom<LOGN>;
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *plane    = reinterpret_cast<double *>(smem);
    const int t      = threadIdx.x;
    const size_t b   = blockIdx.x;
    const double *in = A.in + b * 2 * N;
    double re[16], im[16];
    if (A.mode == 2)
    {
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            const double2 v = *reinterpret_cast<const double2 *>(in + 2 * ((e << CTOP) + t));
            re[e] = v.x, im[e] = v.y;
        }
        fft_tiles<LOGN, true>(re, im, T.ifft_w, plane, t);   // EXACT: the caller may pass NaN / Inf
        double *out = A.out_cplx + b * 2 * N;
#pragma unroll
        for (int e = 0; e < 16; e++)
            *reinterpret_cast<double2 *>(out + 2 * (16 * t + e)) = make_double2(re[e], im[e]);
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const double2 v = *reinterpret_cast<const double2 *>(in + 2 * (16 * t + e));
        re[e] = v.x, im[e] = v.y;
    }
    ifft_tiles<LOGN, false, true>(re, im, T.ifft_w, plane, t);   // EXACT product (transform.cuh, cmul_annexg)
    if (A.out_cplx)
    {
        double *out = A.out_cplx + b * 2 * N;
#pragma unroll
        for (int e = 0; e < 16; e++)
            *reinterpret_cast<double2 *>(out + 2 * ((e << CTOP) + t)) = make_double2(re[e], im[e]);
    }
    if (A.mode != 1) return;
    uint32_t first_bad = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 15; e >= 0; e--)
    {
        const double c = round(__dmul_rn(re[e], P.n_inv));
        if (fabs(c) > 9223372036854775808.0) first_bad = (uint32_t)((e << CTOP) + t);
        // (int64_t) of NaN and of 2^63 is the x86-64 "integer indefinite" value in the reference's build
        A.out_int[b * N + (e << CTOP) + t] = (fabs(c) < 9223372036854775808.0) ? (int64_t)c : INT64_MIN;
    }
    if (first_bad != 0xFFFFFFFFu) atomicMin(A.fail_idx + b, first_bad);
}

template <int LOGN>
static hipError_t launch_fft_n(const DevParams &P, const DevTables &T, const FftArgs &A, size_t count,
                               hipStream_t st)
{
    using G      = XformGeom<LOGN>;
    size_t shmem = (size_t)G::SLOTS * sizeof(double);
    (void)hipFuncSetAttribute((const void *)k_fft_polys<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)shmem);
    hipLaunchKernelGGL((k_fft_polys<LOGN>), dim3((unsigned)count), dim3(G::THREADS), shmem, st, P, T, A);
    return hipGetLastError();
}

hipError_t launch_fft_polys(const DevParams &P, const DevTables &T, const FftArgs &A, size_t count,
                            hipStream_t st)
{
    if (count == 0) return hipSuccess;
    switch (P.logn)
    {
        case 10: return launch_fft_n<10>(P, T, A, count, st);
        case 11: return launch_fft_n<11>(P, T, A, count, st);
        case 12: return launch_fft_n<12>(P, T, A, count, st);
        case 13: return launch_fft_n<13>(P, T, A, count, st);
        case 14: return launch_fft_n<14>(P, T, A, count, st);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// RNS reduction of one polynomial set for prime j, element-wise.
// ------------------------------------------------------------------------------------------
__global__ void k_reduce_poly(DevParams P, int j, const int64_t *pte, const int8_t *e, uint32_t *out,
                              int add, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t q = P.q[j];
    uint32_t r;
    if (pte)
        r = reduce_signed(pte[i], q, P.cr_hi[j], P.cr_lo[j]);
    else
    {
        const int32_t v = e[i];
        r               = (v < 0 ? q : 0u) + (uint32_t)v;
    }
    if (add)
    {
        // add_mod_inpl (uintmodarith.h:26-47): one conditional subtraction of q
        const uint32_t s = out[i] + r;
        r                = s - (s >= q ? q : 0u);
    }
    out[i] = r;
}

hipError_t launch_reduce_poly(const DevParams &P, int j, const int64_t *pte, const int8_t *e, uint32_t *out,
                              bool add, size_t total, hipStream_t st)
{
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_reduce_poly, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, j, pte, e,
                       out, add ? 1 : 0, total);
    return hipGetLastError();
}

// sample_add_poly_cbd_generic_inpl_prng_16's "+=" (sample.c:347-356): m[i] += e[i]
__global__ void k_add_small(int64_t *m, const int8_t *e, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) m[i] += e[i];
}

hipError_t launch_add_small(int64_t *m, const int8_t *e, size_t total, hipStream_t st)
{
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_small, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, m, e, total);
    return hipGetLastError();
}

// set_small_poly_idx packing (sample.c:61-87): four 2-bit codes per byte, first coefficient in the
// two MOST significant bits
__global__ void k_pack_ternary(const int8_t *codes, uint8_t *packed, size_t total_bytes)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_bytes) return;
    const uint32_t w = *reinterpret_cast<const uint32_t *>(codes + 4 * i);
    packed[i] = (uint8_t)(((w & 3u) << 6) | (((w >> 8) & 3u) << 4) | (((w >> 16) & 3u) << 2) | ((w >> 24) & 3u));
}

hipError_t launch_pack_ternary(const int8_t *codes, uint8_t *packed, size_t total_bytes, hipStream_t st)
{
    if (total_bytes == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_ternary, dim3((unsigned)((total_bytes + 255) / 256)), dim3(256), 0, st, codes,
                       packed, total_bytes);
    return hipGetLastError();
}

// convert_poly_ternary (sample.c:138-148): entries > 1 (i.e. q_prev - 1) become q - 1
// sample_poly_ternary's word mapping (sample.c:176-186): w -> (w mod 3) + (w == 0 ? q : 0) - 1 in 32-bit
// arithmetic, literally as the reference computes it; words >= 0xFFFFFFFE are flagged for a redraw
__global__ void k_ternary_words(const uint32_t *in, uint32_t *out, uint32_t *nrej, uint32_t q, uint32_t n, int op)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t w = in[k];
    if (op == 0)
        out[k] = w > 1 ? q - 1 : w;
    else
    {
        if (w >= 0xFFFFFFFEu) atomicAdd(nrej, 1u);
        out[k] = (w % 3u) + (w == 0 ? q : 0u) - 1u;
    }
}

hipError_t launch_ternary_words(const uint32_t *in, uint32_t *out, uint32_t *nrej, uint32_t q, uint32_t n, int op,
                                hipStream_t st)
{
    hipLaunchKernelGGL(k_ternary_words, dim3((n + 255) / 256), dim3(256), 0, st, in, out, nrej, q, n, op);
    return hipGetLastError();
}

__global__ void k_expand_ternary(const uint8_t *packed, uint32_t *out, uint32_t q, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = expand_code(packed, k, q);
}

hipError_t launch_expand_ternary(const uint8_t *packed, uint32_t *out, uint32_t q, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_expand_ternary, dim3((n + 255) / 256), dim3(256), 0, st, packed, out, q, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// ckks_encode_encrypt_sym for ONE prime with explicit operands (ckks_sym.c:240-300):
//   c0 = NTT(expand(s_small));  s_save = c0;  c0 = -(c0 . a);
//   ntt_pte = NTT(reduce(pte) or reduce(ep));  c0 += ntt_pte
// ------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS) void k_lower_sym_prime(DevParams P, DevTables T,
                                                                           LowerSymArgs A)
{
    using G            = XformGeom<LOGN>;
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds32  = reinterpret_cast<uint32_t *>(smem);
    const int t      = threadIdx.x;
    const size_t b   = blockIdx.x;
    const int j      = A.j;
    const uint32_t q = P.q[j], two_q = q << 1, crh = P.cr_hi[j], crl = P.cr_lo[j];
    const uint32_t *RW = T.ntt_rw + 2 * xform_table_len(N) * j;
    const size_t off   = b * N + 16 * t;
    const uint8_t *key = A.s_small + b * (size_t)A.s_stride;
    const uint32_t *ap = A.a + b * (size_t)(A.a_stride ? A.a_stride : N) + 16 * t;
    uint32_t *c0p      = A.c0 + b * (size_t)(A.c0_stride ? A.c0_stride : N) + 16 * t;

    uint32_t x[16], c0[16];
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = expand_code(key, (uint32_t)((e << CTOP) + t), q);
    ntt_tiles<LOGN>(x, RW, q, lds32, t);
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = canon4(x[e], q, two_q);
    if (A.s_save) store16u(A.s_save + off, x);
    {
        uint32_t a[16];
        load16u(a, ap);
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            const uint32_t pr = mulmod64(x[e], a[e], q, crh, crl);
            c0[e]             = pr ? q - pr : 0u;  // poly_neg_mod_inpl (uintmodarith.h:49-70)
        }
    }
    if (A.pte)
    {
#pragma unroll
        for (int e = 0; e < 16; e++) x[e] = reduce_signed(A.pte[b * N + (e << CTOP) + t], q, crh, crl);
    }
    else
    {
#pragma unroll
        for (int e = 0; e < 16; e++)
        {
            const int32_t v = A.ep[b * N + (e << CTOP) + t];
            x[e]            = (v < 0 ? q : 0u) + (uint32_t)v;
        }
    }
    ntt_tiles<LOGN>(x, RW, q, lds32, t);
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        x[e]  = canon4(x[e], q, two_q);
        c0[e] = addmod(c0[e], x[e], q);
    }
    // the caller's ntt_pte and c1 buffers may be the same memory (ckks_sym.c:86-88): write order is
    // the host wrapper's business, the kernel only produces the two polynomials
    store16u(A.ntt_pte + off, x);
    store16u(c0p, c0);
}

// ------------------------------------------------------------------------------------------
// ckks_encode_encrypt_asym for ONE prime with explicit operands (ckks_asym.c:235-284):
//   u_hat = NTT(expand(u));  c1 = pk1 . u_hat + NTT(e1);  c0 = pk0 . u_hat + NTT(reduce(pte))
// ------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(XformGeom<LOGN>::THREADS) void k_lower_asym_prime(DevParams P, DevTables T,
                                                                            LowerAsymArgs A)
{
    using G            = XformGeom<LOGN>;
    constexpr int N    = G::N;
    constexpr int CTOP = LOGN - 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds32  = reinterpret_cast<uint32_t *>(smem);
    const int t      = threadIdx.x;
    const size_t b   = blockIdx.x;
    const int j      = A.j;
    const uint32_t q = P.q[j], two_q = q << 1, crh = P.cr_hi[j], crl = P.cr_lo[j];
    const uint32_t *RW = T.ntt_rw + 2 * xform_table_len(N) * j;
    const size_t off   = b * N + 16 * t;

    uint32_t uh[16], x[16], p1[16], p0[16];
#pragma unroll
    for (int e = 0; e < 16; e++) uh[e] = expand_code(A.u_small + b * (N / 4), (uint32_t)((e << CTOP) + t), q);
    ntt_tiles<LOGN>(uh, RW, q, lds32, t);
#pragma unroll
    for (int e = 0; e < 16; e++) uh[e] = canon4(uh[e], q, two_q);
    if (A.ntt_u_save) store16u(A.ntt_u_save + off, uh);
    load16u(p1, A.pk1 + off);
    load16u(p0, A.pk0 + off);
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        p1[e] = mulmod64(p1[e], uh[e], q, crh, crl);
        p0[e] = mulmod64(p0[e], uh[e], q, crh, crl);
    }
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const int32_t v = A.e1[b * N + (e << CTOP) + t];
        x[e]            = (v < 0 ? q : 0u) + (uint32_t)v;
    }
    ntt_tiles<LOGN>(x, RW, q, lds32, t);
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        x[e]  = canon4(x[e], q, two_q);
        p1[e] = addmod(p1[e], x[e], q);
    }
    if (A.ntt_e1_save) store16u(A.ntt_e1_save + off, x);
    store16u(A.c1 + off, p1);
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = reduce_signed(A.pte[b * N + (e << CTOP) + t], q, crh, crl);
    ntt_tiles<LOGN>(x, RW, q, lds32, t);
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        x[e]  = canon4(x[e], q, two_q);
        p0[e] = addmod(p0[e], x[e], q);
    }
    store16u(A.ntt_pte + off, x);
    store16u(A.c0 + off, p0);
}

// ------------------------------------------------------------------------------------------
// Word-arithmetic inlines of modarith.cuh, one operation per element, for known-answer tests that
// push the reference's own edge vectors (device/test/modulo_tests.c:78-179,
// uintmodarith_tests.c:96-192) through the DEVICE code paths the kernels use.
//   0 barrett32(a)            1 barrett64(a)              2 mul_mod via 64-bit Barrett (a*b)
//   3 Shoup product a*b: csub(mul_shoup_lazy(a, b, floor(b 2^32/q)))   (a any 32-bit, b < q)
//   4 add_mod = csub(a + b)   5 neg_mod                   6 sub_mod
//   7 reduce_signed((int64)a)                             8 canon4(a)   (a < 4q)
//   9 / 10 Harvey butterfly (a, b) with root w = c: outputs x' / y' canonical
//  11 / 12 Gentleman-Sande butterfly (a, b) with root c: outputs x' / y' canonical
// ------------------------------------------------------------------------------------------
__global__ void k_word_ops(DevParams P, int j, int op, const uint64_t *a, const uint64_t *b,
                           const uint64_t *c, uint32_t *out, size_t count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t q = P.q[j], crh = P.cr_hi[j], crl = P.cr_lo[j], two_q = q << 1;
    const uint64_t A = a[i], B = b ? b[i] : 0, Cc = c ? c[i] : 0;
    uint32_t r = 0;
    switch (op)
    {
        case 0: r = barrett32((uint32_t)A, q, crh); break;
        case 1: r = barrett64(A, q, crh, crl); break;
        case 2: r = barrett64((uint64_t)(uint32_t)A * (uint64_t)(uint32_t)B, q, crh, crl); break;
        case 3:
        {
            const uint32_t w = (uint32_t)B, wp = (uint32_t)(((uint64_t)w << 32) / q);
            r = csub(mul_shoup_lazy((uint32_t)A, w, wp, q), q);
            break;
        }
        case 4: r = csub((uint32_t)A + (uint32_t)B, q); break;
        case 5: r = (uint32_t)A ? q - (uint32_t)A : 0u; break;
        case 6: r = csub((uint32_t)A + q - (uint32_t)B, q); break;
        case 7: r = reduce_signed((int64_t)A, q, crh, crl); break;
        case 8: r = canon4((uint32_t)A, q, two_q); break;
        case 9:
        case 10:
        {
            const uint32_t w = (uint32_t)Cc, wp = (uint32_t)(((uint64_t)w << 32) / q);
            uint32_t x = (uint32_t)A, y = (uint32_t)B;
            ct_butterfly(x, y, 0u - w, wp, q, two_q);
            r = canon4(op == 9 ? x : y, q, two_q);
            break;
        }
        case 11:
        case 12:
        {
            const uint32_t w = (uint32_t)Cc, wp = (uint32_t)(((uint64_t)w << 32) / q);
            uint32_t x = (uint32_t)A, y = (uint32_t)B;
            gs_butterfly(x, y, w, wp, 0u - q, two_q);
            r = csub(op == 11 ? x : y, q);
            break;
        }
        default: break;
    }
    out[i] = r;
}

hipError_t launch_word_ops(const DevParams &P, int j, int op, const uint64_t *a, const uint64_t *b,
                           const uint64_t *c, uint32_t *out, size_t count, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_word_ops, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, P, j, op, a, b, c,
                       out, count);
    return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_lower_n(const DevParams &P, const DevTables &T, const LowerSymArgs *S,
                                 const LowerAsymArgs *Y, size_t count, hipStream_t st)
{
    using G      = XformGeom<LOGN>;
    size_t shmem = (size_t)G::SLOTS * sizeof(uint32_t);
    if (S)
    {
        (void)hipFuncSetAttribute((const void *)k_lower_sym_prime<LOGN>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL((k_lower_sym_prime<LOGN>), dim3((unsigned)count), dim3(G::THREADS), shmem, st, P,
                           T, *S);
    }
    else
    {
        (void)hipFuncSetAttribute((const void *)k_lower_asym_prime<LOGN>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL((k_lower_asym_prime<LOGN>), dim3((unsigned)count), dim3(G::THREADS), shmem, st,
                           P, T, *Y);
    }
    return hipGetLastError();
}

static hipError_t launch_lower(const DevParams &P, const DevTables &T, const LowerSymArgs *S,
                               const LowerAsymArgs *Y, size_t count, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    switch (P.logn)
    {
        case 10: return launch_lower_n<10>(P, T, S, Y, count, st);
        case 11: return launch_lower_n<11>(P, T, S, Y, count, st);
        case 12: return launch_lower_n<12>(P, T, S, Y, count, st);
        case 13: return launch_lower_n<13>(P, T, S, Y, count, st);
        case 14: return launch_lower_n<14>(P, T, S, Y, count, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_lower_sym_prime(const DevParams &P, const DevTables &T, const LowerSymArgs &A, size_t count,
                                  hipStream_t st)
{
    return launch_lower(P, T, &A, nullptr, count, st);
}

hipError_t launch_lower_asym_prime(const DevParams &P, const DevTables &T, const LowerAsymArgs &A,
                                   size_t count, hipStream_t st)
{
    return launch_lower(P, T, nullptr, &A, count, st);
}

}  // namespace seamd
