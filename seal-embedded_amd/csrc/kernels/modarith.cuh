// modarith.cuh -- 32-bit residue arithmetic kept in registers (device inlines).
//
// Replaces the word layer of the reference: device/lib/modulo.h:21-116 (shift_result, Barrett
// 32->32 and 64->32), device/lib/uintmodarith.h:26-128 (add/neg/sub/mul mod) and the lazy
// Harvey/Shoup form of uintmodarith.h:293-346 (MUMO).  All results that leave a kernel are
// canonical residues in [0,q), i.e. equal to the reference's element for element; inside the
// NTT values float in [0,4q) (q < 2^30 for every prime of parameters.c:129-174).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace seamd {

// x in [0,2q) -> [0,q): unsigned min trick, 2 VALU ops, no compare/select.
__device__ __forceinline__ uint32_t csub(uint32_t x, uint32_t q)
{
    return min(x, x - q);
}

// Shoup product y*w mod q in [0,2q) for ANY 32-bit y, given wp = floor(w * 2^32 / q).
__device__ __forceinline__ uint32_t mul_shoup_lazy(uint32_t y, uint32_t w, uint32_t wp, uint32_t q)
{
    uint32_t h = __umulhi(y, wp);
    return y * w - h * q;
}

// x mod q for a full 32-bit x (modulo.h:43-75): one mul_hi with floor(2^32/q).
__device__ __forceinline__ uint32_t barrett32(uint32_t x, uint32_t q, uint32_t cr_hi)
{
    uint32_t est = __umulhi(x, cr_hi);
    return csub(x - est * q, q);
}

// The reduction of an accepted sampler word (sample.c:50-56 reduces with barrett32).  Every accepted
// word is below the rejection bound; for the 30-bit primes that bound is 4q - 1, so two conditional
// subtractions (4 full-rate VALU ops) give the same residue as the mul_hi/mul_lo form (R4 = true).
template <bool R4>
__device__ __forceinline__ uint32_t reduce_sample(uint32_t x, uint32_t q, uint32_t cr_hi)
{
    if constexpr (R4)
    {
        x = min(x, x - 2u * q);
        return min(x, x - q);
    }
    else
        return barrett32(x, q, cr_hi);
}

// x mod q for a 64-bit x (modulo.h:84-116), ratio = floor(2^64/q) as (hi,lo).
__device__ __forceinline__ uint32_t barrett64(uint64_t x, uint32_t q, uint32_t cr_hi, uint32_t cr_lo)
{
    uint64_t ratio = ((uint64_t)cr_hi << 32) | cr_lo;
    uint64_t est   = __umul64hi(x, ratio);
    uint32_t r     = (uint32_t)x - (uint32_t)est * q;  // true remainder is in [0,2q): 32 bits suffice
    return csub(r, q);
}

// signed 64-bit plaintext coefficient -> residue (ckks_common.c:224-237).  The reference maps a
// negative multiple of q to the non-canonical value q; the NTT that follows absorbs it
// (add_mod/sub_mod subtract q once), so feeding q here is equivalent -- we keep it identical.
__device__ __forceinline__ uint32_t reduce_signed(int64_t x, uint32_t q, uint32_t cr_hi, uint32_t cr_lo)
{
    uint64_t mag = x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x;
    uint32_t r   = barrett64(mag, q, cr_hi, cr_lo);
    return x < 0 ? q - r : r;
}

// Harvey butterfly, inputs/outputs in [0,4q): (X, Y) -> (X + Y*w, X - Y*w)  (ntt.c:156-162).
// `nw` is the NEGATED root (2^32 - w) and `wp` its Shoup companion floor(w 2^32 / q), so that
// tn = h*q + y*nw = -(y*w - h*q) = -t (mod 2^32) with t in [0,2q), and both outputs are single adds:
// 7 VALU ops: sub+min (x into [0,2q)), mul_hi, mul_lo, mad_u64_u32, sub, add3.
__device__ __forceinline__ void ct_butterfly(uint32_t &x, uint32_t &y, uint32_t nw, uint32_t wp,
                                             uint32_t q, uint32_t two_q)
{
    uint32_t u  = min(x, x - two_q);
    uint32_t h  = __umulhi(y, wp);
    uint32_t tn = (uint32_t)((uint64_t)h * (uint64_t)q + (uint64_t)(y * nw));  // -t mod 2^32
    x           = u - tn;
    y           = u + two_q + tn;
}

// 16 signed plaintext coefficients -> NTT inputs mod q.  The forward NTT is exact arithmetic mod q on
// operands anywhere in [0, 4q) and its result is canonicalised afterwards, so ANY representative of
// m mod q in that range yields the reference's canonical NTT(reduce_set_pte(m)) (the reference itself
// feeds the non-canonical value q for negative multiples, ckks_common.c:234).  `small` (wave-uniform, from
// the encoder) says that every |m + e| of the wave is below 2 q_min - 43: then m + 2q is such a
// representative -- ONE add per coefficient and prime instead of the ~10 operations of the signed Barrett
// reduction.  Otherwise all lanes take the exact path (reduce_pte_core, ckks_common.c:224-237).
__device__ __forceinline__ void reduce_signed16(const int64_t (&m)[16], uint32_t (&x)[16], uint32_t q,
                                                uint32_t cr_hi, uint32_t cr_lo, bool small)
{
    if (small)
    {
        const uint32_t two_q = q << 1;
#pragma unroll
        for (int e = 0; e < 16; e++) x[e] = (uint32_t)(int32_t)m[e] + two_q;   // in (0, 4q)
    }
    else
    {
#pragma unroll
        for (int e = 0; e < 16; e++) x[e] = reduce_signed(m[e], q, cr_hi, cr_lo);
    }
}

// the fast fused kernel's plaintext (int32: every coefficient small by construction)
__device__ __forceinline__ void reduce_signed16(const int32_t (&m)[16], uint32_t (&x)[16], uint32_t q, uint32_t,
                                                uint32_t, bool)
{
    const uint32_t two_q = q << 1;
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = (uint32_t)m[e] + two_q;   // in (0, 4q)
}

// Gentleman-Sande butterfly of the inverse NTT, inputs/outputs in [0,2q):
// (U, V) -> (U + V, (U - V) * w)   (intt.c:188-195)
__device__ __forceinline__ void gs_butterfly(uint32_t &x, uint32_t &y, uint32_t w, uint32_t wp,
                                             uint32_t neg_q, uint32_t two_q)
{
    uint32_t s = x + y;                 // [0,4q)
    uint32_t d = x + two_q - y;         // (0,4q)
    x          = min(s, s - two_q);     // [0,2q)
    uint32_t h = __umulhi(d, wp);
    y          = (uint32_t)((uint64_t)h * (uint64_t)neg_q + (uint64_t)(d * w));  // [0,2q)
}

// [0,4q) -> [0,q)
__device__ __forceinline__ uint32_t canon4(uint32_t x, uint32_t q, uint32_t two_q)
{
    return csub(min(x, x - two_q), q);
}

// Fused epilogues on a LAZY transform output x in [0,4q) (round 4: 11 / 10 operations per coefficient instead of
// 13-14 / 12 -- the canonicalisation of x, of the product and of the result collapse into one canon4):
//   x - y*w mod q:  u = x mod 2q in [0,2q);  p = y*w - h*q in [0,2q);  u + 2q - p in (0,4q)  -> canon4
//   x + y*w mod q:  u + p in [0,4q)                                                          -> canon4
// Same residues as  csub(canon4(x) + q - csub(p)),  csub(csub(p) + canon4(x))  (poly_*_mod_inpl,
// polymodarith.h:39-101): everything is exact arithmetic mod q below 2^32 (4q < 2^32).
__device__ __forceinline__ uint32_t sub_mul_canon(uint32_t x, uint32_t y, uint32_t w, uint32_t wp, uint32_t q,
                                                  uint32_t two_q)
{
    const uint32_t u = min(x, x - two_q);
    return canon4(u + two_q - mul_shoup_lazy(y, w, wp, q), q, two_q);
}
__device__ __forceinline__ uint32_t add_mul_canon(uint32_t x, uint32_t y, uint32_t w, uint32_t wp, uint32_t q,
                                                  uint32_t two_q)
{
    const uint32_t u = min(x, x - two_q);
    return canon4(u + mul_shoup_lazy(y, w, wp, q), q, two_q);
}

}  // namespace seamd
