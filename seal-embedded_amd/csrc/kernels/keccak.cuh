// keccak.cuh -- Keccak-f[1600] for one state per lane, written for gfx950 VALU.
//
// The 25 x 64-bit state lives in 50 VGPRs as (lo, hi) halves.  All 64-bit rotations are pairs of
// v_alignbit_b32, the 5-input column parities are two v_bitop3_b32 (xor3, LUT 0x96) each and
// chi  a ^ (~b & c)  is ONE v_bitop3_b32 (LUT 0xD2) -- about 190 VALU ops per round instead of
// the ~285 the compiler produces from 64-bit C (64-bit shifts are slow-rate on CDNA).
//
// Replaces: /root/reference/device/lib/shake256/keccakf1600.c:51-316 (KeccakF1600_StatePermute)
// and the absorb/squeeze framing of shake256/fips202.c:51-128 as used by rng.h:78-91
// (message = 64-byte seed || 8-byte little-endian counter, rate 136, domain byte 0x1F).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace seamd {

struct KeccakState
{
    uint32_t lo[25];
    uint32_t hi[25];
};

__device__ __constant__ const uint32_t kKeccakRC[24][2] = {
    {0x00000001u, 0x00000000u}, {0x00008082u, 0x00000000u}, {0x0000808au, 0x80000000u},
    {0x80008000u, 0x80000000u}, {0x0000808bu, 0x00000000u}, {0x80000001u, 0x00000000u},
    {0x80008081u, 0x80000000u}, {0x00008009u, 0x80000000u}, {0x0000008au, 0x00000000u},
    {0x00000088u, 0x00000000u}, {0x80008009u, 0x00000000u}, {0x8000000au, 0x00000000u},
    {0x8000808bu, 0x00000000u}, {0x0000008bu, 0x80000000u}, {0x00008089u, 0x80000000u},
    {0x00008003u, 0x80000000u}, {0x00008002u, 0x80000000u}, {0x00000080u, 0x80000000u},
    {0x0000800au, 0x00000000u}, {0x8000000au, 0x80000000u}, {0x80008081u, 0x80000000u},
    {0x00008080u, 0x80000000u}, {0x80000001u, 0x00000000u}, {0x80008008u, 0x80000000u}};

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// a ^ (~b & c); truth table over (a,b,c) with a the most significant selector bit.
__device__ __forceinline__ uint32_t chi3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2);
}

// 64-bit rotate-left of (lo,hi) by a compile-time amount, as two funnel shifts.
template <int R>
__device__ __forceinline__ void rol64(uint32_t lo, uint32_t hi, uint32_t &olo, uint32_t &ohi)
{
    if constexpr (R == 0)
    {
        olo = lo;
        ohi = hi;
    }
    else if constexpr (R == 32)
    {
        olo = hi;
        ohi = lo;
    }
    else if constexpr (R < 32)
    {
        olo = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
        ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - R);
    }
    else
    {
        olo = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
        ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - R);
    }
}

// theta + rho + pi for one source lane: B[DST] = rol(A[SRC] ^ D[SRC % 5], R)
// Two forms of theta.  FOLD = false forms D[x] = C[x-1] ^ rol1(C[x+1]) (10 v_xor) and applies it with
// 50 v_xor: 62 v_xor + 70 v_bitop3 + 58 v_alignbit = 190 ops.  FOLD = true never forms D: A ^ D is one
// xor3(A, C[x-1], rol1(C[x+1])): 2 v_xor + 120 v_bitop3 + 58 v_alignbit = 180 ops.  Measured on gfx950:
// the folded form is faster where several waves share a SIMD and the kernel is throughput-bound
// (k_sample_cbd 3.82 -> 3.39 ms, k_sample_ternary 1.18 -> 1.13 ms per 65 536), the unfolded form where
// one wave per SIMD runs a sequential chain (k_sample_uniform 6.52 vs 6.63 ms: v_bitop3 costs a lone
// wave ~5.3 cycles against 4 for v_xor).  Each kernel picks its form.
#define SEAMD_RHOPI(SRC, DST, R)                                                                      \
    if constexpr (FOLD)                                                                               \
        rol64<R>(xor3(s.lo[SRC], clo[((SRC) % 5 + 4) % 5], dlo[(SRC) % 5]),                           \
                 xor3(s.hi[SRC], chi_[((SRC) % 5 + 4) % 5], dhi[(SRC) % 5]), blo[DST], bhi[DST]);     \
    else                                                                                              \
        rol64<R>(s.lo[SRC] ^ dlo[(SRC) % 5], s.hi[SRC] ^ dhi[(SRC) % 5], blo[DST], bhi[DST])

template <bool FOLD = false>
__device__ __forceinline__ void keccak_round(KeccakState &s, uint32_t rclo, uint32_t rchi)
{
    uint32_t clo[5], chi_[5], dlo[5], dhi[5], blo[25], bhi[25];
#pragma unroll
    for (int x = 0; x < 5; x++)
    {
        clo[x]  = xor3(xor3(s.lo[x], s.lo[x + 5], s.lo[x + 10]), s.lo[x + 15], s.lo[x + 20]);
        chi_[x] = xor3(xor3(s.hi[x], s.hi[x + 5], s.hi[x + 10]), s.hi[x + 15], s.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++)
    {
        uint32_t rl, rh;
        rol64<1>(clo[(x + 1) % 5], chi_[(x + 1) % 5], rl, rh);
        if constexpr (FOLD)
        {
            dlo[x] = rl;
            dhi[x] = rh;
        }
        else
        {
            dlo[x] = clo[(x + 4) % 5] ^ rl;
            dhi[x] = chi_[(x + 4) % 5] ^ rh;
        }
    }
    // B[y][2x+3y] = rol(A[x][y], r[x][y]); lane index = x + 5y
    SEAMD_RHOPI(0, 0, 0);
    SEAMD_RHOPI(1, 10, 1);
    SEAMD_RHOPI(2, 20, 62);
    SEAMD_RHOPI(3, 5, 28);
    SEAMD_RHOPI(4, 15, 27);
    SEAMD_RHOPI(5, 16, 36);
    SEAMD_RHOPI(6, 1, 44);
    SEAMD_RHOPI(7, 11, 6);
    SEAMD_RHOPI(8, 21, 55);
    SEAMD_RHOPI(9, 6, 20);
    SEAMD_RHOPI(10, 7, 3);
    SEAMD_RHOPI(11, 17, 10);
    SEAMD_RHOPI(12, 2, 43);
    SEAMD_RHOPI(13, 12, 25);
    SEAMD_RHOPI(14, 22, 39);
    SEAMD_RHOPI(15, 23, 41);
    SEAMD_RHOPI(16, 8, 45);
    SEAMD_RHOPI(17, 18, 15);
    SEAMD_RHOPI(18, 3, 21);
    SEAMD_RHOPI(19, 13, 8);
    SEAMD_RHOPI(20, 14, 18);
    SEAMD_RHOPI(21, 24, 2);
    SEAMD_RHOPI(22, 9, 61);
    SEAMD_RHOPI(23, 19, 56);
    SEAMD_RHOPI(24, 4, 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5)
    {
#pragma unroll
        for (int x = 0; x < 5; x++)
        {
            s.lo[y + x] = chi3(blo[y + x], blo[y + (x + 1) % 5], blo[y + (x + 2) % 5]);
            s.hi[y + x] = chi3(bhi[y + x], bhi[y + (x + 1) % 5], bhi[y + (x + 2) % 5]);
        }
    }
    s.lo[0] ^= rclo;
    s.hi[0] ^= rchi;
}
#undef SEAMD_RHOPI

template <bool FOLD = false>
__device__ __forceinline__ void keccak_f1600(KeccakState &s)
{
#pragma unroll 2
    for (int r = 0; r < 24; r++) keccak_round<FOLD>(s, kKeccakRC[r][0], kKeccakRC[r][1]);
}

// Same permutation with the first and the last round peeled out of the loop.  For a freshly
// absorbed PRNG message most lanes of the input state are compile-time constants (lanes 10..15 and
// 17..24 are zero, lane 9 = 0x1F, lane 16 = 1<<63), so the peeled round 0 constant-folds (~40 of
// its 190 ops disappear); and when the caller only consumes part of the output (the first word
// for a redraw, 96 bytes for a CBD / ternary block) dead-code elimination prunes the peeled last
// round (~150 resp. ~70 ops).  Use right after prng_absorb().
template <bool FOLD = false>
__device__ __forceinline__ void keccak_f1600_fresh(KeccakState &s)
{
    keccak_round<FOLD>(s, kKeccakRC[0][0], kKeccakRC[0][1]);
#pragma unroll 2
    for (int r = 1; r < 23; r++) keccak_round<FOLD>(s, kKeccakRC[r][0], kKeccakRC[r][1]);
    keccak_round<FOLD>(s, kKeccakRC[23][0], kKeccakRC[23][1]);
}

// State after absorbing the 72-byte PRNG message seed[64] || le64(ctr) with SHAKE256 padding:
// lanes 0..7 = seed, lane 8 = counter, lane 9 = 0x1F, lane 16 = 0x80 << 56 (byte 135).
__device__ __forceinline__ void prng_absorb(KeccakState &s, const uint32_t (&seed)[16], uint64_t ctr)
{
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
        s.lo[i] = seed[2 * i];
        s.hi[i] = seed[2 * i + 1];
    }
    s.lo[8] = (uint32_t)ctr;
    s.hi[8] = (uint32_t)(ctr >> 32);
    s.lo[9] = 0x1Fu;
    s.hi[9] = 0;
#pragma unroll
    for (int i = 10; i < 25; i++)
    {
        s.lo[i] = 0;
        s.hi[i] = 0;
    }
    s.hi[16] = 0x80000000u;
}

}  // namespace seamd
