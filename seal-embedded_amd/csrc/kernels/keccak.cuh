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

#include <type_traits>

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

// ------------------------------------------------------------------------------------------
// Pair-cooperative form: ONE state over TWO adjacent lanes (2k, 2k + 1).
//
// A chain kernel's ciphertext is a sequential sponge squeeze (121 permutations per prime at n = 4096, 482 at
// n = 16384): with one state per lane its latency is 24 x 190 instructions of ONE wave, whatever the batch
// size, and a batch smaller than the chip leaves SIMDs empty.  Here the even lane of a pair holds the LOW
// 32-bit halves of the 25 state lanes, the odd lane the HIGH halves (25 VGPRs each).  theta's parities, the
// application of D, chi and iota are half-local; only a 64-bit rotation needs the partner's half of the same
// state lane: one v_mov_b32_dpp quad_perm:[1,0,3,2] (a VALU move -- no LDS, no barrier) per rotated word, and
// the SAME instruction sequence serves both lanes:
//     R < 32 : lo' = lo << R | hi >> (32-R),  hi' = hi << R | lo >> (32-R)   = alignbit(own, partner, 32 - R)
//     R > 32 : lo' = hi << (R-32) | lo >> (64-R),  hi' likewise              = alignbit(partner, own, 64 - R)
// Per round and lane: 10 (parities) + 5 + 5 (rol1 C) + 25 (folded theta) + 24 + 24 (rho) + 25 (chi) + 3 (iota) = 121
// VALU instructions against 190 for the lane-per-state form: the chain of one ciphertext gets 1.5x shorter,
// the chip does 1.33x the work -- a win exactly where the lane-per-state form leaves SIMDs idle (small
// batches, single calls), a loss on a full chip.  Bit-identical by construction: the
// same Boolean function of keccakf1600.c:51-316, other registers.
// ------------------------------------------------------------------------------------------
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void static_for_keccak(F &&f)
{
    if constexpr (BEGIN < END)
    {
        f(std::integral_constant<int, BEGIN>{});
        static_for_keccak<BEGIN + 1, END>(f);
    }
}

struct KeccakHalf
{
    uint32_t w[25];   // even lane: low halves, odd lane: high halves
};

__device__ __forceinline__ uint32_t pair_swap(uint32_t v)
{
    // quad_perm:[1,0,3,2]: every lane reads its pair partner
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

// one half of rol64(own : partner, R)
template <int R>
__device__ __forceinline__ uint32_t rol64_half(uint32_t own, uint32_t partner)
{
    static_assert(R > 0 && R < 64 && R != 32, "the rho offsets of Keccak-f[1600] other than 0");
    if constexpr (R < 32)
        return __builtin_amdgcn_alignbit(own, partner, 32 - R);
    else
        return __builtin_amdgcn_alignbit(partner, own, 64 - R);
}

// rc = this lane's half of the round constant.  theta is folded (A ^ D = xor3(A, C[x-1], rol1(C[x+1])): no D),
// and rho runs in three sweeps -- all the theta outputs, all the partner fetches, all the funnel shifts -- so
// that every v_mov_b32_dpp has independent instructions between it and the write of its source (a DPP read
// needs two wait states after a VALU write: 8 s_nop per round in the straightforward order).
__device__ __forceinline__ void keccak_half_round(KeccakHalf &s, uint32_t rc)
{
    constexpr int kSrcOfDst[25] = {0, 6, 12, 18, 24, 3, 9, 10, 16, 22, 1, 7, 13, 19, 20, 4, 5, 11, 17, 23, 2, 8, 14, 15, 21};
    constexpr int kRho[25]      = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                                   25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};
    uint32_t c[5], rl[5], t[25], p[25], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = xor3(xor3(s.w[x], s.w[x + 5], s.w[x + 10]), s.w[x + 15], s.w[x + 20]);
    uint32_t cp[5];
#pragma unroll
    for (int x = 0; x < 5; x++) cp[x] = pair_swap(c[x]);
#pragma unroll
    for (int x = 0; x < 5; x++) rl[x] = rol64_half<1>(c[x], cp[x]);   // this lane's half of rol1(C[x])
#pragma unroll
    for (int i = 0; i < 25; i++) t[i] = xor3(s.w[i], c[(i % 5 + 4) % 5], rl[(i % 5 + 1) % 5]);
#pragma unroll
    for (int i = 1; i < 25; i++) p[i] = pair_swap(t[i]);
    // B[dst] = rol(t[src], rho[src]);  dst = y + 5 ((2x + 3y) mod 5) for src = x + 5y
    static_for_keccak<0, 25>([&](auto dc) {
        constexpr int dst = decltype(dc)::value;
        constexpr int src = kSrcOfDst[dst];
        constexpr int R   = kRho[src];
        if constexpr (R == 0)
            b[dst] = t[src];
        else
            b[dst] = rol64_half<(R == 0 ? 1 : R)>(t[src], p[src]);
    });
#pragma unroll
    for (int y = 0; y < 25; y += 5)
    {
#pragma unroll
        for (int x = 0; x < 5; x++) s.w[y + x] = chi3(b[y + x], b[y + (x + 1) % 5], b[y + (x + 2) % 5]);
    }
    s.w[0] ^= rc;
}

// `part` = 0 on the even lane (low halves), 1 on the odd lane (high halves)
__device__ __forceinline__ void keccak_half_f1600(KeccakHalf &s, uint32_t part)
{
#pragma unroll 2
    for (int r = 0; r < 24; r++) keccak_half_round(s, part ? kKeccakRC[r][1] : kKeccakRC[r][0]);
}

// this lane's half of the state prng_absorb() builds (seed[64] || le64(ctr), SHAKE256 padding)
__device__ __forceinline__ void prng_absorb_half(KeccakHalf &s, const uint32_t (&seed)[16], uint64_t ctr,
                                                 uint32_t part)
{
#pragma unroll
    for (int i = 0; i < 8; i++) s.w[i] = part ? seed[2 * i + 1] : seed[2 * i];
    s.w[8] = part ? (uint32_t)(ctr >> 32) : (uint32_t)ctr;
    s.w[9] = part ? 0u : 0x1Fu;
#pragma unroll
    for (int i = 10; i < 25; i++) s.w[i] = 0;
    s.w[16] = part ? 0x80000000u : 0u;
}

// ------------------------------------------------------------------------------------------
// Wave-cooperative form: ONE state over the 64 lanes of a wave -- the latency form for a handful of chains.
//
// State lane (x, y) lives in wave lane 8 y + x + 1 ("primary"), as (lo, hi) in two VGPRs: plane y occupies an
// 8-lane group, two planes per 16-lane DPP row, planes 0..4 = lanes 0..39, the rest of the wave holds zeros.
// Each group is [4' 0 1 2 3 4 0' 1']: lane 0 is a copy of column 4, lanes 6, 7 copies of columns 0, 1, so that
// the in-plane neighbours x-1, x+1, x+2 of every primary lane are plain DPP row shifts (no wrap handling
// except for column 4, whose right-hand copies are stale after chi).  Per round:
//   theta  parity over the planes: v_xor_dpp row_ror:8 (the two planes of a row), then v_permlane16_swap and
//          v_permlane32_swap (gfx950) fold the rows of BOTH halves together: 9 instructions, no LDS;
//          D and A ^= D: 2 v_alignbit + 6 bank- / row-masked v_xor_dpp (wave_keccak_round)
//   rho    one 64-bit rotation by a per-lane amount: 2 v_cndmask (halves swapped for R >= 32) + 2 v_alignbit
//   pi     one ds_bpermute_b32 per half (a fixed permutation; it also refreshes the copy lanes and re-zeroes
//          lanes 40..47)
//   chi    B[x+1], B[x+2] by row_shl:1 / row_shl:2, one v_bitop3 per half;  iota on the lane of (0, 0), skipped for
//          a zero half of the constant
// A lone wave pays one issue slot (~4.5 cycles at best) for EVERY instruction, scalar ones and s_nop included, so the
// count that matters is all of them: 38.6 per round (28.6 VALU, 2 DS, 2 s_waitcnt, 6 s_nop for the DPP / permlane
// read-after-write wait states), straight-line with the round constants as literals.  Round 5's form was 38 VALU
// + 2 DS + 6 s_nop + 9 scalar (constant fetch through s_getpc / s_load, loop, waits) = 55, and its s_waitcnt also sat
// out the scalar load; round 6 measured each step inside one call (profiles/r06_ab_wave_keccak.log): joint parity
// 55 -> 47 (-10 % single-call uniform sampler), unrolled with literal constants -> 42 (-15 %), masked-DPP theta
// -> 38.6 (-20 %: 476 -> 377 us for the 121 permutations of
// sample_poly_uniform at n = 4096, ~3.1 us per permutation incl. its emit; the lane form takes ~9-10 us as a lone
// wave).  13x the work per state of the lane form -- for launches of at most a few waves per SIMD (a single
// se_encrypt call, batches of up to ~2 000 ciphertexts incl. the virtual ones of the prime speculation).
// ------------------------------------------------------------------------------------------
struct WaveKeccak
{
    uint32_t lo, hi;      // this lane's state lane (or copy, or zero)
    // per-lane constants (set up once per kernel by wave_keccak_init)
    uint32_t sh;          // rho: v_alignbit amount
    uint32_t pi_addr;     // pi: 4 * source lane for ds_bpermute
    uint32_t iota;        // all-ones on the primary lane of state lane (0, 0)
    uint64_t swap;        // rho: lane mask of the lanes whose rotation amount is >= 32 (or == 0): halves swapped first
    int index;            // x + 5 y on primary lanes of planes 0..4, -1 elsewhere
};

__device__ __constant__ const uint8_t kKeccakRho[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                                                        25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};

__device__ __forceinline__ void wave_keccak_init(WaveKeccak &k, int lane)
{
    const int g = lane & 7, y = lane >> 3;
    const int x = (g + 4) % 5;
    const bool in_state = y < 5;
    const bool primary  = in_state && g >= 1 && g <= 5;
    k.index = primary ? x + 5 * y : -1;
    k.iota  = (lane == 1) ? 0xFFFFFFFFu : 0u;
    const int R = in_state ? kKeccakRho[x + 5 * y] : 0;
    // rol64 by R on (lo, hi): for R in 1..31 lo' = alignbit(lo, hi, 32 - R), hi' = alignbit(hi, lo, 32 - R);
    // for R >= 32 the same on swapped halves with R - 32; R == 0 is "swapped, amount 0" (alignbit by 0 yields
    // its second source)
    k.swap = __builtin_amdgcn_ballot_w64((R >= 32) || (R == 0));
    k.sh   = (uint32_t)((32 - (R & 31)) & 31);
    // pi: B[x'][y'] = rol(A[x][y]) with x' = y, y' = 2x + 3y  =>  this lane (x', y') pulls from the primary lane of
    // x = 3 y' + x' (mod 5), y = x'
    const int xs = (3 * y + x) % 5, ys = x;
    // (lanes 40..47 pull the always-zero lane 63: theta leaves the column parities in them, see wave_keccak_round)
    k.pi_addr = in_state ? (uint32_t)(4 * (8 * ys + xs + 1)) : (uint32_t)(4 * (lane < 48 ? 63 : lane));
    k.lo = k.hi = 0;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// xor over the five planes of every column, valid on every lane of planes 0..4 (and beyond), for BOTH halves at once
// (round 6): the row-folding steps work on a register PAIR, so the low and the high
// half share them -- permlane16_swap(tl, th) leaves [L0 H0 L2 H2] / [L1 H1 L3 H3] (rows), their xor has the low half's
// partial parities in the even rows and the high half's in the odd rows; one permlane32_swap folds the wave halves of
// both; a last permlane16_swap of the result with itself deals the even rows (C_lo) to every row of one register and
// the odd rows (C_hi) to every row of the other.  9 instructions instead of 14.
__device__ __forceinline__ void wave_column_parity2(uint32_t lo, uint32_t hi, uint32_t &clo, uint32_t &chi_)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const uint32_t tl = lo ^ dpp_row<0x128>(lo), th = hi ^ dpp_row<0x128>(hi);   // row_ror:8: the two planes of a row
    u2 r              = __builtin_amdgcn_permlane16_swap(tl, th, false, false);  // {L0 H0 L2 H2 ; L1 H1 L3 H3}
    const uint32_t z  = r.x ^ r.y;                                               // L01 H01 L23 H23
    r                 = __builtin_amdgcn_permlane32_swap(z, z, false, false);    // {lower, lower ; upper, upper}
    const uint32_t w  = r.x ^ r.y;                                               // Clo Chi Clo Chi
    r                 = __builtin_amdgcn_permlane16_swap(w, w, false, false);    // {Clo x 4 ; Chi x 4}
    clo  = r.x;
    chi_ = r.y;
}

// Keccak round constant r from the degree-8 LFSR of FIPS 202 (3.2.5), at compile time: the rounds of the wave form
// are instantiated with their constants (below), so nothing is loaded and a zero half costs nothing.
constexpr uint64_t keccak_round_constant(int r)
{
    uint64_t rc = 0;
    uint32_t R  = 1;
    for (int i = 0; i < 7 * r; i++) R = ((R << 1) ^ ((R >> 7) * 0x71u)) & 0xFFu;
    for (int j = 0; j < 7; j++)
    {
        if (R & 1u) rc |= 1ull << ((1 << j) - 1);
        R = ((R << 1) ^ ((R >> 7) * 0x71u)) & 0xFFu;
    }
    return rc;
}

static_assert(keccak_round_constant(0) == 0x0000000000000001ull && keccak_round_constant(2) == 0x800000000000808aull &&
                  keccak_round_constant(12) == 0x000000008000808bull &&
                  keccak_round_constant(23) == 0x8000000080008008ull,
              "round constants (kKeccakRC rows 0, 2, 12, 23)");

template <uint32_t RCLO, uint32_t RCHI>
__device__ __forceinline__ void wave_keccak_round(WaveKeccak &k)
{
    // theta
    uint32_t clo, chi_;
    wave_column_parity2(k.lo, k.hi, clo, chi_);
    // D[x] = C[x-1] ^ rol1(C[x+1]) and A ^= D in eight instructions.  With G[g] = C[c(g)] ^ rol1(C[c(g)+2]) on group
    // lanes g = 0..4 (columns c = 4, 0, 1, 2, 3), D of the primary lane g is G[g-1].  C[c+2] sits two lanes to the
    // right for g = 0..3 and three lanes to the LEFT for g = 4 (the copy lanes right of column 4 are stale after
    // chi): the row_shr:3 form is written to every lane, then the row_shl:2 form overwrites DPP banks 0 and 2
    // (g = 0..3 of both groups of a row; bank_mask 0x5) -- g = 4 keeps the first, g = 5..7 are never read.  The
    // apply is a v_xor_b32_dpp in place with row_mask 0x7: row 3 (lanes 48..63) is not written and stays zero, and
    // lanes 40..47, which do pick up C, are re-zeroed by pi (they pull lane 63) before anything reads them.  The
    // compiler forms neither masked DPP, hence the asm; it does not see the DPP operands in there either, so the
    // two wait states a DPP read needs after a VALU write of the same register are placed by hand (s_nop 0 + the
    // other half's instruction).
    // rho rides in the same block (one 64-bit rotation by a per-lane amount: halves swapped under the wave-uniform
    // lane mask k.swap, then two v_alignbit): with it outside, the compiler puts one more s_nop behind the block.
    uint32_t elo, ehi, glo, ghi, rlo, rhi;
    asm("v_alignbit_b32 %2, %8, %9, 31\n\t"
        "v_alignbit_b32 %3, %9, %8, 31\n\t"
        "s_nop 0\n\t"
        "v_xor_b32_dpp %4, %2, %8 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_xor_b32_dpp %5, %3, %9 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_xor_b32_dpp %4, %2, %8 row_shl:2 row_mask:0xf bank_mask:0x5\n\t"
        "v_xor_b32_dpp %5, %3, %9 row_shl:2 row_mask:0xf bank_mask:0x5\n\t"
        "s_nop 0\n\t"
        "v_xor_b32_dpp %0, %4, %0 row_shr:1 row_mask:0x7 bank_mask:0xf\n\t"
        "v_xor_b32_dpp %1, %5, %1 row_shr:1 row_mask:0x7 bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %2, %0, %1, %10\n\t"   // a = swap ? hi : lo
        "v_cndmask_b32_e64 %3, %1, %0, %10\n\t"   // b = swap ? lo : hi
        "v_alignbit_b32 %6, %2, %3, %11\n\t"
        "v_alignbit_b32 %7, %3, %2, %11"
        : "+v"(k.lo), "+v"(k.hi), "=&v"(elo), "=&v"(ehi), "=&v"(glo), "=&v"(ghi), "=&v"(rlo), "=&v"(rhi)
        : "v"(clo), "v"(chi_), "s"(k.swap), "v"(k.sh));
    // pi
    const uint32_t blo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)k.pi_addr, (int)rlo);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)k.pi_addr, (int)rhi);
    // chi, iota
    k.lo = chi3(blo, dpp_row<0x101>(blo), dpp_row<0x102>(blo));
    k.hi = chi3(bhi, dpp_row<0x101>(bhi), dpp_row<0x102>(bhi));
    if constexpr (RCLO != 0) k.lo = __builtin_amdgcn_bitop3_b32(k.lo, k.iota, RCLO, 0x78);
    if constexpr (RCHI != 0) k.hi = __builtin_amdgcn_bitop3_b32(k.hi, k.iota, RCHI, 0x78);
}

// All 24 rounds straight-line.  The rolled form (two rounds per iteration, constants fetched from kKeccakRC) spent 12
// of its 47 instructions per round on scalar work -- s_getpc / s_add / s_addc / s_load for the constant, the loop
// counter and branch -- and a lone wave pays a full issue slot (~4.5 cycles) for every one of them.
template <int R = 0>
__device__ __forceinline__ void wave_keccak_f1600(WaveKeccak &k)
{
    if constexpr (R < 24)
    {
        constexpr uint64_t rc = keccak_round_constant(R);
        wave_keccak_round<(uint32_t)rc, (uint32_t)(rc >> 32)>(k);
        wave_keccak_f1600<R + 1>(k);
    }
}

// The state prng_absorb() builds (seed[64] || le64(ctr), SHAKE256 padding), spread over the wave: `seed`
// points at the 64 seed bytes in global memory.
__device__ __forceinline__ void wave_prng_absorb(WaveKeccak &k, const uint8_t *seed, uint64_t ctr, int lane)
{
    const int g = lane & 7, y = lane >> 3;
    const int i = (y < 5) ? ((g + 4) % 5) + 5 * y : 25;   // state lane this wave lane holds (copies included)
    uint32_t lo = 0, hi = 0;
    if (i < 8)
    {
        const uint2 v = *reinterpret_cast<const uint2 *>(seed + 8 * i);
        lo = v.x, hi = v.y;
    }
    else if (i == 8)
        lo = (uint32_t)ctr, hi = (uint32_t)(ctr >> 32);
    else if (i == 9)
        lo = 0x1Fu;
    else if (i == 16)
        hi = 0x80000000u;
    k.lo = lo, k.hi = hi;
}

}  // namespace seamd
