// se_types.h -- plain-old-data shared by the host code and the gfx950 kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace seamd {

constexpr int kMaxPrimes  = 13;  // parameters.c:159-171
constexpr int kSeedBytes  = 64;  // defines.h:67 (SE_PRNG_SEED_BYTE_COUNT)
constexpr int kShakeRate  = 136;

// Passed BY VALUE to every kernel (lands in SGPRs / kernarg).
struct DevParams
{
    uint32_t n;
    uint32_t logn;
    uint32_t nprimes;
    uint32_t q[kMaxPrimes];        // modulus chain
    uint32_t cr_hi[kMaxPrimes];    // floor(2^64/q) >> 32  == floor(2^32/q)
    uint32_t cr_lo[kMaxPrimes];    // floor(2^64/q) & 0xffffffff
    uint32_t bound[kMaxPrimes];    // uniform rejection bound (sample.c:46)
    double n_inv;                  // scale / n (ckks_common.c:183)
    uint32_t inv_n[kMaxPrimes];    // n^-1 mod q            (intt.c:230-420 constants)
    uint32_t inv_n_sh[kMaxPrimes]; // floor(inv_n * 2^32 / q)
    double scale;                  // CKKS scale (decode divides by it)
    uint32_t num_cus;              // compute units of the device (launch geometry of the chain kernels)
    double small_bound;            // 2 min_j q_j - 64: below it every |m + e| of a plaintext lies in (-2 q_j, 2 q_j)
};

// Device-resident read-only tables (pointers into one HBM slab owned by the context).
struct DevTables
{
    const uint16_t *inv_map;   // [n]   x[k] <- values[inv_map[k] & (n/2-1)]
    const double *ifft_w;      // [n][2] (re, im) of W[t], t = h + j  (fft.c:129), then the thread-major copy of
                               // the window-0 entries [15][n/16][2] (xform_table_len)
    const uint32_t *ntt_rw;    // [np][ (-root mod 2^32, shoup(root)) indexed h + g (ntt.c:40-52): [n][2], then the
                               // thread-major copy of the window-0 entries [15][n/16][2] ]
    const uint32_t *s_hat;     // [np][n][2] (NTT(s), shoup)       sym
    const uint32_t *pk0;       // [np][n][2] (pk0, shoup)          asym
    const uint32_t *pk1;       // [np][n][2] (pk1, shoup)          asym
    const uint32_t *intt_rw;   // [np][n][2] (psi^-bitrev(h+g), shoup) indexed h + g (intt.c:26-58)
    const uint16_t *index_map; // [n] forward index map (decode slot pick)
    const uint16_t *gather_map; // encoder gather: LDS position (sv_slot) of the value that feeds point 16 t + e,
                                // stored [e / 8][t][e % 8] (one uint4 per thread and half)
};

// Root tables carry a second, thread-major copy of the entries a pass over window 0 reads (transform.cuh:
// thread t needs entry h_b + (t << (3 - b)) + g for the stages b = 3..0, g < 2^(3-b), i.e. 8 / 4 / 2 / 1
// CONSECUTIVE entries per lane -- 64 different cache lines per wave load).  Row e = (8 >> b) - 1 + g of the
// copy holds that entry for t = 0 .. n/16-1, so a wave load reads 64 consecutive entries.
// Elements (pairs) of one table: n natural + 15 n/16 transposed.
constexpr size_t xform_table_len(size_t n)
{
    return n + 15 * (n / 16);
}

// LDS position of values[i] in the encoder's staging array.  Thread t gathers, for each e, the value
// feeding point 16t + e; over the 32 lanes of an LDS lane group those indices are i0 + 2^(logn-10) * h
// with 32 distinct h (the index map is a discrete log base 3 modulo 2n, and the lanes step the argument
// by a multiple of 2^(logn-8)), i.e. only 32 / 2^(logn-10) banks of a linear array: a 4-way conflict at
// n = 4096, 16-way at n = 16384.  Rotating the index right by logn-10 bits puts h into the bank bits.
constexpr uint32_t sv_slot(uint32_t i, uint32_t logn)
{
    const uint32_t s = logn - 10;
    return ((i >> s) & 31u) | ((i & ((1u << s) - 1u)) << 5) | ((i >> (s + 5)) << (s + 5));
}


enum Mode : int
{
    kModeSym        = 0,  // c1 = a, c0 = -a.s + NTT(m+e)
    kModeAsym       = 1,  // c1 = pk1.u + NTT(e1), c0 = pk0.u + NTT(m+e0)
    kModeEncodeOnly = 2,  // out = NTT(m mod q_j), no sampling (BASELINE config 5)
};

}  // namespace seamd
