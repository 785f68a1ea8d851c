// samplers.hip -- SHAKE256-expanded polynomial samplers, one Keccak state per lane.
//
// Replaces (batched):
//   prng_fill_buffer                        /root/reference/device/lib/rng.h:78-91
//   sample_poly_uniform                     /root/reference/device/lib/sample.c:39-57
//   sample_small_poly_ternary_prng_96       /root/reference/device/lib/sample.c:218-242
//   sample_poly_cbd_generic_prng_16 and
//   sample_add_poly_cbd_generic_inpl_prng_16 /root/reference/device/lib/sample.c:311-356
//
// The PRNG stream semantics are reproduced exactly, including the data-dependent counters:
//   * uniform: block(ctr) of 4n bytes, then ONE fresh 4-byte block per rejection draw, consumed
//     in coefficient order; the next prime continues at the next free counter.  Because every
//     redraw is an independent SHAKE call, "the k-th rejected coefficient takes the k-th accepted
//     candidate of the stream V[c] = block(c)[0:4], c = ctr+1, ctr+2, .." -- that is what the
//     second phase computes.
//   * ternary: 96-byte block per 96 coefficients, 1-byte redraw blocks interleaved between blocks.
//   * CBD: counters are static (base + k), fully parallel.
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <type_traits>

#include "../se_types.h"
#include "kernel_args.h"
#include "keccak.cuh"
#include "keccak_sync.cuh"
#include "modarith.cuh"

namespace seamd {

__device__ __forceinline__ void load_seed(uint32_t (&seed)[16], const uint8_t *seeds, size_t b)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(seeds + b * kSeedBytes);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        uint4 v         = p[i];
        seed[4 * i]     = v.x;
        seed[4 * i + 1] = v.y;
        seed[4 * i + 2] = v.z;
        seed[4 * i + 3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------
// Uniform `a` for all primes of one ciphertext per lane (the chain of counters is sequential per
// ciphertext: prime j+1 starts where prime j's redraws stopped).
//
// Phase 1 (bulk block): 4n bytes are squeezed 136 at a time; each 32-bit word is reduced mod q
// in registers (or replaced by a marker when it fails the rejection bound) and stored straight
// to the lane's own polynomial (16-byte pieces).  64 lanes write 64 different polynomials, so a
// store instruction touches 64 lines -- but every lane completes a 128-byte line within about
// one permutation, the partially written lines sit in L2 (64 x 128 B per wave) and reach HBM
// as full lines.  (A cooperative LDS-transposed store was measured slower: +3x on the bulk step.)
// The emit is branch-free (one wave per SIMD pays an issue slot for every scalar/branch
// instruction): reject bits go into per-lane masks that become reject-list entries once per step.
// Phase 2 (redraws): the k-th rejected coefficient takes the k-th accepted candidate of the
// stream block(ctr+1)[0:4], block(ctr+2)[0:4], ...  Candidates are independent SHAKE calls, so the
// phase is balanced: over the wave (lane slots dealt round-robin to the lanes that still need
// draws), over the workgroup when helper waves are present (small batches), and with helper waves
// precomputing candidates while the chains squeeze.  Rejected positions live in a per-ciphertext
// list in HBM scratch (rej_cap entries); beyond that the lane rescans its own output for the
// marker word.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kRejMarker = 0xFFFFFFFFu;  // >= every modulus, never a valid residue

// Workgroup size of the one-permutation-per-thread kernels (k_sample_cbd, k_candidates): 512 threads are two waves per
// SIMD and workgroup, which the phase-synchronised permutation of keccak_sync.cuh keeps in phase.
constexpr int kCbdThreads = 512;

// MAXT: the largest workgroup the instantiation is launched with.  Up to 8 waves per workgroup (every
// batch <= 4 x 64 x CUs, i.e. all BASELINE shapes) the kernel may use 256 VGPRs: 160, no spills; the
// 1024-thread form is capped at 128 and spills 140 B per lane (uniform stage 6.28 -> 6.11 ms at C2).
// LANE_PRIME: UniformArgs::prime_of is honoured (one prime per ciphertext, chosen per lane) -- an instantiation
// of its own so that the per-lane modulus constants cost the batch kernels nothing (166 vs 160 VGPRs, +1.5 %
// on the dominant kernel of C2 when it was a run-time branch).
template <int LOGN, int MAXT, bool LANE_PRIME = false>
__global__ __launch_bounds__(MAXT) void k_sample_uniform(DevParams P, UniformArgs A)
{
    constexpr int N          = 1 << LOGN;
    constexpr int FULL_STEPS = (N * 4) / 136;            // permutations that yield 34 words
    constexpr int TAIL_WORDS = N - FULL_STEPS * 34;      // words taken from one more permutation

    // dynamic LDS: reserved (> 80 KiB) to pin one workgroup per CU; its first bytes hold the
    // per-wave rank -> lane table of the balanced redraw phase
    extern __shared__ __attribute__((aligned(16))) unsigned char pin_lds[];
    const int lane      = threadIdx.x & 63;
    const int wave      = threadIdx.x >> 6;
    uint8_t *rank2lane  = pin_lds + wave * 64;

    // Waves [0, master_waves) of the workgroup own one ciphertext per lane; any further waves are
    // helpers that only take part in the redraw phase (small batches leave SIMDs empty otherwise).
    const uint32_t mthreads = A.master_waves * 64;
    const bool master = threadIdx.x < mthreads;
    const size_t bq   = (size_t)blockIdx.x * mthreads + threadIdx.x;
    // lanes past the batch (or masked out of a redo launch) stay alive as redraw helpers
    const bool active = master && bq < A.B &&
                        (!A.only_from || (A.only_from[bq] != 0 && A.prime_lo >= A.only_from[bq]));
    if (A.only_from && !__syncthreads_or(active)) return;  // redo launch with nothing to redo here
    const size_t b    = active ? bq : (size_t)A.B - 1;
    const bool wg_pool = blockDim.x > mthreads;   // helper waves present: pool the whole workgroup

    // workgroup-pool scratch (only used when wg_pool): carved from the reserved LDS
    uint32_t *lds_seed  = reinterpret_cast<uint32_t *>(pin_lds + 1024);            // [mthreads][16]
    uint64_t *lds_ctr   = reinterpret_cast<uint64_t *>(lds_seed + (size_t)mthreads * 16);
    uint32_t *lds_cand  = reinterpret_cast<uint32_t *>(lds_ctr + mthreads);          // [blockDim]
    uint16_t *lds_table = reinterpret_cast<uint16_t *>(lds_cand + blockDim.x);       // [mthreads]
    uint32_t *lds_cnt   = reinterpret_cast<uint32_t *>(lds_table + mthreads);        // [16]

    uint32_t seed[16];
    load_seed(seed, A.seeds, b);
    uint64_t ctr     = A.ctr_in ? A.ctr_in[b] : 0;
    uint32_t *mylist = A.rej_list + b * A.rej_cap;
    if (wg_pool && master)
    {
#pragma unroll
        for (int i = 0; i < 16; i++) lds_seed[threadIdx.x * 16 + i] = seed[i];
    }

    // prime_of: one prime per ciphertext, chosen per LANE (the modulus constants become lane values; the loop
    // runs once for the whole workgroup, so its barriers stay uniform)
    const uint32_t j_first = LANE_PRIME ? 0u : A.prime_lo, j_end = LANE_PRIME ? 1u : A.prime_hi;
    for (uint32_t jl = j_first; jl < j_end; jl++)
    {
        const uint32_t j = LANE_PRIME ? (uint32_t)A.prime_of[b] : jl;
        const uint32_t q = P.q[j], crh = P.cr_hi[j], bound = P.bound[j];
        uint32_t *mypoly = A.out + (LANE_PRIME ? b : b * A.out_primes + (j - A.out_prime_base)) * (size_t)N;

        uint32_t nrej = 0;
        const uint64_t bulk_ctr = ctr;   // the 4n-byte block; redraw candidates follow at ctr + 1 ..
        ctr++;
        const bool speculate = wg_pool && A.spec;
        if (speculate)
        {
            // Helper waves know every candidate counter up front (the bulk block consumes exactly
            // one), so while the masters squeeze they precompute spec_cap candidates per ciphertext
            // into HBM scratch; the masters then only walk that list.
            if (master) lds_ctr[threadIdx.x] = ctr;
            __syncthreads();
            if (!master)
            {
                const uint32_t hthreads = blockDim.x - mthreads;
                const size_t ct0        = (size_t)blockIdx.x * mthreads;
                for (uint32_t idx = threadIdx.x - mthreads; idx < mthreads * A.spec_cap; idx += hthreads)
                {
                    const uint32_t ctl = idx % mthreads, off = idx / mthreads;
                    if (ct0 + ctl >= A.B) continue;
                    uint32_t tseed[16];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        uint4 v = *reinterpret_cast<const uint4 *>(lds_seed + ctl * 16 + 4 * i);
                        tseed[4 * i] = v.x, tseed[4 * i + 1] = v.y, tseed[4 * i + 2] = v.z, tseed[4 * i + 3] = v.w;
                    }
                    KeccakState cs;
                    prng_absorb(cs, tseed, lds_ctr[ctl] + off);
                    keccak_f1600_fresh(cs);
                    A.spec[(ct0 + ctl) * A.spec_cap + off] = cs.lo[0];
                }
            }
        }
        // ---- phase 0 (full batches, no helper waves): the first K0 redraw candidates up front ---
        // The candidate counters are known before the bulk block is squeezed (it consumes exactly
        // one), so every lane computes its first K0 candidates V[ctr .. ctr+K0) into LDS now, and the
        // bulk phase patches a rejected word right after the 16-byte piece holding its marker was
        // stored -- the line is still in L2, so the patch costs no HBM write of its own (patching
        // after the whole polynomial has been written costs a 128-byte line per 4-byte patch: 1.64x
        // write amplification).  K0 = mean - 1.5 sigma of the per-polynomial reject count (capped
        // by LDS), so nearly every lane uses all of them; what is still rejected afterwards goes
        // through the reject list and the balanced phase 2 as before.  The result does not depend
        // on K0 (a candidate is only *consumed* when the lane needs one, in counter order).
        uint32_t k0 = 0, cpos = 0;
        uint32_t *lds_c0 = reinterpret_cast<uint32_t *>(pin_lds + 1024);  // [k0][blockDim.x]
        if (!wg_pool)
        {
            const float mean = (float)N * ((float)(0u - bound) * (1.0f / 4294967296.0f));
            const float lo   = mean - 1.5f * sqrtf(mean);
            k0               = lo > 0.0f ? (uint32_t)lo : 0u;
            k0               = min(k0, min(64u, (uint32_t)((83u * 1024u) / (4u * blockDim.x))));
            for (uint32_t c = 0; c < k0; c++)
            {
                KeccakState cs;
                prng_absorb(cs, seed, ctr + c);
                keccak_f1600_fresh(cs);  // only cs.lo[0] is consumed
                lds_c0[c * blockDim.x + threadIdx.x] = cs.lo[0];
            }
        }

        if (active)
        {
            KeccakState st;
            prng_absorb(st, seed, bulk_ctr);

            // Reject test and reduction of one word (sample.c:50-56), branch-free: a lone wave per
            // SIMD pays an issue slot for EVERY instruction, scalar and branch ones included, and
            // with 64 lanes some lane rejects at ~70 % of the word positions, so a per-word
            // `if (reject)` costs its mask/branch scaffolding 34 times per squeeze step.  Here each
            // word is reduce + compare + select-marker + shift the reject bit into a per-lane
            // mask; the masks are turned into patches / reject-list entries once per GROUP of steps (flush_group).
            // (r4: every accepted word is below 4q, see reduce_sample; a rejected word's value is not used)
            auto word = [&](auto r4, uint32_t x, uint32_t &mask) -> uint32_t {
                const bool rej   = x >= bound;
                const uint32_t r = reduce_sample<decltype(r4)::value>(x, q, crh);
                mask             = (mask << 1) | (rej ? 1u : 0u);
                return rej ? kRejMarker : r;
            };
            const bool red4 = !LANE_PRIME && (uint64_t)bound <= 4ull * q;   // uniform per prime
            // The reject bits of kFlushGroup = 3 consecutive squeeze steps (102 words) are collected in a 128-bit string
            // (mh : ml), the group's first word at the top of its `nbits` valid bits, and turned into patches / list
            // entries ONCE per group: the loop below runs max-over-lanes(rejects of the lane) times, ~5.7 per group of 102
            // words against 3 x (3.1 + 0.9) when every step flushed its 32- and its 2-bit mask separately (round 6:
            // the bookkeeping was 5 % of the chain, profiles/r06_ab_emit_diet.log).  Positions are handled in ascending
            // order; a patch lands at most three steps (~30 us) behind the store of its marker.
            // `cnext` is the lane's next phase-0 candidate, fetched from LDS one consumption ahead so that the loop
            // never waits for the read.
            uint64_t mh = 0, ml = 0;
            uint32_t cnext = k0 ? lds_c0[threadIdx.x] : 0u;
            // One candidate per pass and lane, no inner loop (a lone wave pays for every scalar / branch instruction of
            // nested divergent control flow): a lane whose candidate is itself rejected (1.9 %) keeps its bit and
            // takes the next candidate in the next pass.  The patch and the list append are ONE predicated store.
            auto flush_group = [&](uint32_t nbits, uint32_t first_pos) {
                while (__any((mh | ml) != 0))
                {
                    if ((mh | ml) != 0)
                    {
                        const uint32_t c   = mh ? (uint32_t)__clzll((long long)mh) : 64u + (uint32_t)__clzll((long long)ml);
                        const uint32_t pos = first_pos + (c - (128u - nbits));
                        const bool have    = cpos < k0;          // a phase-0 candidate is left
                        const uint32_t x   = cnext;
                        const bool acc     = have && x < bound;  // ... and accepted: patch in place
                        const bool listed  = !have;              // none left: the position goes to the reject list
                        cpos += have ? 1u : 0u;
                        ctr += have ? 1u : 0u;
                        cnext = lds_c0[min(cpos, k0 - 1u) * blockDim.x + threadIdx.x];   // (k0 = 0: never consumed)
                        uint32_t *dst      = acc ? mypoly + pos : mylist + nrej;
                        const uint32_t val = acc ? barrett32(x, q, crh) : pos;
                        if (acc || (listed && nrej < A.rej_cap)) *dst = val;
                        nrej += listed ? 1u : 0u;
                        if (acc || listed)
                        {
                            const uint64_t bit = 0x8000000000000000ull >> (c & 63u);
                            mh &= c < 64u ? ~bit : ~0ull;
                            ml &= c < 64u ? ~0ull : ~bit;
                        }
                    }
                }
            };
            constexpr uint32_t kFlushGroup = 3;

            uint32_t idx = 0, gbase = 0, gsteps = 0;   // first word / steps of the group being collected
            for (int step = 0; step < FULL_STEPS; step++)
            {
                keccak_f1600(st);
                uint32_t m0 = 0, m1 = 0;  // words 0..31 / 32..33 of this step
                auto emit = [&](auto r4) {
#pragma unroll
                    for (int i = 0; i < 17; i++)
                    {
                        uint32_t &mk = (i < 16) ? m0 : m1;
                        uint32_t w0  = word(r4, st.lo[i], mk);
                        uint32_t w1  = word(r4, st.hi[i], mk);
                        *reinterpret_cast<uint2 *>(mypoly + idx + 2 * i) = make_uint2(w0, w1);
                    }
                };
                if (red4)
                    emit(std::true_type{});
                else
                    emit(std::false_type{});
                // (mh : ml) <<= 34, the step's 34 bits appended (word 0 of the step first)
                mh = (mh << 34) | (ml >> 30);
                ml = (ml << 34) | ((uint64_t)m0 << 2) | (uint64_t)m1;
                idx += 34;
                if (++gsteps == kFlushGroup)
                {
                    flush_group(34u * kFlushGroup, gbase);
                    gsteps = 0;
                    gbase  = idx;
                }
            }
            if constexpr (TAIL_WORDS > 0)
            {
                static_assert(TAIL_WORDS <= 32 && TAIL_WORDS % 2 == 0, "tail fits one mask");
                static_assert(34 * (kFlushGroup - 1) + TAIL_WORDS <= 128, "an unfinished group and the tail share the string");
                keccak_f1600(st);
                uint32_t m0 = 0;
#pragma unroll
                for (int i = 0; i < TAIL_WORDS / 2; i++)
                {
                    uint32_t w0 = word(std::false_type{}, st.lo[i], m0);
                    uint32_t w1 = word(std::false_type{}, st.hi[i], m0);
                    *reinterpret_cast<uint2 *>(mypoly + idx + 2 * i) = make_uint2(w0, w1);
                }
                mh = (mh << TAIL_WORDS) | (ml >> (64 - TAIL_WORDS));
                ml = (ml << TAIL_WORDS) | (uint64_t)m0;
                flush_group(34u * gsteps + (uint32_t)TAIL_WORDS, gbase);
            }
            else if (gsteps)
                flush_group(34u * gsteps, gbase);
        }
        // bulk stores (and list entries) must have landed before phase 2 patches / reads them
        __builtin_amdgcn_s_waitcnt(0);
        __threadfence_block();

        // ---- phase 2: redraws, balanced over the wave ----------------------------------------
        // The k-th rejected coefficient takes the k-th accepted candidate of the stream
        // block(ctr)[0:4], block(ctr+1)[0:4], ... ; a draw is consumed (counter advanced) only
        // while the ciphertext still needs one.  Candidates are independent SHAKE calls, so lanes
        // that are done compute candidates for lanes that are not: each round the 64 lane slots
        // are dealt round-robin over the R needy lanes (slot s -> needy lane s % R, counter offset
        // s / R) and every needy lane then consumes its candidates in counter order.  The wave
        // finishes in ~ceil(total draws / 64) rounds instead of max-over-lanes draws.
        uint32_t need    = nrej;
        uint32_t k       = 0;  // rejected coefficients resolved so far
        uint32_t scanpos = 0;  // list-overflow path: next index to scan for a marker

        // one accepted candidate x for this lane's k-th rejected coefficient.  The position comes
        // from the reject list; `hint` is that entry when the caller already fetched it (the list
        // read is a dependent global load, so callers issue it ahead of the work that hides it).
        auto list_entry = [&](uint32_t kk) -> uint32_t {
            return __hip_atomic_load(mylist + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto place = [&](uint32_t x, bool have_hint, uint32_t hint) {
            uint32_t pos;
            if (k < A.rej_cap)
            {
                pos = have_hint ? hint : list_entry(k);
            }
            else
            {
                // list overflow: rejected positions are exactly the marker words
                pos = scanpos;
                while (__hip_atomic_load(mypoly + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                       kRejMarker)
                    pos++;
            }
            scanpos     = pos + 1;
            mypoly[pos] = barrett32(x, q, crh);
            k++;
            need--;
        };
        // entry k of the list, fetched ahead (0 when there is none to fetch)
        auto prefetch_entry = [&]() -> uint32_t {
            return (need > 0 && k < A.rej_cap) ? list_entry(k) : 0u;
        };

        if (speculate)
        {
            __syncthreads();  // helpers' candidates are in HBM/L2 (their vmcnt drained above)
            const uint32_t *row = A.spec + b * (size_t)A.spec_cap;
            uint32_t t = 0;
            // four candidates and the next four list entries per round trip: the walk is a chain
            // of dependent L2/HBM loads otherwise (~300 candidates per polynomial at n = 16384)
            for (; t + 4 <= A.spec_cap && need > 0; t += 4)
            {
                uint32_t x[4], pos4[4];
                const uint32_t k0 = k;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    x[i] = __hip_atomic_load(row + t + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    pos4[i] = (k0 + i < A.rej_cap && (uint32_t)i < need) ? list_entry(k0 + i) : 0u;
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    if (need > 0)
                    {
                        ctr++;
                        if (x[i] < bound)
                        {
                            const uint32_t a = k - k0;  // accepted so far in this group
                            place(x[i], true, a == 0 ? pos4[0] : a == 1 ? pos4[1] : a == 2 ? pos4[2] : pos4[3]);
                        }
                    }
                }
            }
            for (; t < A.spec_cap && need > 0; t++)
            {
                const uint32_t x = __hip_atomic_load(row + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ctr++;
                if (x < bound) place(x, false, 0u);
            }
        }

        // Workgroup pool (helper waves present): same dealing scheme as below, but over all
        // 64*(masters+helpers) lanes of the workgroup, with seeds / counters / candidates passed
        // through LDS and block barriers.
        while (wg_pool)
        {
            const uint64_t wmask = __ballot(need > 0);
            if (lane == 0) lds_cnt[wave] = (uint32_t)__popcll(wmask);
            if (master) lds_ctr[threadIdx.x] = ctr;
            __syncthreads();
            uint32_t R = 0, base = 0;
            const uint32_t nwaves = blockDim.x >> 6;
            for (uint32_t w = 0; w < nwaves; w++)
            {
                uint32_t c = lds_cnt[w];
                if (w < (uint32_t)wave) base += c;
                R += c;
            }
            if (R == 0) break;  // uniform over the workgroup
            const uint32_t grank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
            if (need > 0) lds_table[grank] = (uint16_t)threadIdx.x;
            __syncthreads();
            const uint32_t d      = threadIdx.x / R;
            const uint32_t target = lds_table[threadIdx.x - d * R];
            uint32_t tseed[16];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                uint4 v = *reinterpret_cast<const uint4 *>(lds_seed + target * 16 + 4 * i);
                tseed[4 * i] = v.x, tseed[4 * i + 1] = v.y, tseed[4 * i + 2] = v.z, tseed[4 * i + 3] = v.w;
            }
            const uint32_t hint = prefetch_entry();  // lands while the permutation runs
            const uint32_t hk   = k;
            KeccakState cs;
            prng_absorb(cs, tseed, lds_ctr[target] + d);
            keccak_f1600_fresh(cs);
            lds_cand[threadIdx.x] = cs.lo[0];
            __syncthreads();
            const uint32_t dmax = (blockDim.x - 1u) / R + 1u;
            for (uint32_t dd = 0; dd < dmax; dd++)
            {
                const uint32_t src = grank + dd * R;
                if (need > 0 && src < blockDim.x)
                {
                    const uint32_t x = lds_cand[src];
                    ctr++;
                    if (x < bound) place(x, k == hk, hint);
                }
            }
            __syncthreads();
        }

        for (; !wg_pool;)
        {
            const uint64_t mask = __ballot(need > 0);
            if (mask == 0) break;
            const uint32_t R      = (uint32_t)__popcll(mask);
            const uint32_t myrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (need > 0) rank2lane[myrank] = (uint8_t)lane;
            __builtin_amdgcn_wave_barrier();
            const uint32_t d      = (uint32_t)lane / R;          // counter offset of this slot
            const uint32_t target = rank2lane[(uint32_t)lane - d * R];
            __builtin_amdgcn_wave_barrier();

            const uint32_t hint = prefetch_entry();  // lands while the permutation runs
            const uint32_t hk   = k;
            uint32_t tseed[16];
#pragma unroll
            for (int i = 0; i < 16; i++) tseed[i] = (uint32_t)__shfl((int)seed[i], (int)target);
            const uint32_t tlo = (uint32_t)__shfl((int)(uint32_t)ctr, (int)target);
            const uint32_t thi = (uint32_t)__shfl((int)(uint32_t)(ctr >> 32), (int)target);
            KeccakState cs;
            prng_absorb(cs, tseed, ((((uint64_t)thi) << 32) | tlo) + d);
            keccak_f1600_fresh(cs);  // only cs.lo[0] is consumed
            const uint32_t cand = cs.lo[0];

            const uint32_t dmax = (63u / R) + 1u;  // most candidates any lane received
            for (uint32_t dd = 0; dd < dmax; dd++)
            {
                const uint32_t src = myrank + dd * R;
                const uint32_t x   = (uint32_t)__shfl((int)cand, (int)(src & 63u));
                if (need > 0 && src < 64u)
                {
                    ctr++;
                    if (x < bound) place(x, k == hk, hint);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (A.ctr_out && active) A.ctr_out[b] = ctr;
}

// ------------------------------------------------------------------------------------------
// The redraw phase of one polynomial by one WAVE (used by k_sample_uniform_wave and k_resolve_wave): `need`
// rejected coefficients take the accepted candidates of the stream block(ctr)[0:4], block(ctr + 1)[0:4], ...
// in counter order; a candidate is drawn -- `ctr` advanced -- only while one is still needed (sample.c:50-56).
// 64 candidates per round, one per lane: the first `row_len` come precomputed from `row` (k_candidates), the
// rest are computed here in the lane-per-state form.  The r-th accepted candidate of a round goes to the
// (done + r)-th rejected position: from the reject list while it holds all of them, otherwise the rejected
// positions are the marker words of the polynomial, found by a wave-wide scan (rare: tiny list capacities).
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wave_redraws(uint32_t q, uint32_t crh, uint32_t bound, uint32_t *mypoly,
                                             const uint32_t *mylist, uint32_t rej_cap, uint32_t need,
                                             const uint32_t (&seed)[16], uint64_t &ctr, const uint32_t *row,
                                             uint32_t row_len, int lane)
{
    const uint64_t lt  = (1ull << lane) - 1ull;   // lanes below this one
    const bool by_list = need <= rej_cap;
    uint32_t done = 0, scanpos = 0, t = 0;         // t: candidates of `row` already looked at
    while (need > 0)
    {
        uint32_t x, cnt;                            // this lane's candidate; candidates in this round
        if (t < row_len)
        {
            cnt = min(64u, row_len - t);
            x   = (uint32_t)lane < cnt ? __hip_atomic_load(row + t + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : 0xFFFFFFFFu;
            t += cnt;
        }
        else
        {
            cnt = 64;
            KeccakState cs;
            prng_absorb(cs, seed, ctr + (uint64_t)lane);
            keccak_f1600_fresh(cs);                 // only cs.lo[0] is consumed
            x = cs.lo[0];
        }
        const bool acc     = (uint32_t)lane < cnt && x < bound;
        const uint64_t am  = __ballot(acc);
        const uint32_t pre = (uint32_t)__popcll(am & lt);      // accepted candidates before this one
        const bool take    = acc && pre < need;                 // drawn (pre < need) and accepted
        const uint32_t val = barrett32(x, q, crh);
        if (by_list)
        {
            if (take)
            {
                const uint32_t pos = __hip_atomic_load(mylist + done + pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mypoly[pos]        = val;
            }
        }
        else
        {
            uint64_t tm = __ballot(take);
            while (tm != 0)
            {
                const int src = __builtin_ctzll(tm);
                tm &= tm - 1;
                const uint32_t v = (uint32_t)__shfl((int)val, src);
                for (;;)
                {
                    const uint32_t p  = scanpos + (uint32_t)lane;
                    const uint32_t c  = p < (uint32_t)N
                                            ? __hip_atomic_load(mypoly + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : 0u;
                    const uint64_t mk = __ballot(p < (uint32_t)N && c == kRejMarker);
                    if (mk != 0)
                    {
                        const uint32_t hit = scanpos + (uint32_t)__builtin_ctzll(mk);
                        if (lane == 0) mypoly[hit] = v;
                        __builtin_amdgcn_s_waitcnt(0);
                        __threadfence_block();
                        scanpos = hit + 1;
                        break;
                    }
                    scanpos += 64;
                    if (scanpos >= (uint32_t)N) break;   // cannot happen: `need` markers are left
                }
            }
        }
        const uint32_t got = (uint32_t)__popcll(am);
        if (got >= need)
        {
            // the need-th accepted candidate ends the draws: everything up to it was consumed
            const uint64_t last = __ballot(acc && pre == need - 1);
            ctr += (uint64_t)__builtin_ctzll(last) + 1;
            done += need;
            need = 0;
        }
        else
        {
            ctr += cnt;
            done += got;
            need -= got;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Staged form of the sampler for batches that leave SIMDs without a chain wave (one prime per launch):
//   k_bulk_pair   the bulk squeeze, ONE ciphertext per LANE PAIR (keccak.cuh, KeccakHalf: 126 instructions per
//                 round and lane instead of 190 -- the chain of a ciphertext, the critical path of the whole
//                 step at n = 16384, gets ~1.5x shorter); writes residues / markers, the reject list and count
//   k_candidates  block(ctr + 1 + k)[0:4], k < spec_cap, for every ciphertext: a plain throughput kernel on a
//                 second stream BESIDE the chains (what the helper waves of k_sample_uniform do inside the
//                 chain workgroups, at 160 VGPRs and pinned to the chains' SIMDs)
//   k_resolve_wave  one wave per ciphertext: consumes the candidates in counter order (wave_redraws), patches
//                 the rejected coefficients, leaves the next prime's start counter
// Same values, same counters as k_sample_uniform (tests: every shape against the oracle and the other forms).
// ------------------------------------------------------------------------------------------
// Write amplification (measured, round 4): a squeeze step yields 136 bytes of a row, so a row's 128-byte lines are
// completed by two consecutive steps ~10 us apart and reach HBM in two pieces -- 18.3 GB written per C4 step for
// 12.9 GB of a (1.42x).  Whole-line stores (a 16-word register carry per lane, the 16 stores of a line issued in
// one step) bring that to 1.21x but slow the chain by 7 % (16 scattered stores back to back stall a lone wave; the
// phases fully unrolled: 14 %, instruction cache), non-temporal accesses in the kernel streaming beside it change
// nothing: profiles/r04_ab_bulk_pair_lines.log.  The chain is the critical path of C4, the traffic costs no time:
// the direct stores stay.
template <int LOGN>
__global__ __launch_bounds__(512) void k_bulk_pair(DevParams P, UniformArgs A)
{
    constexpr int N          = 1 << LOGN;
    constexpr int FULL_STEPS = (N * 4) / 136;
    constexpr int TAIL_WORDS = N - FULL_STEPS * 34;
    static_assert(TAIL_WORDS % 2 == 0 && TAIL_WORDS <= 32, "the tail is a whole number of state lanes");
    extern __shared__ __attribute__((aligned(16))) unsigned char pin_lds[];   // reserved: one workgroup per CU
    (void)pin_lds;
    const uint32_t part = threadIdx.x & 1u;   // 0: low halves (and owner of the ciphertext's bookkeeping)
    const size_t bq     = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    const bool active   = bq < A.B;
    const size_t b      = active ? bq : (size_t)A.B - 1;   // idle pairs shadow the last ciphertext, store nothing
    const uint32_t j    = A.prime_lo;
    const uint32_t q = P.q[j], crh = P.cr_hi[j], bound = P.bound[j];
    uint32_t *mypoly = A.out + (b * A.out_primes + (j - A.out_prime_base)) * (size_t)N;
    uint32_t *mylist = A.rej_list + b * A.rej_cap;
    uint32_t seed[16];
    load_seed(seed, A.seeds, b);
    KeccakHalf st;
    prng_absorb_half(st, seed, A.ctr_in ? A.ctr_in[b] : 0, part);
    uint32_t nrej = 0;   // maintained on the owner lane
    const bool red4 = (uint64_t)bound <= 4ull * q;   // see reduce_sample

    // one squeeze step: word 2 i + part of the step comes from st.w[i]
    auto emit = [&](auto r4, uint32_t idx, int lanes64) {
        uint32_t mask = 0;   // bit (31 - i): word i of this lane rejected
#pragma unroll
        for (int i = 0; i < 17; i++)
        {
            if (i < lanes64)
            {
                const uint32_t x = st.w[i];
                const bool rej   = x >= bound;
                const uint32_t r = reduce_sample<decltype(r4)::value>(x, q, crh);
                mask |= (rej ? 0x80000000u : 0u) >> i;
                if (active) mypoly[idx + 2 * i + part] = rej ? kRejMarker : r;
            }
        }
        const uint32_t other = pair_swap(mask);
        if (__any((mask | other) != 0))
        {
            // owner: both lanes' rejects in position order (word 2 i of the owner before word 2 i + 1 of the partner)
            uint32_t m0 = part ? 0u : mask, m1 = part ? 0u : other;
            while (m0 | m1)
            {
                const uint32_t i0 = m0 ? (uint32_t)__clz((int)m0) : 64u, i1 = m1 ? (uint32_t)__clz((int)m1) : 64u;
                uint32_t pos;
                if (i0 <= i1)
                    pos = 2 * i0, m0 &= ~(0x80000000u >> i0);
                else
                    pos = 2 * i1 + 1, m1 &= ~(0x80000000u >> i1);
                if (active && nrej < A.rej_cap) mylist[nrej] = idx + pos;
                nrej++;
            }
        }
    };
    uint32_t idx = 0;
    for (int step = 0; step < FULL_STEPS; step++)
    {
        keccak_half_f1600(st, part);
        if (red4)
            emit(std::true_type{}, idx, 17);
        else
            emit(std::false_type{}, idx, 17);
        idx += 34;
    }
    if constexpr (TAIL_WORDS > 0)
    {
        keccak_half_f1600(st, part);
        emit(std::false_type{}, idx, TAIL_WORDS / 2);
    }
    if (active && part == 0) A.nrej[b] = nrej;
    if (A.flagged && blockIdx.x == 0 && threadIdx.x == 0) A.flagged[0] = 0;   // read by this prime's k_resolve_wave
}

// candidates V[b][k] = block(ctr_in[b] + 1 + k)[0:4], k < spec_cap; consecutive threads = consecutive k of one
// ciphertext.  512-thread workgroups and the phase-synchronised permutation, as k_sample_cbd.
__global__ __launch_bounds__(kCbdThreads) void k_candidates(UniformArgs A)
{
    const size_t gid   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)A.B * A.spec_cap;
    if (!__any(gid < total)) return;   // whole waves past the end END before the first barrier ...
    if (gid >= total) return;          // ... lanes of a partial wave are masked, the wave runs the permutation
    // (hipcc folds the two tests into one exec-masked region with a branch to s_endpgm: either way a wave that does not
    // run the block has ended -- the contract of keccak_sync.cuh, checked on the ISA by tests/test_keccak_sync.py)
    const size_t b   = gid / A.spec_cap;
    const uint32_t k = (uint32_t)(gid - b * A.spec_cap);
    uint32_t seed[16];
    load_seed(seed, A.seeds, b);
    const uint64_t ctr = (A.ctr_in ? A.ctr_in[b] : 0) + 1 + k;
    uint32_t w[18];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = seed[i];
    w[16] = (uint32_t)ctr;
    w[17] = (uint32_t)(ctr >> 32);
    keccak_fresh4_sync(w, &kKeccakRC[0][0]);
    A.spec[gid] = w[0];
}

// The candidate row of ciphertext b (k_candidates ran with this prime's start counters: the bulk block took counter c,
// the candidates are the counters c + 1, c + 2, ...).
__device__ __forceinline__ const uint32_t *candidate_row(const UniformArgs &A, size_t b, uint32_t &row_len)
{
    row_len = A.spec ? A.spec_cap : 0u;
    return A.spec ? A.spec + b * (size_t)A.spec_cap : nullptr;
}

// The common case of the resolve step without any Keccak in the kernel (24 VGPRs instead of 130: 8 waves per
// SIMD hide the load latencies; 0.32 -> 0.1 ms per prime at C4, on the critical path): a wave fetches its
// ciphertext's whole candidate row at once (up to 8 x 64 candidates), walks it, patches.  A ciphertext whose
// row runs out, or whose reject list overflowed, is flagged in A.nrej (top bit) and left to k_resolve_wave,
// which redoes it from the start (the patches are idempotent) with the candidates it computes itself.
template <int LOGN>
__global__ __launch_bounds__(256) void k_resolve_light(DevParams P, UniformArgs A)
{
    constexpr int N = 1 << LOGN;
    const int lane  = threadIdx.x & 63;
    const size_t b  = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= A.B) return;   // wave-uniform
    const uint32_t j    = A.prime_lo;
    const uint32_t nrej = A.nrej[b];
    uint32_t need       = nrej;
    const uint64_t c0   = A.ctr_in ? A.ctr_in[b] : 0;
    uint64_t ctr        = c0 + 1;   // the bulk block took one counter
    uint32_t row_len;
    const uint32_t *row   = candidate_row(A, b, row_len);
    const uint32_t rounds = (row_len + 63u) / 64u;
    auto flag = [&]() {
        if (lane == 0)
        {
            A.nrej[b] = nrej | 0x80000000u;
            if (A.flagged) A.flagged[1 + atomicAdd(A.flagged, 1u)] = (uint32_t)b;
        }
    };
    if (need > A.rej_cap || rounds > 8u || !A.spec)
    {
        flag();
        return;
    }
    const uint32_t q = P.q[j], crh = P.cr_hi[j], bound = P.bound[j];
    uint32_t *mypoly       = A.out + (b * A.out_primes + (j - A.out_prime_base)) * (size_t)N;
    const uint32_t *mylist = A.rej_list + b * A.rej_cap;
    const uint64_t lt      = (1ull << lane) - 1ull;
    uint32_t x[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
        x[r] = ((uint32_t)r < rounds && 64u * r + (uint32_t)lane < row_len) ? row[64 * r + lane] : 0xFFFFFFFFu;
    uint32_t done = 0;
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
        if (need > 0 && (uint32_t)r < rounds)
        {
            const uint32_t cnt = min(64u, row_len - 64u * r);
            const bool acc     = (uint32_t)lane < cnt && x[r] < bound;
            const uint64_t am  = __ballot(acc);
            const uint32_t pre = (uint32_t)__popcll(am & lt);
            if (acc && pre < need) mypoly[mylist[done + pre]] = barrett32(x[r], q, crh);
            const uint32_t got = (uint32_t)__popcll(am);
            if (got >= need)
            {
                ctr += (uint64_t)__builtin_ctzll(__ballot(acc && pre == need - 1)) + 1;
                done += need, need = 0;
            }
            else
                ctr += cnt, done += got, need -= got;
        }
    }
    if (need > 0)
    {
        flag();   // row too short: k_resolve_wave redoes this ciphertext
        return;
    }
    if (A.ctr_out && lane == 0) A.ctr_out[b] = ctr;
}

template <int LOGN>
__global__ __launch_bounds__(256) void k_resolve_wave(DevParams P, UniformArgs A)
{
    constexpr int N = 1 << LOGN;
    const int lane  = threadIdx.x & 63;
    const uint32_t j = A.prime_lo;
    // with a list of flagged ciphertexts: a small grid, every wave takes entries w, w + waves, ... of it
    const size_t w0 = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), wstep = (size_t)gridDim.x * (blockDim.x >> 6);
    const size_t count = A.flagged ? (size_t)A.flagged[0] : (size_t)A.B;
    for (size_t w = w0; w < count; w += wstep)   // wave-uniform
    {
        const size_t b    = A.flagged ? (size_t)A.flagged[1 + w] : w;
        const uint64_t c0 = A.ctr_in ? A.ctr_in[b] : 0;
        uint64_t ctr      = c0 + 1;   // the bulk block took one counter
        // after k_resolve_light: only the ciphertexts it flagged (top bit of the count) are left
        const uint32_t raw = A.nrej[b];
        if (A.master_waves == 1u && !(raw & 0x80000000u)) continue;
        const uint32_t need = raw & 0x7FFFFFFFu;
        if (need > 0)
        {
            uint32_t seed[16];
            load_seed(seed, A.seeds, b);
            uint32_t *mypoly = A.out + (b * A.out_primes + (j - A.out_prime_base)) * (size_t)N;
            uint32_t row_len;
            const uint32_t *row = candidate_row(A, b, row_len);
            wave_redraws<N>(P.q[j], P.cr_hi[j], P.bound[j], mypoly, A.rej_list + b * A.rej_cap, A.rej_cap, need, seed, ctr,
                            row, row_len, lane);
        }
        if (A.ctr_out && lane == 0) A.ctr_out[b] = ctr;
    }
}

// ------------------------------------------------------------------------------------------
// The same sampler for a HANDFUL of ciphertexts: one WAVE per ciphertext (keccak.cuh, WaveKeccak).
//
// The bulk block of a polynomial is one sequential sponge squeeze (121 permutations at n = 4096, 482 at
// n = 16384); a lane-per-ciphertext kernel spends ~10 us per permutation on it however few ciphertexts there
// are, which is all a single se_encrypt call consists of ("single calls").  Here the 64
// lanes of a wave share ONE state (~3.2 us per permutation measured, tools/ubench5), the 17 lanes that hold the
// rate words reduce / test / store their own two words (one 136-byte contiguous store per step), rejected
// positions are ranked with ballots, and the redraws -- independent SHAKE calls -- are 64 candidates per round,
// one per lane in the ordinary lane-per-state form, consumed in counter order exactly as sample.c:50-56 does
// (the k-th rejected coefficient takes the k-th accepted candidate; a candidate is drawn only while one is
// needed).  13x the instructions per state of the lane form: used for launches of at most a few waves per
// SIMD (launch_sample_uniform), i.e. single calls, the virtual ciphertexts of the prime speculation, small
// batches.  Same outputs, same end counters, same reject-list / marker conventions as k_sample_uniform.
// ------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(256) void k_sample_uniform_wave(DevParams P, UniformArgs A)
{
    constexpr int N          = 1 << LOGN;
    constexpr int FULL_STEPS = (N * 4) / 136;
    constexpr int TAIL_WORDS = N - FULL_STEPS * 34;
    static_assert(TAIL_WORDS % 2 == 0 && TAIL_WORDS <= 32, "the tail is a whole number of state lanes");

    const int lane  = threadIdx.x & 63;
    const size_t bq = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // wave-uniform: a wave without a ciphertext (or masked out of a redo launch) has nothing to do
    if (bq >= A.B) return;
    if (A.only_from && !(A.only_from[bq] != 0 && A.prime_lo >= A.only_from[bq])) return;
    const size_t b = bq;

    uint32_t seed[16];   // every lane: the lane-per-state candidates of the redraw phase need all of it
    load_seed(seed, A.seeds, b);
    const uint8_t *seedp = A.seeds + b * kSeedBytes;
    uint64_t ctr         = A.ctr_in ? A.ctr_in[b] : 0;
    uint32_t *mylist     = A.rej_list + b * A.rej_cap;
    const uint64_t lt    = (1ull << lane) - 1ull;   // lanes below this one

    WaveKeccak wk;
    wave_keccak_init(wk, lane);

    const uint32_t j_first = A.prime_of ? (uint32_t)A.prime_of[b] : A.prime_lo;   // wave-uniform
    const uint32_t j_end   = A.prime_of ? j_first + 1u : A.prime_hi;
    for (uint32_t j = j_first; j < j_end; j++)
    {
        const uint32_t q = P.q[j], crh = P.cr_hi[j], bound = P.bound[j];
        uint32_t *mypoly = A.out + (A.prime_of ? b : b * A.out_primes + (j - A.out_prime_base)) * (size_t)N;
        wave_prng_absorb(wk, seedp, ctr, lane);
        ctr++;
        uint32_t nrej = 0;   // wave-uniform

        // one squeeze step: the lanes holding state lanes 0 .. words/2 - 1 emit two words each
        auto emit = [&](uint32_t idx, int words) {
            const bool mine = wk.index >= 0 && 2 * wk.index < words;
            const bool r0 = mine && wk.lo >= bound, r1 = mine && wk.hi >= bound;
            if (mine)
            {
                const uint32_t w0 = r0 ? kRejMarker : barrett32(wk.lo, q, crh);
                const uint32_t w1 = r1 ? kRejMarker : barrett32(wk.hi, q, crh);
                *reinterpret_cast<uint2 *>(mypoly + idx + 2 * wk.index) = make_uint2(w0, w1);
            }
            const uint64_t m0 = __ballot(r0), m1 = __ballot(r1);
            if ((m0 | m1) != 0)
            {
                // primary lanes ascend with the state-lane index, and a lane's low word precedes its high
                // word: the rank of a rejected word in position order
                const uint32_t before = nrej + (uint32_t)__popcll(m0 & lt) + (uint32_t)__popcll(m1 & lt);
                if (r0 && before < A.rej_cap) mylist[before] = idx + 2 * wk.index;
                if (r1 && before + (r0 ? 1u : 0u) < A.rej_cap) mylist[before + (r0 ? 1u : 0u)] = idx + 2 * wk.index + 1;
                nrej += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
            }
        };
        uint32_t idx = 0;
        for (int step = 0; step < FULL_STEPS; step++)
        {
            wave_keccak_f1600(wk);
            emit(idx, 34);
            idx += 34;
        }
        if constexpr (TAIL_WORDS > 0)
        {
            wave_keccak_f1600(wk);
            emit(idx, TAIL_WORDS);
        }
        // bulk stores and list entries must have landed before other lanes patch / read them
        __builtin_amdgcn_s_waitcnt(0);
        __threadfence_block();

        // ---- redraws: 64 candidates block(ctr + lane)[0:4] per round -------------------------------
        wave_redraws<N>(q, crh, bound, mypoly, mylist, A.rej_cap, nrej, seed, ctr, nullptr,
                        0u, lane);
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (A.ctr_out && lane == 0) A.ctr_out[b] = ctr;
}

// ------------------------------------------------------------------------------------------
// Centered binomial error (k = 21): one 96-byte block -> 16 int8 coefficients per thread.
// counter = ctr_base[b] (or 0) + ctr_offset + block index.  Output int8 [B][blocks*16].
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t byte_window(const uint32_t (&w)[24], int byte_off)
{
    // 32 bits starting at a compile-time byte offset of the 96-byte block
    const int wi = byte_off >> 2, sh = (byte_off & 3) * 8;
    if (sh == 0) return w[wi];
    uint32_t hi = (wi + 1 < 24) ? w[wi + 1] : 0u;
    return __builtin_amdgcn_alignbit(hi, w[wi], sh);
}

// Round 4: 512-thread workgroups (two waves per SIMD and workgroup) and the phase-synchronised permutation of
// keccak_sync.cuh -- waves of one SIMD only pair their v_xor / v_bitop3 when they are in the same phase of the round
// (profiles/r04_ubench7_keccak_schedules.txt).
__global__ __launch_bounds__(kCbdThreads) void k_sample_cbd(CbdArgs A)
{
    const size_t gid   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)A.B * A.blocks_per_ct;
    if (!__any(gid < total)) return;   // whole waves past the end END before the first barrier ...
    if (gid >= total) return;          // ... lanes of a partial wave are masked, the wave runs the permutation
    // (hipcc folds the two tests into one exec-masked region with a branch to s_endpgm: either way a wave that does not
    // run the block has ended -- the contract of keccak_sync.cuh, checked on the ISA by tests/test_keccak_sync.py)
    const size_t b   = gid / A.blocks_per_ct;
    const uint32_t k = (uint32_t)(gid - b * A.blocks_per_ct);
    uint32_t seed[16];
    load_seed(seed, A.seeds, b);
    uint64_t ctr = (A.ctr_base ? A.ctr_base[b] : 0) + k;
    uint32_t w[24];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = seed[i];
    w[16] = (uint32_t)ctr;
    w[17] = (uint32_t)(ctr >> 32);
    keccak_fresh96_sync(w, &kKeccakRC[0][0]);
    // sample i = popcnt(bytes 6i,6i+1, low 5 bits of 6i+2) - popcnt(bytes 6i+3,6i+4, low 5 of 6i+5)
    uint32_t packed[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        uint32_t pos = byte_window(w, 6 * i) & 0x001FFFFFu;
        uint32_t neg = byte_window(w, 6 * i + 3) & 0x001FFFFFu;
        int v        = __popc(pos) - __popc(neg);
        packed[i >> 2] |= ((uint32_t)v & 0xFFu) << (8 * (i & 3));
    }
    *reinterpret_cast<uint4 *>(A.out + gid * 16) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
}

// ------------------------------------------------------------------------------------------
// Ternary u (asymmetric): lane per ciphertext.  Each iteration every lane runs ONE permutation
// of whichever kind it needs next -- a 96-byte block, or a 1-byte redraw for the lowest pending
// rejected byte of its current block (96-bit pending mask in 3 VGPRs) -- so lanes do not wait
// for each other's rejection loops.  Output: one int8 code (0,1,2) per coefficient.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t mod3_u8(uint32_t r)
{
    // r < 256: r mod 3 via multiply-shift (exact for 8-bit inputs)
    return r - 3u * ((r * 171u) >> 9);
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_sample_ternary(TernaryArgs A)
{
    const size_t b    = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = b < A.B;
    const size_t bs   = active ? b : (size_t)A.B - 1;
    const uint32_t n  = A.n;
    const uint32_t nblocks = (n + 95) / 96;
    uint32_t seed[16];
    load_seed(seed, A.seeds, bs);
    int8_t *out = A.codes + bs * n;

    uint64_t ctr  = A.ctr_in ? A.ctr_in[bs] : 0;
    uint32_t blk  = 0;                 // next block to draw
    uint32_t pend[3] = {0, 0, 0};      // pending rejected bytes of the current block
    uint32_t cur_base = 0;             // first coefficient of the current block
    bool done = !active;

    while (__any(!done))
    {
        KeccakState st;
        prng_absorb(st, seed, ctr);
        keccak_f1600_fresh<true>(st);  // a block uses 96 bytes, a redraw 1 byte
        if (!done)
        {
            ctr++;
            const bool redraw = (pend[0] | pend[1] | pend[2]) != 0;
            if (redraw)
            {
                uint32_t r = st.lo[0] & 0xFFu;
                if (r < 0xFEu)
                {
                    uint32_t pos;
                    if (pend[0]) { pos = __builtin_ctz(pend[0]); pend[0] &= pend[0] - 1; }
                    else if (pend[1]) { pos = 32 + __builtin_ctz(pend[1]); pend[1] &= pend[1] - 1; }
                    else { pos = 64 + __builtin_ctz(pend[2]); pend[2] &= pend[2] - 1; }
                    out[cur_base + pos] = (int8_t)mod3_u8(r);
                }
            }
            else
            {
                cur_base            = blk * 96;
                const uint32_t stop = min(96u, n - cur_base);
                uint32_t w[24];
#pragma unroll
                for (int i = 0; i < 12; i++)
                {
                    w[2 * i]     = st.lo[i];
                    w[2 * i + 1] = st.hi[i];
                }
#pragma unroll
                for (int wi = 0; wi < 24; wi++)
                {
                    uint32_t codes = 0, rejbits = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        uint32_t r = (w[wi] >> (8 * k)) & 0xFFu;
                        bool rej   = (r >= 0xFEu) && ((uint32_t)(4 * wi + k) < stop);
                        rejbits |= (rej ? 1u : 0u) << k;
                        codes |= mod3_u8(r) << (8 * k);
                    }
                    pend[wi >> 3] |= rejbits << (4 * (wi & 7));
                    if ((uint32_t)(4 * wi) < stop)  // stop is a multiple of 4 (n is a power of two >= 1024)
                        *reinterpret_cast<uint32_t *>(out + cur_base + 4 * wi) = codes;
                }
                blk++;
            }
            if (blk == nblocks && (pend[0] | pend[1] | pend[2]) == 0) done = true;
        }
    }
    if (A.ctr_out && active) A.ctr_out[b] = ctr;
}

// The same sampler for a handful of ciphertexts: one WAVE per ciphertext (keccak.cuh, WaveKeccak).  The chain
// of a ciphertext -- a 96-byte block per 96 coefficients, a 1-byte redraw for every byte >= 0xFE in byte
// order, ~75 sequential permutations at n = 4096 -- costs ~3.3 us per permutation instead of ~10 us; the 12
// lanes that hold the block's bytes turn their 8 bytes into codes and reject flags, the redraws are resolved
// one at a time in position order exactly as sample.c:226-238 does.
__global__ __launch_bounds__(256) void k_sample_ternary_wave(TernaryArgs A)
{
    const int lane  = threadIdx.x & 63;
    const size_t b  = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= A.B) return;   // wave-uniform
    const uint32_t n       = A.n;
    const uint32_t nblocks = (n + 95) / 96;
    const uint8_t *seedp   = A.seeds + b * kSeedBytes;
    int8_t *out            = A.codes + b * n;
    uint64_t ctr           = A.ctr_in ? A.ctr_in[b] : 0;
    WaveKeccak wk;
    wave_keccak_init(wk, lane);
    for (uint32_t blk = 0; blk < nblocks; blk++)
    {
        wave_prng_absorb(wk, seedp, ctr, lane);
        ctr++;
        wave_keccak_f1600(wk);
        const uint32_t base = blk * 96, stop = min(96u, n - base);   // stop is a multiple of 8 for n = 2^k >= 1024
        const bool mine     = wk.index >= 0 && 8u * (uint32_t)wk.index < stop;
        uint32_t pend       = 0;   // this lane's rejected bytes, bit k = byte 8 * index + k
        if (mine)
        {
            uint32_t c[2];
            const uint32_t w[2] = {wk.lo, wk.hi};
#pragma unroll
            for (int h = 0; h < 2; h++)
            {
                uint32_t codes = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const uint32_t r = (w[h] >> (8 * k)) & 0xFFu;
                    pend |= (r >= 0xFEu ? 1u : 0u) << (4 * h + k);
                    codes |= mod3_u8(r) << (8 * k);
                }
                c[h] = codes;
            }
            *reinterpret_cast<uint2 *>(out + base + 8 * wk.index) = make_uint2(c[0], c[1]);
        }
        // redraws, in byte order (primary lanes ascend with the state-lane index)
        uint64_t pm = __ballot(pend != 0);
        while (pm != 0)
        {
            const int src       = __builtin_ctzll(pm);
            const uint32_t bits = (uint32_t)__shfl((int)pend, src);
            const uint32_t k    = (uint32_t)__builtin_ctz(bits);
            const uint32_t idx  = (uint32_t)__shfl(wk.index, src);
            for (;;)
            {
                wave_prng_absorb(wk, seedp, ctr, lane);
                ctr++;
                wave_keccak_f1600(wk);
                const uint32_t r = (uint32_t)__shfl((int)wk.lo, 1) & 0xFFu;   // byte 0: state lane (0, 0) = wave lane 1
                if (r < 0xFEu)
                {
                    if (lane == 0) out[base + 8 * idx + k] = (int8_t)mod3_u8(r);
                    break;
                }
            }
            if (lane == src) pend &= pend - 1;
            pm = __ballot(pend != 0);
        }
    }
    if (A.ctr_out && lane == 0) A.ctr_out[b] = ctr;
}

// ------------------------------------------------------------------------------------------
// Ternary u, WINDOW form (round 4).  A block and a redraw are prefixes of the SAME stream function -- block(c) = the
// first 96 bytes, a redraw = the first byte of SHAKE256(seed || c) -- and the chain of a ciphertext only decides
// WHICH counter plays which part: counter c is a block, then one counter per redraw of its rejected bytes (in byte
// order, repeated while the redrawn byte is >= 0xFE: sample.c:226-238), then the next block.  The chain touches
// nblocks + n/128 +- a few counters (75.3 +- 5.7 at n = 4096), so:
//   1. every thread computes block(ctr0 + k)[0:96] for ONE counter k of a window of W counters of one ciphertext --
//      one phase-synchronised permutation (keccak_sync.cuh), no chain: 102 permutations per ciphertext instead of
//      ~75 sequential ones, ~20 us instead of ~1 ms;
//   2. one lane per ciphertext walks the window with per-counter summaries from LDS (96-bit reject mask, first byte)
//      and gives every counter its role: block j | redraw for coefficient p | consumed and rejected | unused;
//   3. the threads write: block counters their 96 codes, then (after a fence and a barrier) redraw counters their one.
// A ciphertext whose chain leaves the window (W = mean + 4.7 sigma: ~1e-6) is marked and redone by one lane of
// k_sample_ternary_redo with the sequential algorithm of k_sample_ternary.  Same codes, same end counter (tests: against the oracle and the other forms, and
// with a window so small that most ciphertexts take the fallback).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kTernWg = 512;

// the sequential chain of ONE ciphertext on one lane (the fallback of the window form)
__device__ void ternary_chain_lane(const uint32_t (&seed)[16], uint64_t &ctr, uint32_t n, int8_t *out)
{
    const uint32_t nblocks = (n + 95) / 96;
    for (uint32_t blk = 0; blk < nblocks; blk++)
    {
        KeccakState st;
        prng_absorb(st, seed, ctr);
        ctr++;
        keccak_f1600_fresh<true>(st);
        const uint32_t base = blk * 96, stop = min(96u, n - base);
        uint32_t w[24], pend[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 12; i++) w[2 * i] = st.lo[i], w[2 * i + 1] = st.hi[i];
#pragma unroll
        for (int wi = 0; wi < 24; wi++)
        {
            uint32_t codes = 0, rejbits = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t r = (w[wi] >> (8 * k)) & 0xFFu;
                rejbits |= ((r >= 0xFEu && (uint32_t)(4 * wi + k) < stop) ? 1u : 0u) << k;
                codes |= mod3_u8(r) << (8 * k);
            }
            pend[wi >> 3] |= rejbits << (4 * (wi & 7));
            if ((uint32_t)(4 * wi) < stop) *reinterpret_cast<uint32_t *>(out + base + 4 * wi) = codes;
        }
        while ((pend[0] | pend[1] | pend[2]) != 0)
        {
            KeccakState rs;
            prng_absorb(rs, seed, ctr);
            ctr++;
            keccak_f1600_fresh<true>(rs);
            const uint32_t r = rs.lo[0] & 0xFFu;
            if (r < 0xFEu)
            {
                uint32_t pos;
                if (pend[0]) { pos = __builtin_ctz(pend[0]); pend[0] &= pend[0] - 1; }
                else if (pend[1]) { pos = 32 + __builtin_ctz(pend[1]); pend[1] &= pend[1] - 1; }
                else { pos = 64 + __builtin_ctz(pend[2]); pend[2] &= pend[2] - 1; }
                out[base + pos] = (int8_t)mod3_u8(r);
            }
        }
    }
}

constexpr int8_t kTernRedo = 0x7F;   // codes[b][0] of a ciphertext whose chain left the window (real codes are 0, 1, 2)

__global__ __launch_bounds__(kTernWg) void k_sample_ternary_window(TernaryArgs A, uint32_t W, uint32_t cpw)
{
    __shared__ uint32_t s_mask[kTernWg * 3];
    __shared__ uint32_t s_role[kTernWg];    // block counter: j; accepted redraw: 0x80000000 | block counter << 8 | ordinal
    __shared__ uint16_t s_sum[kTernWg + 8]; // per counter: rejected bytes (all 96 | first `last`) and "first byte accepted"
    __shared__ uint32_t s_over[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t ctl = tid / W;          // ciphertext of this workgroup
    const uint32_t k   = tid - ctl * W;    // counter of its window
    const size_t b0    = (size_t)blockIdx.x * cpw;
    const bool valid   = ctl < cpw && b0 + ctl < A.B;
    if (!__any(valid)) return;             // whole waves without work END before the first barrier (keccak_sync.cuh)
    const size_t b     = valid ? b0 + ctl : (size_t)A.B - 1;
    const uint32_t n = A.n, nblocks = (n + 95) / 96, last = n - (nblocks - 1) * 96;
    int8_t *out = A.codes + b * n;

    // 1. this thread's counter
    uint32_t w[24];
    {
        uint32_t seed[16];
        load_seed(seed, A.seeds, b);
        const uint64_t c = (A.ctr_in ? A.ctr_in[b] : 0) + k;
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = seed[i];
        w[16] = (uint32_t)c;
        w[17] = (uint32_t)(c >> 32);
        keccak_fresh96_sync(w, &kKeccakRC[0][0]);
    }
    {
        uint32_t m[3] = {0, 0, 0};
#pragma unroll
        for (int wi = 0; wi < 24; wi++)
        {
            uint32_t rejbits = 0;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) rejbits |= ((((w[wi] >> (8 * kk)) & 0xFFu) >= 0xFEu) ? 1u : 0u) << kk;
            m[wi >> 3] |= rejbits << (4 * (wi & 7));
        }
        s_mask[3 * tid] = m[0], s_mask[3 * tid + 1] = m[1], s_mask[3 * tid + 2] = m[2];
        // rejected bytes among all 96 / among the first `last` (what counts when this counter is the LAST block)
        uint32_t l0 = m[0], l1 = m[1], l2 = m[2];
        if (last < 32) l0 &= (1u << last) - 1u;
        if (last <= 32) l1 = 0; else if (last < 64) l1 &= (1u << (last - 32)) - 1u;
        if (last <= 64) l2 = 0; else if (last < 96) l2 &= (1u << (last - 64)) - 1u;
        const uint32_t full = __popc(m[0]) + __popc(m[1]) + __popc(m[2]), part = __popc(l0) + __popc(l1) + __popc(l2);
        s_sum[tid]  = (uint16_t)(full | (part << 7) | (((w[0] & 0xFFu) < 0xFEu) ? 0x4000u : 0u));
        s_role[tid] = 0xFFFFFFFFu;
        if (tid < 16) s_over[tid] = 0;
    }
    __syncthreads();

    // 2. one lane per ciphertext deals the roles: the counters are visited in order, one summary each
    if (tid < cpw && b0 + tid < A.B)
    {
        const uint32_t base = tid * W;
        uint32_t j = 0, need = 0, cb = 0, ord = 0, c = 0;
        for (; c < W && (j < nblocks || need > 0); c++)
        {
            const uint32_t sm = s_sum[base + c];
            if (need == 0)
            {
                s_role[base + c] = j;
                need = (j == nblocks - 1) ? ((sm >> 7) & 0x7Fu) : (sm & 0x7Fu);
                cb = c, ord = 0, j++;
            }
            else if (sm & 0x4000u)
            {
                s_role[base + c] = 0x80000000u | (cb << 8) | ord;
                ord++, need--;
            }
        }
        const bool over = j < nblocks || need > 0;
        s_over[tid]     = over ? 1u : 0u;
        if (over)
            A.codes[(b0 + tid) * n] = kTernRedo;     // k_sample_ternary_redo picks it up
        else if (A.ctr_out)
            A.ctr_out[b0 + tid] = (A.ctr_in ? A.ctr_in[b0 + tid] : 0) + c;
    }
    __syncthreads();

    // 3. block counters write their codes, then redraw counters theirs
    const uint32_t role = s_role[tid];
    const bool skip     = !valid || s_over[ctl < 16 ? ctl : 0] != 0;
    if (!skip && role < 0x80000000u)
    {
        const uint32_t stop = (role == nblocks - 1) ? last : 96u;
        int8_t *dst         = out + role * 96;
#pragma unroll
        for (int q4 = 0; q4 < 6; q4++)
        {
            uint32_t cd[4];
#pragma unroll
            for (int wi = 0; wi < 4; wi++)
            {
                uint32_t codes = 0;
#pragma unroll
                for (int kk = 0; kk < 4; kk++) codes |= mod3_u8((w[4 * q4 + wi] >> (8 * kk)) & 0xFFu) << (8 * kk);
                cd[wi] = codes;
            }
            if ((uint32_t)(16 * q4) < stop)     // stop is a multiple of 16 (n = 2^k >= 1024)
                *reinterpret_cast<uint4 *>(dst + 16 * q4) = make_uint4(cd[0], cd[1], cd[2], cd[3]);
        }
    }
    // the block stores of this workgroup must be in L2 before a redraw byte lands inside one of their words: the stores
    // are acknowledged (vmcnt 0) and the order is a WORKGROUP matter (an agent-scope fence writes the L2 back: 5x slower)
    __builtin_amdgcn_s_waitcnt(0);
    __threadfence_block();
    __syncthreads();
    if (!skip && role != 0xFFFFFFFFu && role >= 0x80000000u)
    {
        // the ord-th rejected byte of the block counter's 96
        const uint32_t cbk = ctl * W + ((role >> 8) & 0x3FFu), ord = role & 0xFFu;
        uint32_t m0 = s_mask[3 * cbk], m1 = s_mask[3 * cbk + 1], m2 = s_mask[3 * cbk + 2], pos = 0;
        for (uint32_t i = 0; i <= ord; i++)
        {
            if (m0) { pos = __builtin_ctz(m0); m0 &= m0 - 1; }
            else if (m1) { pos = 32 + __builtin_ctz(m1); m1 &= m1 - 1; }
            else { pos = 64 + __builtin_ctz(m2); m2 &= m2 - 1; }
        }
        out[s_role[cbk] * 96 + pos] = (int8_t)mod3_u8(w[0] & 0xFFu);
    }
}

// The ciphertexts k_sample_ternary_window could not finish (marker in codes[b][0]): sequentially, a lane each.  Launched
// behind every window launch; with windows of mean + 4.7 sigma its lanes read one byte and leave.
__global__ __launch_bounds__(64) void k_sample_ternary_redo(TernaryArgs A)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B || A.codes[b * A.n] != kTernRedo) return;
    uint32_t seed[16];
    load_seed(seed, A.seeds, b);
    uint64_t ctr = A.ctr_in ? A.ctr_in[b] : 0;
    ternary_chain_lane(seed, ctr, A.n, A.codes + b * A.n);
    if (A.ctr_out) A.ctr_out[b] = ctr;
}

// Raw PRNG blocks for tests: out[i] = SHAKE256(seed[i] || le64(ctr[i]))[0 : outlen], lane per block.
__global__ __launch_bounds__(64) void k_prng_blocks(const uint8_t *seeds, const uint64_t *ctrs,
                                                    uint8_t *out, uint32_t outlen, uint32_t count)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    uint32_t seed[16];
    load_seed(seed, seeds, i);
    KeccakState st;
    prng_absorb(st, seed, ctrs[i]);
    uint8_t *dst = out + i * outlen;
    for (uint32_t off = 0; off < outlen; off += kShakeRate)
    {
        keccak_f1600(st);
        uint32_t take = min((uint32_t)kShakeRate, outlen - off);
        for (uint32_t by = 0; by < take; by++)
        {
            uint32_t lane64 = by >> 3, sh = (by & 7) * 8;
            uint32_t word = 0;
#pragma unroll
            for (int l = 0; l < 17; l++)
                if ((uint32_t)l == lane64) word = sh < 32 ? st.lo[l] >> sh : st.hi[l] >> (sh - 32);
            dst[off + by] = (uint8_t)word;
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// Lane-per-ciphertext kernels run ONE long sequential chain per lane, so the kernel lasts as long
// as its slowest SIMD: the waves must be spread evenly over the chip.  Left to the dispatcher,
// 1-wave workgroups launched behind another kernel were observed to pile 2 waves on some SIMDs
// and none on others (k_sample_uniform 7.0 ms alone -> 11.4 ms in the pipeline).  We therefore
// launch workgroups of w waves (w = 1, 2, 3, 4, 8, 12, 16) and reserve > 80 KiB of dynamic LDS per
// workgroup, which admits exactly one workgroup per CU: every CU gets the same number of waves
// and the hardware deals a workgroup's waves round-robin over its 4 SIMDs.
static void chain_geometry(size_t B, unsigned num_cus, unsigned &threads, unsigned &grid, size_t &lds_bytes,
                           unsigned *master_waves = nullptr, bool allow_helpers = false,
                           size_t fill_to = 8)
{
    const size_t cus   = num_cus ? num_cus : 256;
    const size_t waves = (B + 63) / 64;
    size_t w           = (waves + cus - 1) / cus;   // waves per CU if spread over all CUs
    if (w < 1) w = 1;
    if (w > 4) w = ((w + 3) / 4) * 4;           // beyond one per SIMD: whole multiples of 4
    if (w > 16) w = 16;
    size_t helpers = 0;
    if (allow_helpers && w < 4) helpers = fill_to - w;  // small batch: fill the CU to 2 waves per SIMD
    threads   = (unsigned)((w + helpers) * 64);   // with helper waves for the redraw phase
    grid      = (unsigned)((B + w * 64 - 1) / (w * 64));
    lds_bytes = 84 * 1024;
    if (master_waves) *master_waves = (unsigned)w;
}

template <int LOGN>
static hipError_t launch_uniform_wave(const DevParams &P, const UniformArgs &A, hipStream_t st)
{
    const unsigned waves_per_wg = 4;   // one per SIMD of the CU a workgroup lands on
    hipLaunchKernelGGL((k_sample_uniform_wave<LOGN>), dim3((A.B + waves_per_wg - 1) / waves_per_wg),
                       dim3(64 * waves_per_wg), 0, st, P, A);
    return hipGetLastError();
}

hipError_t launch_sample_uniform(const DevParams &P, const UniformArgs &A0, hipStream_t st)
{
    if (A0.B == 0) return hipSuccess;
    // A handful of chains: one wave per ciphertext (2.2-2.7x shorter chains up to one wave per SIMD, break-even
    // near 4 per SIMD -- tools/ubench5).  debug_flags 32 / 64 force the lane / the wave form (tests).
    if ((A0.B <= uniform_wave_limit(P.num_cus) && !(A0.debug_flags & 32)) || (A0.debug_flags & 64))
    {
        switch (P.logn)
        {
            case 10: return launch_uniform_wave<10>(P, A0, st);
            case 11: return launch_uniform_wave<11>(P, A0, st);
            case 12: return launch_uniform_wave<12>(P, A0, st);
            case 13: return launch_uniform_wave<13>(P, A0, st);
            case 14: return launch_uniform_wave<14>(P, A0, st);
            default: return hipErrorInvalidValue;
        }
    }
    unsigned threads, grid_x, mw;
    size_t lds;
    chain_geometry(A0.B, P.num_cus, threads, grid_x, lds, &mw, !(A0.debug_flags & 8),
                   A0.helper_fill ? A0.helper_fill : 8);
    UniformArgs A   = A0;
    A.master_waves  = mw;
    if (A.debug_flags & 16) A.spec = nullptr;  // A/B: helper waves without speculation
    // the per-lane-prime form exists for workgroups of up to 8 waves (every speculation launch: <= 65 536
    // virtual ciphertexts, small_limit)
    if (A.prime_of && threads > 512) return hipErrorInvalidValue;
    dim3 grid(grid_x), block(threads);
#define SEAMD_LAUNCH_UNIFORM_T(L, T)                                                             \
    (void)hipFuncSetAttribute((const void *)k_sample_uniform<L, T>,                              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL((k_sample_uniform<L, T>), grid, block, lds, st, P, A)
#define SEAMD_LAUNCH_UNIFORM_P(L)                                                                \
    (void)hipFuncSetAttribute((const void *)k_sample_uniform<L, 512, true>,                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL((k_sample_uniform<L, 512, true>), grid, block, lds, st, P, A)
#define SEAMD_LAUNCH_UNIFORM(L)                                                                  \
    if (A.prime_of) { SEAMD_LAUNCH_UNIFORM_P(L); }                                               \
    else if (threads > 512) { SEAMD_LAUNCH_UNIFORM_T(L, 1024); } else { SEAMD_LAUNCH_UNIFORM_T(L, 512); }
    switch (P.logn)
    {
        case 10: SEAMD_LAUNCH_UNIFORM(10); break;
        case 11: SEAMD_LAUNCH_UNIFORM(11); break;
        case 12: SEAMD_LAUNCH_UNIFORM(12); break;
        case 13: SEAMD_LAUNCH_UNIFORM(13); break;
        case 14: SEAMD_LAUNCH_UNIFORM(14); break;
        default: return hipErrorInvalidValue;
    }
#undef SEAMD_LAUNCH_UNIFORM
#undef SEAMD_LAUNCH_UNIFORM_P
#undef SEAMD_LAUNCH_UNIFORM_T
    return hipGetLastError();
}

// ---- staged form (one prime per launch; se_context.cpp runs k_candidates on a stream of its own) ----
hipError_t launch_uniform_bulk_pair(const DevParams &P, const UniformArgs &A, hipStream_t st)
{
    if (A.B == 0) return hipSuccess;
    if (!A.nrej || A.prime_hi != A.prime_lo + 1) return hipErrorInvalidValue;
    // like chain_geometry: workgroups of w waves, one per CU (84 KiB of reserved LDS), w <= 8 pair waves
    const size_t cus   = P.num_cus ? P.num_cus : 256;
    const size_t waves = (2 * (size_t)A.B + 63) / 64;
    size_t w           = (waves + cus - 1) / cus;
    if (w < 1) w = 1;
    if (w > 8) w = 8;
    const unsigned threads = (unsigned)(w * 64), grid = (unsigned)((2 * (size_t)A.B + threads - 1) / threads);
    const size_t lds = 84 * 1024;
#define SEAMD_LAUNCH_PAIR(L)                                                                                    \
    (void)hipFuncSetAttribute((const void *)k_bulk_pair<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_bulk_pair<L>), dim3(grid), dim3(threads), lds, st, P, A)
    switch (P.logn)
    {
        case 10: SEAMD_LAUNCH_PAIR(10); break;
        case 11: SEAMD_LAUNCH_PAIR(11); break;
        case 12: SEAMD_LAUNCH_PAIR(12); break;
        case 13: SEAMD_LAUNCH_PAIR(13); break;
        case 14: SEAMD_LAUNCH_PAIR(14); break;
        default: return hipErrorInvalidValue;
    }
#undef SEAMD_LAUNCH_PAIR
    return hipGetLastError();
}

hipError_t launch_uniform_candidates(const UniformArgs &A, hipStream_t st)
{
    const size_t total = (size_t)A.B * A.spec_cap;
    if (total == 0 || !A.spec) return hipSuccess;
    hipLaunchKernelGGL(k_candidates, dim3((unsigned)((total + kCbdThreads - 1) / kCbdThreads)), dim3(kCbdThreads), 0, st, A);
    return hipGetLastError();
}

hipError_t launch_uniform_resolve(const DevParams &P, const UniformArgs &A, hipStream_t st)
{
    if (A.B == 0) return hipSuccess;
    if (!A.nrej) return hipErrorInvalidValue;
    const dim3 grid((A.B + 3) / 4), block(256);
    // the light kernel resolves (nearly) every ciphertext; the heavy one -- master_waves = 1 tells it that the
    // light one ran -- picks up the flagged rest (normally none: its waves read one word and leave)
    UniformArgs H   = A;
    H.master_waves  = 1;
    // with a flagged list: one workgroup per CU walks it (normally it is empty or a handful of entries)
    const dim3 hgrid(A.flagged ? std::min<unsigned>((unsigned)((A.B + 3) / 4), P.num_cus ? P.num_cus : 256u) : grid.x);
#define SEAMD_LAUNCH_RESOLVE(L)                                              \
    hipLaunchKernelGGL((k_resolve_light<L>), grid, block, 0, st, P, A);      \
    hipLaunchKernelGGL((k_resolve_wave<L>), hgrid, block, 0, st, P, H)
    switch (P.logn)
    {
        case 10: SEAMD_LAUNCH_RESOLVE(10); break;
        case 11: SEAMD_LAUNCH_RESOLVE(11); break;
        case 12: SEAMD_LAUNCH_RESOLVE(12); break;
        case 13: SEAMD_LAUNCH_RESOLVE(13); break;
        case 14: SEAMD_LAUNCH_RESOLVE(14); break;
        default: return hipErrorInvalidValue;
    }
#undef SEAMD_LAUNCH_RESOLVE
    return hipGetLastError();
}

hipError_t launch_sample_cbd(const CbdArgs &A, hipStream_t st)
{
    size_t total = (size_t)A.B * A.blocks_per_ct;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sample_cbd, dim3((unsigned)((total + kCbdThreads - 1) / kCbdThreads)), dim3(kCbdThreads), 0, st, A);
    return hipGetLastError();
}

hipError_t launch_sample_ternary(const TernaryArgs &A, hipStream_t st)
{
    if (A.B == 0) return hipSuccess;
    if (!(A.debug_flags & (32 | 64)))
    {
        // Window form (k_sample_ternary_window): W = blocks + mean + 4.7 sigma of the redraws (2 of 256 byte values are
        // rejected), as many ciphertexts per 512-thread workgroup as fit, the window widened to the lanes that are left.
        // n <= 4096: every batch size (102 permutations per ciphertext, no chain); beyond that the windows no longer
        // fill a workgroup (n = 16384: 354 of 512 lanes, 1.7x the permutations of the chain) and the form serves the
        // batches whose chains would leave SIMDs empty (up to one chain wave per SIMD: n = 16384, B = 16 384: 3.56 ms
        // of 300-deep chains against ~1 ms).
        // debug_flags 4096: a window of blocks + 2 counters, so that nearly every ciphertext takes the fallback (tests).
        const uint32_t nblocks = (A.n + 95) / 96;
        const double mean      = (double)A.n * (2.0 / 254.0);
        uint32_t W = (A.debug_flags & 4096) ? nblocks + 2 : nblocks + (uint32_t)(mean + 4.7 * sqrt(mean) + 1.0);
        if (W <= kTernWg && (A.n <= 4096 || A.B <= (size_t)256 * (A.num_cus ? A.num_cus : 256u)))
        {
            const uint32_t cpw = std::min<uint32_t>(kTernWg / W, 15u);
            if (!(A.debug_flags & 4096)) W = kTernWg / cpw;
            hipLaunchKernelGGL(k_sample_ternary_window, dim3((A.B + cpw - 1) / cpw), dim3(kTernWg), 0, st, A, W, cpw);
            hipLaunchKernelGGL(k_sample_ternary_redo, dim3((A.B + 63) / 64), dim3(64), 0, st, A);
            return hipGetLastError();
        }
    }
    if (A.B <= uniform_wave_limit(A.num_cus) && !(A.debug_flags & 32))
    {
        // a handful of chains: one wave per ciphertext (see k_sample_uniform_wave)
        hipLaunchKernelGGL(k_sample_ternary_wave, dim3((A.B + 3) / 4), dim3(256), 0, st, A);
        return hipGetLastError();
    }
    unsigned threads, grid_x;
    size_t lds;
    chain_geometry(A.B, A.num_cus, threads, grid_x, lds);
    if (threads > 512)
    {
        (void)hipFuncSetAttribute((const void *)k_sample_ternary<1024>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sample_ternary<1024>, dim3(grid_x), dim3(threads), lds, st, A);
    }
    else   // 167 VGPRs, no spill (1.20 -> 1.06 ms per 65 536)
    {
        (void)hipFuncSetAttribute((const void *)k_sample_ternary<512>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sample_ternary<512>, dim3(grid_x), dim3(threads), lds, st, A);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Small-batch prime speculation.  Prime j+1 of a ciphertext starts at the PRNG counter where prime
// j's redraws stopped, which makes the primes of ONE ciphertext a sequential chain of np long
// squeezes -- all there is to do when the batch is a single ciphertext.  The stop counter is
// 1 + (draws consumed) with draws ~ Binomial(n, p) plus a few rejected candidates: a narrow window.
// So the sampler of prime j is run for EVERY start counter of that window as independent "virtual
// ciphertexts" (same seed, guessed counter, own output row), all primes at once, and afterwards
// the true chain is followed through the results: c_1 = end of prime 0, pick guess c_1 - base_1,
// take its end counter as c_2, ...  A counter outside a window sets fail[b] = j and leaves the true
// counter in ctr0[b]; the caller then launches the ordinary per-prime chain masked to those
// ciphertexts (UniformArgs::only_from), which normally finds nothing to do.  Idle lanes are free in a one-ciphertext call; the latency drops from np
// squeezes to one.
// ------------------------------------------------------------------------------------------
__global__ void k_spec_setup(SpecPlan S, const uint8_t *seeds, uint8_t *seeds_v, uint64_t *ctr_v,
                             uint8_t *prime_v)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.total) return;
    uint32_t j = 1;
    while (j + 1 < S.nprimes && v >= S.offset[j + 1]) j++;
    const uint32_t r = v - S.offset[j];
    const uint32_t b = r / S.count[j], g = r - b * S.count[j];
    const uint4 *src = reinterpret_cast<const uint4 *>(seeds + (size_t)b * kSeedBytes);
    uint4 *dst       = reinterpret_cast<uint4 *>(seeds_v + (size_t)v * kSeedBytes);
#pragma unroll
    for (int i = 0; i < 4; i++) dst[i] = src[i];
    ctr_v[v]   = S.base[j] + g;
    prime_v[v] = (uint8_t)j;
}

// one workgroup per real ciphertext: follow the counter chain through the guesses and copy the
// selected rows into c1[b][j][:]
__global__ __launch_bounds__(256) void k_spec_select(SpecPlan S, uint32_t n, uint64_t *ctr0,
                                                     const uint64_t *ctrout_v, const uint32_t *rows,
                                                     uint32_t *c1, uint32_t *fail)
{
    const uint32_t b = blockIdx.x;
    // every thread must have the start counter before thread 0 may overwrite ctr0[b] on a miss
    __shared__ uint64_t start_ctr;
    if (threadIdx.x == 0) start_ctr = ctr0[b];
    __syncthreads();
    uint64_t ctr = start_ctr;
    for (uint32_t j = 1; j < S.nprimes; j++)
    {
        const uint64_t g = ctr - S.base[j];   // wraps to a huge value when ctr < base
        if (g >= S.count[j])
        {
            // miss: prime j .. np-1 of this ciphertext are redone sequentially from this counter
            if (threadIdx.x == 0)
            {
                fail[b] = j;
                ctr0[b] = ctr;
            }
            return;
        }
        const size_t v      = (size_t)S.offset[j] + (size_t)b * S.count[j] + (size_t)g;
        const uint4 *src    = reinterpret_cast<const uint4 *>(rows + v * n);
        uint4 *dst          = reinterpret_cast<uint4 *>(c1 + ((size_t)b * S.nprimes + j) * n);
        for (uint32_t i = threadIdx.x; i < n / 4; i += blockDim.x) dst[i] = src[i];
        ctr = ctrout_v[v];
    }
    if (threadIdx.x == 0) fail[b] = 0;
}

hipError_t launch_spec_setup(const SpecPlan &S, const uint8_t *seeds, uint8_t *seeds_v, uint64_t *ctr_v,
                             uint8_t *prime_v, hipStream_t st)
{
    if (S.total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_spec_setup, dim3((S.total + 255) / 256), dim3(256), 0, st, S, seeds, seeds_v, ctr_v,
                       prime_v);
    return hipGetLastError();
}

hipError_t launch_spec_select(const SpecPlan &S, uint32_t n, uint64_t *ctr0, const uint64_t *ctrout_v,
                              const uint32_t *rows, uint32_t *c1, uint32_t *fail, hipStream_t st)
{
    if (S.B == 0) return hipSuccess;
    hipLaunchKernelGGL(k_spec_select, dim3(S.B), dim3(256), 0, st, S, n, ctr0, ctrout_v, rows, c1, fail);
    return hipGetLastError();
}

hipError_t launch_prng_blocks(const uint8_t *seeds, const uint64_t *ctrs, uint8_t *out,
                              uint32_t outlen, uint32_t count, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prng_blocks, dim3((count + 63) / 64), dim3(64), 0, st, seeds, ctrs, out,
                       outlen, count);
    return hipGetLastError();
}

}  // namespace seamd
