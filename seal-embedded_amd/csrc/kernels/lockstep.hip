// lockstep.hip -- the two samplers of a symmetric ciphertext (uniform `a`, CBD error `e`) in ONE kernel whose
// waves run their SHAKE256 permutations in lockstep (round 4).
//
// Replaces, for full batches of the fused symmetric pipeline, the pair k_sample_uniform || k_sample_cbd:
//   sample_poly_uniform                      /root/reference/device/lib/sample.c:39-57   (per prime, counters carried)
//   sample_add_poly_cbd_generic_inpl_prng_16 /root/reference/device/lib/sample.c:311-356 (counters 0 .. n/16 - 1)
//   prng_fill_buffer                         /root/reference/device/lib/rng.h:78-91
//
// Why: gfx950 issues the v_xor / v_bitop3 two thirds of a Keccak round at their fast rate only when two waves of a
// SIMD present such instructions at the same time (tools/keccak_sched.py, profiles/r04_ubench7_keccak_schedules.txt).
// The bulk squeeze of `a` is one sequential chain per ciphertext -- a lone wave per SIMD at 65 536 ciphertexts --
// and a co-running kernel cannot be kept in phase with it.  Here a workgroup is 4 MASTER waves (one per SIMD, a
// ciphertext per lane: the chains) plus 4 HELPER waves (one per SIMD), and every permutation of the kernel is the
// phase-synchronised form of keccak_sync.cuh (96 workgroup barriers per permutation): while a master squeezes step t
// of its polynomial, the helper lane of the same ciphertext computes redraw candidate t (the candidate counters are
// known before the squeeze: the bulk block consumes exactly one) and, once the candidates are out, CBD blocks.
// Masters never compute candidates; what a ciphertext needs beyond `cand_cap` candidates comes from a pooled
// workgroup loop afterwards (all 512 lanes, still in lockstep).  After the last prime all 8 waves finish the CBD
// blocks.  Same values and counters as k_sample_uniform / k_sample_cbd (tests: every form against the oracle).
//
// EVERY wave of the workgroup executes the same number of synchronised permutations (keccak_sync.cuh, CONTRACT):
// the loops below have workgroup-uniform trip counts and each iteration holds exactly one permutation per wave.
#include <hip/hip_runtime.h>

#include "../se_types.h"
#include "kernel_args.h"
#include "keccak.cuh"
#include "keccak_sync.cuh"
#include "modarith.cuh"

namespace seamd {

namespace {

constexpr uint32_t kMarker = 0xFFFFFFFFu;   // >= every modulus: "rejected, to be redrawn" (as in samplers.hip)
constexpr uint32_t kCts    = 256;           // ciphertexts (master lanes) per workgroup
constexpr uint32_t kWg     = 512;           // 4 master + 4 helper waves

__device__ __forceinline__ void ls_load_seed(uint32_t (&seed)[16], const uint8_t *seeds, size_t b)
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

__device__ __forceinline__ uint32_t ls_byte_window(const uint32_t (&w)[24], int byte_off)
{
    const int wi = byte_off >> 2, sh = (byte_off & 3) * 8;
    if (sh == 0) return w[wi];
    uint32_t hi = (wi + 1 < 24) ? w[wi + 1] : 0u;
    return __builtin_amdgcn_alignbit(hi, w[wi], sh);
}

// first 96 bytes of block(seed, ctr), every wave of the workgroup in lockstep
__device__ __forceinline__ void ls_block96(uint32_t (&w)[24], const uint32_t (&seed)[16], uint64_t ctr)
{
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = seed[i];
    w[16] = (uint32_t)ctr;
    w[17] = (uint32_t)(ctr >> 32);
    keccak_fresh96_sync(w, &kKeccakRC[0][0]);
}

// one CBD block (sample.c:263-284, :311-321): 96 bytes -> 16 int8 coefficients
__device__ __forceinline__ void ls_cbd_store(const uint32_t (&w)[24], int8_t *dst)
{
    uint32_t packed[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        uint32_t pos = ls_byte_window(w, 6 * i) & 0x001FFFFFu;
        uint32_t neg = ls_byte_window(w, 6 * i + 3) & 0x001FFFFFu;
        int v        = __popc(pos) - __popc(neg);
        packed[i >> 2] |= ((uint32_t)v & 0xFFu) << (8 * (i & 3));
    }
    *reinterpret_cast<uint4 *>(dst) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
}

}  // namespace

template <int LOGN>
__global__ __launch_bounds__(kWg) void k_sym_lockstep(DevParams P, LockstepArgs A)
{
    constexpr int N          = 1 << LOGN;
    constexpr int FULL_STEPS = (N * 4) / 136;
    constexpr int TAIL_WORDS = N - FULL_STEPS * 34;
    constexpr int STEPS      = FULL_STEPS + (TAIL_WORDS > 0 ? 1 : 0);
    constexpr uint32_t BPC   = N / 16;                    // CBD blocks per ciphertext
    static_assert(TAIL_WORDS <= 32 && TAIL_WORDS % 2 == 0, "tail fits one mask");
    static_assert((kCts * BPC) % kWg == 0 && BPC % 64 == 0, "CBD jobs are dealt by whole waves");

    // dynamic LDS: > 80 KiB are requested so that ONE workgroup runs per CU (its 8 waves are 2 per SIMD)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *lds_seed  = reinterpret_cast<uint32_t *>(lds_raw);            // [256][16] shareable seeds (a)
    uint32_t *lds_eseed = lds_seed + kCts * 16;                             // [256][16] error seeds (e)
    uint64_t *lds_ctr   = reinterpret_cast<uint64_t *>(lds_eseed + kCts * 16);  // [256]
    uint32_t *lds_cand  = reinterpret_cast<uint32_t *>(lds_ctr + kCts);     // [512]
    uint16_t *lds_table = reinterpret_cast<uint16_t *>(lds_cand + kWg);     // [256]
    uint32_t *lds_cnt   = reinterpret_cast<uint32_t *>(lds_table + kCts);   // [8]

    const uint32_t tid  = threadIdx.x;
    const int lane      = (int)(tid & 63u);
    // wave-uniform values the COMPILER knows to be uniform: every branch around a synchronised permutation must be a
    // scalar branch (an exec-masked region that is entered with an empty mask would still execute its s_barriers)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool master   = wave < kCts / 64;
    const uint32_t cti  = master ? tid : tid - kCts;      // ciphertext of the workgroup this lane serves
    const size_t ct0    = (size_t)blockIdx.x * kCts;
    const size_t b      = ct0 + cti;                      // the host launches B / 256 workgroups exactly

    // masters and the helper lane of the same ciphertext both hold the shareable seed; the error seeds go to LDS
    uint32_t seed[16];
    ls_load_seed(seed, A.share_seeds, b);
    {
        uint32_t es[16];
        if (!master) ls_load_seed(es, A.seeds, b);
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            if (master)
                lds_seed[cti * 16 + i] = seed[i];
            else
                lds_eseed[cti * 16 + i] = es[i];
        }
    }
    uint64_t ctr     = 0;
    uint32_t *mylist = A.rej_list + b * A.rej_cap;
    uint32_t cbd_next = 0;                                // CBD jobs of this workgroup already done (uniform)
    constexpr uint32_t cbd_total = kCts * BPC;
    const uint32_t cand_cap = min(A.cand_cap, (uint32_t)STEPS);
    __syncthreads();

    // one CBD job: block `blk` of ciphertext `c` of this workgroup
    auto cbd_job = [&](uint32_t job) {
        const uint32_t c = job / BPC, blk = job - c * BPC;
        uint32_t es[16];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            uint4 v = *reinterpret_cast<const uint4 *>(lds_eseed + c * 16 + 4 * i);
            es[4 * i] = v.x, es[4 * i + 1] = v.y, es[4 * i + 2] = v.z, es[4 * i + 3] = v.w;
        }
        uint32_t w[24];
        ls_block96(w, es, (uint64_t)blk);
        ls_cbd_store(w, A.err + ((ct0 + c) * BPC + blk) * 16);
    };

    for (uint32_t j = 0; j < A.nprimes; j++)
    {
        const uint32_t q = P.q[j], crh = P.cr_hi[j], bound = P.bound[j];
        uint32_t *mypoly = A.out + (b * A.nprimes + j) * (size_t)N;
        uint32_t nrej    = 0;
        const uint64_t bulk_ctr = ctr;   // the 4n-byte block; the redraw candidates follow at ctr + 1 ..
        ctr++;                           // masters and helpers carry the same counter chain
        const bool red4 = (uint64_t)bound <= 4ull * q;

        if (master)
        {
            // ---- the chain: STEPS squeeze steps, rejected positions to the list (sample.c:48-56) ----------
            uint32_t st[50];
#pragma unroll
            for (int i = 0; i < 16; i++) st[i] = seed[i];
            st[16] = (uint32_t)bulk_ctr;
            st[17] = (uint32_t)(bulk_ctr >> 32);
            st[18] = 0x1Fu;
#pragma unroll
            for (int i = 19; i < 50; i++) st[i] = 0;
            st[33] = 0x80000000u;

            auto word = [&](auto r4, uint32_t x, uint32_t &mask) -> uint32_t {
                const bool rej   = x >= bound;
                const uint32_t r = reduce_sample<decltype(r4)::value>(x, q, crh);
                mask             = (mask << 1) | (rej ? 1u : 0u);
                return rej ? kMarker : r;
            };
            // mask holds `count` words, word w of the step at bit (count - 1 - w); ascending positions
            auto flush = [&](uint32_t mask, uint32_t count, uint32_t first_pos) {
                while (__any(mask != 0))
                {
                    if (mask != 0)
                    {
                        const uint32_t p = (uint32_t)__clz((int)mask);
                        mask &= ~(0x80000000u >> p);
                        if (nrej < A.rej_cap) mylist[nrej] = first_pos + (p - (32u - count));
                        nrej++;
                    }
                }
            };
            uint32_t idx = 0;
            for (int step = 0; step < STEPS; step++)
            {
                keccak_f1600_sync(st, &kKeccakRC[0][0]);
                if (step < FULL_STEPS)
                {
                    uint32_t m0 = 0, m1 = 0;
                    auto emit = [&](auto r4) {
#pragma unroll
                        for (int i = 0; i < 17; i++)
                        {
                            uint32_t &mk = (i < 16) ? m0 : m1;
                            uint32_t w0  = word(r4, st[2 * i], mk);
                            uint32_t w1  = word(r4, st[2 * i + 1], mk);
                            *reinterpret_cast<uint2 *>(mypoly + idx + 2 * i) = make_uint2(w0, w1);
                        }
                    };
                    if (red4)
                        emit(std::true_type{});
                    else
                        emit(std::false_type{});
                    if (__any((m0 | m1) != 0))
                    {
                        flush(m0, 32, idx);
                        flush(m1, 2, idx + 32);
                    }
                    idx += 34;
                }
                else if constexpr (TAIL_WORDS > 0)
                {
                    uint32_t m0 = 0;
#pragma unroll
                    for (int i = 0; i < TAIL_WORDS / 2; i++)
                    {
                        uint32_t w0 = word(std::false_type{}, st[2 * i], m0);
                        uint32_t w1 = word(std::false_type{}, st[2 * i + 1], m0);
                        *reinterpret_cast<uint2 *>(mypoly + idx + 2 * i) = make_uint2(w0, w1);
                    }
                    flush(m0, TAIL_WORDS, idx);
                }
            }
        }
        else
        {
            // ---- helpers: candidate `step` of the own ciphertext, then CBD blocks ---------------------------
            for (int step = 0; step < STEPS; step++)
            {
                if ((uint32_t)step < cand_cap)
                {
                    uint32_t w[24];
                    ls_block96(w, seed, ctr + (uint64_t)step);
                    A.spec[b * A.spec_cap + (uint32_t)step] = w[0];
                }
                else
                {
                    // whole waves: cbd_total and every operand are multiples of 64
                    const uint32_t job0 = __builtin_amdgcn_readfirstlane(cbd_next + ((uint32_t)step - cand_cap) * kCts +
                                                                         (cti & ~63u));
                    if (job0 < cbd_total)
                        cbd_job(job0 + (uint32_t)lane);
                    else
                        keccak_null_sync();
                }
            }
        }
        cbd_next = min(cbd_total, cbd_next + ((uint32_t)STEPS - cand_cap) * kCts);   // the same in every thread

        // the bulk stores, the list entries and the helpers' candidates must have landed before they are read
        __builtin_amdgcn_s_waitcnt(0);
        __threadfence_block();
        __syncthreads();

        // ---- redraws: the k-th rejected coefficient takes the k-th accepted candidate of the stream
        //      block(ctr)[0:4], block(ctr + 1)[0:4], ...; a draw is consumed only while one is still needed ---------
        uint32_t need    = master ? nrej : 0u;
        uint32_t k       = 0;
        uint32_t scanpos = 0;
        auto list_entry = [&](uint32_t kk) -> uint32_t {
            return __hip_atomic_load(mylist + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto place = [&](uint32_t x) {
            uint32_t pos;
            if (k < A.rej_cap)
                pos = list_entry(k);
            else
            {
                pos = scanpos;   // list overflow: the rejected positions are exactly the marker words
                while (__hip_atomic_load(mypoly + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kMarker) pos++;
            }
            scanpos     = pos + 1;
            mypoly[pos] = barrett32(x, q, crh);
            k++;
            need--;
        };
        if (master)
        {
            const uint32_t *row = A.spec + b * (size_t)A.spec_cap;
            for (uint32_t t = 0; t < cand_cap && need > 0; t += 4)
            {
                uint32_t x[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    x[i] = (t + i < cand_cap)
                               ? __hip_atomic_load(row + t + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : kMarker;
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    if (need > 0 && t + i < cand_cap)
                    {
                        ctr++;
                        if (x[i] < bound) place(x[i]);
                    }
                }
            }
        }
        // pooled loop for what is still missing: all 512 lanes, one lockstep permutation per round; slot s serves
        // needy lane s % R with counter offset s / R (the dealing scheme of k_sample_uniform's workgroup pool)
        for (;;)
        {
            const uint64_t wmask = __ballot(need > 0);
            if (lane == 0) lds_cnt[wave] = (uint32_t)__popcll(wmask);
            if (master) lds_ctr[cti] = ctr;
            __syncthreads();
            uint32_t R = 0, base = 0;
            for (uint32_t w = 0; w < kWg / 64; w++)
            {
                const uint32_t c = lds_cnt[w];
                if (w < wave) base += c;
                R += c;
            }
            if (R == 0) break;   // uniform over the workgroup
            const uint32_t grank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
            if (need > 0) lds_table[grank] = (uint16_t)cti;
            __syncthreads();
            const uint32_t d      = tid / R;
            const uint32_t target = lds_table[tid - d * R];
            uint32_t tseed[16];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                uint4 v = *reinterpret_cast<const uint4 *>(lds_seed + target * 16 + 4 * i);
                tseed[4 * i] = v.x, tseed[4 * i + 1] = v.y, tseed[4 * i + 2] = v.z, tseed[4 * i + 3] = v.w;
            }
            uint32_t w[24];
            ls_block96(w, tseed, lds_ctr[target] + d);
            lds_cand[tid] = w[0];
            __syncthreads();
            const uint32_t dmax = (kWg - 1u) / R + 1u;
            for (uint32_t dd = 0; dd < dmax; dd++)
            {
                const uint32_t src = grank + dd * R;
                if (need > 0 && src < kWg)
                {
                    const uint32_t x = lds_cand[src];
                    ctr++;
                    if (x < bound) place(x);
                }
            }
            __syncthreads();
        }
        __builtin_amdgcn_s_waitcnt(0);
        // the helper lane of a ciphertext follows its master's counter into the next prime
        if (master) lds_ctr[cti] = ctr;
        __syncthreads();
        ctr = lds_ctr[cti];
        __syncthreads();
    }
    if (A.ctr_out && master) A.ctr_out[b] = ctr;

    // ---- the rest of the CBD blocks: all 8 waves ------------------------------------------------------------
    for (uint32_t base = cbd_next; base < cbd_total; base += kWg)
    {
        const uint32_t job0 = __builtin_amdgcn_readfirstlane(base + (tid & ~63u));   // whole waves
        if (job0 < cbd_total)
            cbd_job(job0 + (uint32_t)lane);
        else
            keccak_null_sync();
    }
}

hipError_t launch_sym_lockstep(const DevParams &P, const LockstepArgs &A, int logn, hipStream_t st)
{
    if (A.B == 0 || A.B % kCts != 0 || A.cand_cap > A.spec_cap) return hipErrorInvalidValue;
    const size_t lds = 84 * 1024;   // one workgroup per CU
    const dim3 grid(A.B / kCts), block(kWg);
#define SEAMD_LS_CASE(L)                                                                                       \
    case L:                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_sym_lockstep<L>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                   \
        hipLaunchKernelGGL(k_sym_lockstep<L>, grid, block, lds, st, P, A);                                     \
        break;
    switch (logn)
    {
        SEAMD_LS_CASE(10)
        SEAMD_LS_CASE(11)
        SEAMD_LS_CASE(12)
    default:
        return hipErrorInvalidValue;
    }
#undef SEAMD_LS_CASE
    return hipGetLastError();
}

}  // namespace seamd
