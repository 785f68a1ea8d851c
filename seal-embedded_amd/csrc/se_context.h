// se_context.h -- internal: the per-parameter-set GPU context behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <vector>

#include "kernels/kernel_args.h"
#include "se_host_tables.h"
#include "se_types.h"

namespace seamd {

// error codes of include/seal_embedded_amd.h (kept numerically identical; checked in se_api.cpp)
constexpr int kErrInvalid  = -22;    // SE_ERR_INVALD_ARGUMENT
constexpr int kErrNoDevice = -19;    // SE_ERR_NO_DEVICE
constexpr int kErrHip      = -1001;  // SE_ERR_HIP
constexpr int kErrNoKey    = -1002;  // SE_ERR_NO_KEY

constexpr int kStageCount = 6;  // cbd, uniform, ternary, encode_encrypt (fused), encode_rns, ntt_fuse

struct HostPipe;

struct StageEvent
{
    int stage;
    hipEvent_t start, stop;
};

struct Context
{
    HostParams hp;
    DevParams dp;
    DevTables dt{};
    int device = 0;
    std::vector<uint16_t> index_map;  // host copy (SE_PTRS::index_map_ptr, tests)

    // read-only device slabs
    uint16_t *d_inv_map = nullptr;
    double *d_ifft_w    = nullptr;
    uint32_t *d_ntt_rw  = nullptr;
    uint32_t *d_intt_rw = nullptr;
    uint16_t *d_map     = nullptr;
    uint16_t *d_gather  = nullptr;
    uint32_t *d_s_hat   = nullptr;
    uint32_t *d_pk0     = nullptr;
    uint32_t *d_pk1     = nullptr;
    bool have_sk = false, have_pk = false;

    // scratch, grown on demand
    int8_t *d_err      = nullptr;  // [cap][2n]
    int8_t *d_ucodes   = nullptr;  // [cap][n]
    uint64_t *d_ctr    = nullptr;  // [cap]
    uint32_t *d_rej    = nullptr;  // [cap][rej_cap]
    uint32_t *d_spec   = nullptr;  // [cap][spec_cap] speculative redraw candidates (helper waves)
    uint32_t spec_cap  = 128;
    uint32_t *d_a      = nullptr;  // [a_cap][np][n]: `a` when the caller does not want c1 back
    size_t a_cap       = 0;
    // small-batch prime speculation (encrypt_sym_small): virtual-ciphertext scratch and streams
    uint8_t *d_sp_seeds   = nullptr;  // [sp_cap][64]
    uint64_t *d_sp_ctr    = nullptr;  // [sp_cap] guessed start counters
    uint64_t *d_sp_ctrout = nullptr;  // [sp_cap] end counters under each guess
    uint32_t *d_sp_rows   = nullptr;  // [sp_cap][n] a_j under each guess
    uint8_t *d_sp_prime   = nullptr;  // [sp_cap] prime of each virtual ciphertext
    uint32_t *d_sp_fail   = nullptr;  // [sp_fail_cap] 0 = chain resolved, j = window of prime j missed
    size_t sp_cap = 0, sp_fail_cap = 0;
    uint32_t small_limit = getenv("SE_AMD_SMALL_LIMIT") ? (uint32_t)atoi(getenv("SE_AMD_SMALL_LIMIT")) : 65536;  // virtual ciphertexts a small call may fan out to
    // ... and the bytes their output rows may take (one n-word row per virtual ciphertext)
    size_t small_bytes = getenv("SE_AMD_SMALL_BYTES") ? (size_t)atoll(getenv("SE_AMD_SMALL_BYTES")) : ((size_t)1 << 30);
    hipStream_t spec_stream = nullptr;   // the guesses of ALL primes run as one launch on it
    // staged sampler (k_bulk_pair / k_candidates / k_resolve_wave): candidates on a stream of their own
    hipStream_t cand_stream = nullptr;
    hipEvent_t ev_cand[kMaxPrimes] = {};
    uint32_t *d_nrej = nullptr;          // [scratch_cap] rejected coefficients of the current polynomial
    uint32_t *d_flagged = nullptr;       // [1 + scratch_cap] staged forms: ciphertexts k_resolve_light left to k_resolve_wave
    uint8_t *d_compact  = nullptr;  // [scratch_cap] k_encode_rns -> k_ntt_fuse: plaintext b travels as one int32 row
    uint32_t *d_general = nullptr;  // [1 + general_cap] plaintexts the fast fused kernel declined (count, indices)
    size_t general_cap  = 0;
    size_t scratch_cap = 0;   // ciphertexts d_err / d_ucodes / d_ctr hold
    size_t rows_cap    = 0;   // rows of d_rej / d_spec (>= scratch_cap: virtual ciphertexts need only these)
    uint32_t rej_cap   = 256;
    uint32_t debug_flags = 0;  // timing ablations of the uniform sampler (tests/tools only)

    // second stream: the CBD error sampler runs beside the uniform sampler (different seeds, no
    // data dependency); joined before the fused encode+encrypt kernel.
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_cbd = nullptr, ev_enc = nullptr;
    hipEvent_t ev_prime[kMaxPrimes] = {};
    // One set of scratch per context: successive calls are ordered on it.  Host threads serialise on
    // `mu`; a call waits (on its own stream) for `ev_done` of the previous call, whatever stream
    // that one ran on, before it touches the scratch or forks the auxiliary streams.
    std::mutex mu;
    hipEvent_t ev_done = nullptr;
    bool have_done     = false;
    int num_cus        = 256;
    // public-key path: chunks the batch is cut into so that the CBD sampler of chunk k+1 runs beside the
    // fused kernel of chunk k (se_context.cpp, encrypt_asym_impl); 1 = serial
    size_t asym_chunks = getenv("SE_AMD_ASYM_CHUNKS") ? (size_t)atoi(getenv("SE_AMD_ASYM_CHUNKS")) : 1;
    int spec_mode = -1;    // prime speculation of small symmetric calls: -1 = estimate per call, 0 never, 1 whenever planned
    int staged_mode = -1;  // pair-form staged sampler of the per-prime pipeline: -1 = by batch size, 0 never, 1 always (SE_AMD_STAGED)
    bool overlap = true;   // run independent kernels on the auxiliary stream
    int split_mode = 2;    // symmetric path: 0 = fused kernel, 1 = per-prime software pipeline
                           // (encode_rns + uniform_j || ntt_fuse_{j-1}), 2 = choose per call: the split
                           // form wins whenever the uniform sampler's chains leave SIMDs empty (fewer
                           // than 4 chain waves per CU) or the fused kernel spills (n >= 8192);
                           // at n = 4096, B = 65536 both measure the same and fused moves less data

    // host-pointer entry points: chunked PCIe pipeline (se_hostpipe.h), created on first use
    HostPipe *host_pipe = nullptr;

    // profiling
    bool profiling = false;
    std::vector<StageEvent> events;
    float stage_ms[kStageCount]          = {};
    uint64_t stage_launches[kStageCount] = {};

    ~Context();
    int init(size_t n, size_t nprimes, int device);
    int ensure_scratch(size_t B, size_t rows = 0);
    int ensure_general(size_t B);
    int begin_call(hipStream_t st);
    int end_call(hipStream_t st, int rc);
    // u codes (0/1/2 per coefficient) and e1 of ciphertext 0 of the last asymmetric call (host out)
    int fetch_asym_randomness(int8_t *ucodes, int8_t *e1);
    int set_secret_key(const uint8_t *sk_packed);
    int set_secret_key_impl(const uint8_t *sk_packed);   // caller holds `mu`
    int set_public_key(const uint32_t *pk0, const uint32_t *pk1);
    int gen_public_key(const uint8_t *sk_packed, const uint8_t *pk_seed, const uint8_t *ep_seed,
                       uint32_t *pk0_out, uint32_t *pk1_out);
    // K key pairs in one launch chain (host pointers); does not touch the context's installed keys
    int gen_keys_batch(size_t K, const uint8_t *sk_in, const uint8_t *sk_seeds, const uint8_t *pk_seeds,
                       const uint8_t *ep_seeds, uint8_t *sk_out, uint32_t *pk0_out, uint32_t *pk1_out);

    // public entries: serialised on the context's scratch (begin_call / end_call around *_impl)
    int encrypt_sym(const float *d_values, size_t B, const uint8_t *d_share_seeds,
                    const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1, uint32_t *d_ntt_pte,
                    int64_t *d_pte, uint8_t *d_status, hipStream_t st);
    int encrypt_sym_seeded(const float *d_values, size_t B, const uint8_t *d_share_seeds,
                           const uint8_t *d_seeds, uint32_t *d_c0, uint8_t *d_status, hipStream_t st);
    int encrypt_asym(const float *d_values, size_t B, const uint8_t *d_seeds, uint32_t *d_c0,
                     uint32_t *d_c1, uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status,
                     hipStream_t st);
    int encode_ntt(const float *d_values, size_t B, uint32_t *d_out, int64_t *d_pte,
                   uint8_t *d_status, hipStream_t st);
    int sample_uniform(const uint8_t *d_seeds, const uint64_t *d_ctr_in, size_t B, uint32_t *d_out,
                       uint64_t *d_ctr_out, hipStream_t st);
    int encrypt_sym_impl(const float *d_values, size_t B, const uint8_t *d_share_seeds,
                         const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1, uint32_t *d_ntt_pte,
                         int64_t *d_pte, uint8_t *d_status, hipStream_t st);
    int encrypt_asym_impl(const float *d_values, size_t B, const uint8_t *d_seeds, uint32_t *d_c0,
                          uint32_t *d_c1, uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status,
                          hipStream_t st);
    // Small batches (a handful of ciphertexts): all primes' uniform samplers at once under guessed
    // start counters (kernels/samplers.hip, k_spec_*).  small_batch_plan says whether a batch
    // qualifies (encrypt_sym dispatches on it).  A counter outside its window (~1e-7 per prime) is
    // redone on the device by the masked per-prime chain that follows the selection.
    bool small_batch_plan(size_t B, SpecPlan &plan) const;
    // speculation or the plain per-prime chain for this batch (estimated chain latencies of both)
    bool speculation_pays(size_t B, const SpecPlan &plan) const;
    int encrypt_sym_small(const SpecPlan &plan, const float *d_values, const uint8_t *d_share_seeds,
                          const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1, uint32_t *d_ntt_pte,
                          int64_t *d_pte, uint8_t *d_status, hipStream_t st);

    void stage_begin(int stage, hipStream_t st);
    void stage_end(hipStream_t st);
    void collect_events();
};

void set_last_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);

}  // namespace seamd

// the opaque handle of include/seal_embedded_amd.h
struct se_amd_ctx
{
    seamd::Context c;
};

#define SEAMD_HIP(call)                                             \
    do                                                              \
    {                                                               \
        hipError_t e__ = (call);                                    \
        if (e__ != hipSuccess) return seamd::hip_fail(e__, #call);  \
    } while (0)
