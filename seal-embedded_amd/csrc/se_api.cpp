// se_api.cpp -- the extern "C" surface declared in include/seal_embedded_amd.h.
//
// Layer 1 mirrors /root/reference/device/lib/seal_embedded.c:24-235 (se_setup*, se_encrypt*,
// se_cleanup) on top of the GPU context; layer 2 exposes the batched context API.
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/seal_embedded_amd.h"
#include "se_context.h"
#include "se_hostpipe.h"
#include "se_sha256.h"

namespace seamd {
const std::string &last_error();
}
using seamd::Context;
using seamd::HostPipe;

static_assert(seamd::kErrInvalid == SE_ERR_INVALD_ARGUMENT && seamd::kErrNoDevice == SE_ERR_NO_DEVICE &&
                  seamd::kErrHip == SE_ERR_HIP && seamd::kErrNoKey == SE_ERR_NO_KEY,
              "internal error codes must match the public header");

static hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

const char *se_amd_last_error(void) { return seamd::last_error().c_str(); }
const char *se_amd_version(void) { return "seal-embedded_amd 0.1 (gfx950)"; }

int se_amd_create(se_amd_ctx **out, size_t degree, size_t nprimes, int device)
{
    if (!out) return SE_ERR_INVALD_ARGUMENT;
    *out          = nullptr;
    // The small-batch path runs one sampler launch per prime on streams of their own; with the ROCm
    // default of 4 hardware queues per process some of them share a queue and serialise (n = 16384 /
    // 6 primes: 12.9 ms per call instead of 7.0 ms; 3-prime chains fit 4 queues).  The library does not
    // touch the host process's environment on its own: a caller who wants the extra queues exports
    // GPU_MAX_HW_QUEUES itself, or opts in with SE_AMD_HW_QUEUES=<n>, which is applied here -- and only
    // effective when this is the first HIP use of the process.
    if (const char *hq = getenv("SE_AMD_HW_QUEUES"))
        if (atoi(hq) > 0) setenv("GPU_MAX_HW_QUEUES", hq, 0);
    se_amd_ctx *h = new (std::nothrow) se_amd_ctx();
    if (!h) return SE_ERR_NO_MEMORY;
    int rc = h->c.init(degree, nprimes, device);
    if (rc != 0)
    {
        delete h;
        return rc;
    }
    *out = h;
    return SE_SUCCESS;
}

void se_amd_destroy(se_amd_ctx *ctx) { delete ctx; }

size_t se_amd_degree(const se_amd_ctx *ctx) { return ctx ? ctx->c.hp.n : 0; }
size_t se_amd_nprimes(const se_amd_ctx *ctx) { return ctx ? ctx->c.hp.nprimes : 0; }
double se_amd_scale(const se_amd_ctx *ctx) { return ctx ? ctx->c.hp.scale : 0.0; }

int se_amd_moduli(const se_amd_ctx *ctx, uint32_t *q)
{
    if (!ctx || !q) return SE_ERR_INVALD_ARGUMENT;
    for (size_t j = 0; j < ctx->c.hp.nprimes; j++) q[j] = ctx->c.hp.q[j];
    return SE_SUCCESS;
}

int se_amd_index_map(const se_amd_ctx *ctx, uint16_t *map)
{
    if (!ctx || !map) return SE_ERR_INVALD_ARGUMENT;
    memcpy(map, ctx->c.index_map.data(), ctx->c.hp.n * sizeof(uint16_t));
    return SE_SUCCESS;
}

int se_amd_set_secret_key(se_amd_ctx *ctx, const uint8_t *sk_packed)
{
    if (!ctx || !sk_packed) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.set_secret_key(sk_packed);
}

int se_amd_set_public_key(se_amd_ctx *ctx, const uint32_t *pk0, const uint32_t *pk1)
{
    if (!ctx || !pk0 || !pk1) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.set_public_key(pk0, pk1);
}

int se_amd_gen_public_key(se_amd_ctx *ctx, const uint8_t *sk_packed, const uint8_t *pk_seed,
                          const uint8_t *ep_seed, uint32_t *pk0, uint32_t *pk1)
{
    if (!ctx || !sk_packed || !pk_seed || !ep_seed || !pk0 || !pk1) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.gen_public_key(sk_packed, pk_seed, ep_seed, pk0, pk1);
}

int se_amd_gen_keys_batch(se_amd_ctx *ctx, size_t K, const uint8_t *sk_in, const uint8_t *sk_seeds,
                          const uint8_t *pk_seeds, const uint8_t *ep_seeds, uint8_t *sk_out, uint32_t *pk0,
                          uint32_t *pk1)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.gen_keys_batch(K, sk_in, sk_seeds, pk_seeds, ep_seeds, sk_out, pk0, pk1);
}

static int read_exact(const std::string &path, void *dst, size_t bytes)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
    {
        seamd::set_last_error("cannot open key file " + path + ": " + strerror(errno));
        return SE_ERR_INVALD_ARGUMENT;
    }
    size_t got = fread(dst, 1, bytes, f);
    fclose(f);
    if (got != bytes)
    {
        seamd::set_last_error("short read on key file " + path);
        return SE_ERR_INVALD_ARGUMENT;
    }
    return SE_SUCCESS;
}

// File formats: fileops.c:140-204 (device side) / adapter/fileops.cpp:58-75,209-258 (writer side)
int se_amd_load_keys_from_dir(se_amd_ctx *ctx, const char *dir, int want_pk)
{
    if (!ctx || !dir) return SE_ERR_INVALD_ARGUMENT;
    const size_t n = ctx->c.hp.n, np = ctx->c.hp.nprimes;
    char name[128];
    if (!want_pk)
    {
        std::vector<uint8_t> sk(n / 4);
        snprintf(name, sizeof(name), "/sk_%zu.dat", n);
        int rc = read_exact(std::string(dir) + name, sk.data(), sk.size());
        if (rc) return rc;
        return ctx->c.set_secret_key(sk.data());
    }
    std::vector<uint32_t> pk0(np * n), pk1(np * n);
    for (size_t j = 0; j < np; j++)
    {
        snprintf(name, sizeof(name), "/pk0_ntt_%zu_%u.dat", n, ctx->c.hp.q[j]);
        int rc = read_exact(std::string(dir) + name, pk0.data() + j * n, n * 4);
        if (rc) return rc;
        snprintf(name, sizeof(name), "/pk1_ntt_%zu_%u.dat", n, ctx->c.hp.q[j]);
        rc = read_exact(std::string(dir) + name, pk1.data() + j * n, n * 4);
        if (rc) return rc;
    }
    return ctx->c.set_public_key(pk0.data(), pk1.data());
}

int se_amd_encrypt_sym_device(se_amd_ctx *ctx, const float *d_values, size_t B,
                              const uint8_t *d_share_seeds, const uint8_t *d_seeds, uint32_t *d_c0,
                              uint32_t *d_c1, uint32_t *d_ntt_pte, int64_t *d_pte,
                              uint8_t *d_status, void *stream)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.encrypt_sym(d_values, B, d_share_seeds, d_seeds, d_c0, d_c1, d_ntt_pte, d_pte,
                              d_status, as_stream(stream));
}

// Seed-compressed symmetric ciphertext (the reference only has a stub for it,
// seal_embedded.c:184-194): c1 = a is a deterministic expansion of the 64-byte shareable seed
// from counter 0, so a sender ships (share_seed, c0) -- half the bytes -- and the receiver
// re-expands a with se_amd_expand_c1_device (= sample_poly_uniform over the prime chain).
int se_amd_encrypt_sym_seeded_device(se_amd_ctx *ctx, const float *d_values, size_t B,
                                     const uint8_t *d_share_seeds, const uint8_t *d_seeds,
                                     uint32_t *d_c0, uint8_t *d_status, void *stream)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.encrypt_sym_seeded(d_values, B, d_share_seeds, d_seeds, d_c0, d_status, as_stream(stream));
}

int se_amd_expand_c1_device(se_amd_ctx *ctx, const uint8_t *d_share_seeds, size_t B, uint32_t *d_c1,
                            void *stream)
{
    return se_amd_sample_uniform_device(ctx, d_share_seeds, nullptr, B, d_c1, nullptr, stream);
}

int se_amd_encrypt_asym_device(se_amd_ctx *ctx, const float *d_values, size_t B,
                               const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1,
                               uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status, void *stream)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.encrypt_asym(d_values, B, d_seeds, d_c0, d_c1, d_ntt_pte, d_pte, d_status,
                               as_stream(stream));
}

int se_amd_encode_ntt_device(se_amd_ctx *ctx, const float *d_values, size_t B, uint32_t *d_out,
                             int64_t *d_pte, uint8_t *d_status, void *stream)
{
    if (!ctx || !d_out) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.encode_ntt(d_values, B, d_out, d_pte, d_status, as_stream(stream));
}

int se_amd_encode_device(se_amd_ctx *ctx, const float *d_values, size_t B, int64_t *d_out,
                         uint8_t *d_status, void *stream)
{
    if (!ctx || !d_out || !d_values) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.encode_ntt(d_values, B, nullptr, d_out, d_status, as_stream(stream));
}

int se_amd_ntt_device(se_amd_ctx *ctx, size_t prime, uint32_t *d_polys, size_t count, void *stream)
{
    if (!ctx || !d_polys || prime >= ctx->c.hp.nprimes) return SE_ERR_INVALD_ARGUMENT;
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    SEAMD_HIP(seamd::launch_ntt_polys(ctx->c.dp, ctx->c.dt, (int)prime, d_polys, nullptr, count,
                                      as_stream(stream)));
    return SE_SUCCESS;
}

int se_amd_intt_device(se_amd_ctx *ctx, size_t prime, uint32_t *d_polys, size_t count, void *stream)
{
    if (!ctx || !d_polys || prime >= ctx->c.hp.nprimes) return SE_ERR_INVALD_ARGUMENT;
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    // in place: natural-order result written back over the input polynomial
    SEAMD_HIP(seamd::launch_decrypt_decode(ctx->c.dp, ctx->c.dt, d_polys, nullptr, 1, (int)prime,
                                           nullptr, d_polys, nullptr, count, as_stream(stream)));
    return SE_SUCCESS;
}

int se_amd_decrypt_decode_device(se_amd_ctx *ctx, const uint32_t *d_c0, const uint32_t *d_c1,
                                 size_t B, size_t prime, uint32_t *d_dec_ntt, uint32_t *d_pt,
                                 float *d_values, void *stream)
{
    if (!ctx || !d_c0 || !d_c1 || prime >= ctx->c.hp.nprimes) return SE_ERR_INVALD_ARGUMENT;
    if (!ctx->c.have_sk)
    {
        seamd::set_last_error("decrypt needs the secret key (se_amd_set_secret_key)");
        return SE_ERR_NO_KEY;
    }
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    SEAMD_HIP(seamd::launch_decrypt_decode(ctx->c.dp, ctx->c.dt, d_c0, d_c1,
                                           (uint32_t)ctx->c.hp.nprimes, (int)prime, d_dec_ntt, d_pt,
                                           d_values, B, as_stream(stream)));
    return SE_SUCCESS;
}

int se_amd_prng_blocks_device(se_amd_ctx *ctx, const uint8_t *d_seeds, const uint64_t *d_ctrs,
                              uint8_t *d_out, size_t outlen, size_t count, void *stream)
{
    if (!ctx || !d_seeds || !d_ctrs || !d_out) return SE_ERR_INVALD_ARGUMENT;
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    SEAMD_HIP(seamd::launch_prng_blocks(d_seeds, d_ctrs, d_out, (uint32_t)outlen, (uint32_t)count,
                                        as_stream(stream)));
    return SE_SUCCESS;
}

int se_amd_sample_uniform_device(se_amd_ctx *ctx, const uint8_t *d_seeds, const uint64_t *d_ctr_in,
                                 size_t B, uint32_t *d_out, uint64_t *d_ctr_out, void *stream)
{
    if (!ctx || !d_seeds || !d_out) return SE_ERR_INVALD_ARGUMENT;
    return ctx->c.sample_uniform(d_seeds, d_ctr_in, B, d_out, d_ctr_out, as_stream(stream));
}

int se_amd_sample_ternary_device(se_amd_ctx *ctx, const uint8_t *d_seeds, size_t B, int8_t *d_codes,
                                 uint64_t *d_ctr_out, void *stream)
{
    if (!ctx || !d_seeds || !d_codes) return SE_ERR_INVALD_ARGUMENT;
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    seamd::TernaryArgs ta{d_seeds, d_codes, d_ctr_out, (uint32_t)ctx->c.hp.n, (uint32_t)B, nullptr,
                          (uint32_t)ctx->c.num_cus, ctx->c.debug_flags};
    SEAMD_HIP(seamd::launch_sample_ternary(ta, as_stream(stream)));
    return SE_SUCCESS;
}

int se_amd_sample_cbd_device(se_amd_ctx *ctx, const uint8_t *d_seeds, const uint64_t *d_ctr_base,
                             size_t B, size_t blocks_per_ct, int8_t *d_out, void *stream)
{
    if (!ctx || !d_seeds || !d_out) return SE_ERR_INVALD_ARGUMENT;
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    seamd::CbdArgs ca{d_seeds, d_ctr_base, d_out, (uint32_t)blocks_per_ct, (uint32_t)B};
    SEAMD_HIP(seamd::launch_sample_cbd(ca, as_stream(stream)));
    return SE_SUCCESS;
}

// sample.c:61-87: 2 bits per coefficient, first coefficient in the two MOST significant bits
void se_amd_pack_ternary_host(const int8_t *codes, size_t n, uint8_t *packed)
{
    memset(packed, 0, n / 4);
    for (size_t i = 0; i < n; i++)
        packed[i / 4] |= (uint8_t)((codes[i] & 3) << (6 - 2 * (i % 4)));
}

int se_amd_word_ops_device(se_amd_ctx *ctx, size_t prime, int op, const uint64_t *d_a, const uint64_t *d_b,
                           const uint64_t *d_c, uint32_t *d_out, size_t count, void *stream)
{
    if (!ctx || !d_a || !d_out || prime >= ctx->c.hp.nprimes || op < 0 || op > 12) return SE_ERR_INVALD_ARGUMENT;
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    SEAMD_HIP(seamd::launch_word_ops(ctx->c.dp, (int)prime, op, d_a, d_b, d_c, d_out, count, as_stream(stream)));
    return SE_SUCCESS;
}

int se_amd_set_profiling(se_amd_ctx *ctx, int enabled)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    ctx->c.profiling = enabled != 0;
    return SE_SUCCESS;
}

int se_amd_stage_ms(se_amd_ctx *ctx, float *ms, uint64_t *launches, int reset)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    (void)hipSetDevice(ctx->c.device);
    ctx->c.collect_events();
    for (int i = 0; i < seamd::kStageCount; i++)
    {
        if (ms) ms[i] = ctx->c.stage_ms[i];
        if (launches) launches[i] = ctx->c.stage_launches[i];
        if (reset)
        {
            ctx->c.stage_ms[i]       = 0;
            ctx->c.stage_launches[i] = 0;
        }
    }
    return SE_SUCCESS;
}

int se_amd_set_reject_list_capacity(se_amd_ctx *ctx, uint32_t cap)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    ctx->c.rej_cap     = cap;
    ctx->c.rows_cap    = 0;  // force re-allocation with the new stride
    return SE_SUCCESS;
}

int se_amd_set_debug_flags(se_amd_ctx *ctx, uint32_t flags)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    ctx->c.debug_flags = flags;
    return SE_SUCCESS;
}

int se_amd_ifft_table_sha256(se_amd_ctx *ctx, char out_hex[65])
{
    if (!ctx || !out_hex) return SE_ERR_INVALD_ARGUMENT;
    // the table the kernels read, copied back from the DEVICE: W[t] = (re, im), t = 0 .. n-1, little-endian doubles
    const size_t n = ctx->c.hp.n;
    std::vector<double> w(2 * n);
    SEAMD_HIP(hipSetDevice(ctx->c.device));
    SEAMD_HIP(hipMemcpy(w.data(), ctx->c.dt.ifft_w, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
    seamd::Sha256 h;
    h.update(w.data(), 2 * n * sizeof(double));
    const std::string hex = h.hex();
    memcpy(out_hex, hex.c_str(), 65);
    return SE_SUCCESS;
}

int se_amd_host_tables(size_t degree, size_t nprimes, uint32_t *q, uint32_t *const_ratio,
                       double *scale, uint16_t *index_map, double *ifft_w, uint32_t *ntt_rw,
                       uint32_t *intt_rw)
{
    seamd::HostParams hp;
    if (seamd::host_params_init(hp, degree, nprimes) != 0)
    {
        seamd::set_last_error("unsupported parameter set (degree, nprimes)");
        return SE_ERR_INVALD_ARGUMENT;
    }
    const size_t n = hp.n;
    for (size_t j = 0; j < hp.nprimes; j++)
    {
        if (q) q[j] = hp.q[j];
        if (const_ratio) const_ratio[2 * j] = hp.cr_lo[j], const_ratio[2 * j + 1] = hp.cr_hi[j];
    }
    if (scale) *scale = hp.scale;
    if (index_map)
    {
        std::vector<uint16_t> map, inv;
        seamd::host_index_map(hp, map, inv);
        memcpy(index_map, map.data(), n * sizeof(uint16_t));
    }
    if (ifft_w)
    {
        std::vector<double> w;
        seamd::host_ifft_twiddles(hp, w);
        memcpy(ifft_w, w.data(), 2 * n * sizeof(double));
    }
    for (size_t j = 0; j < hp.nprimes; j++)
    {
        std::vector<uint32_t> rw;
        if (ntt_rw)
        {
            seamd::host_ntt_root_pairs(hp, j, rw);
            memcpy(ntt_rw + 2 * n * j, rw.data(), 2 * n * sizeof(uint32_t));
        }
        if (intt_rw)
        {
            seamd::host_intt_root_pairs(hp, j, rw);
            memcpy(intt_rw + 2 * n * j, rw.data(), 2 * n * sizeof(uint32_t));
        }
    }
    return SE_SUCCESS;
}

int se_amd_set_pipeline(se_amd_ctx *ctx, int overlap, int split)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    ctx->c.overlap = overlap != 0;
    ctx->c.split_mode = split;
    return SE_SUCCESS;
}

int se_amd_set_asym_chunks(se_amd_ctx *ctx, size_t chunks)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    ctx->c.asym_chunks = chunks ? chunks : 1;
    return SE_SUCCESS;
}

int se_amd_set_speculation_capacity(se_amd_ctx *ctx, uint32_t cap)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    ctx->c.spec_cap    = cap ? cap : 1;
    ctx->c.rows_cap    = 0;  // force re-allocation with the new stride
    return SE_SUCCESS;
}

int se_amd_reserve(se_amd_ctx *ctx, size_t B)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    return ctx->c.ensure_scratch(B);
}

// ---- host-pointer wrappers ------------------------------------------------------------------
static int host_pipe(se_amd_ctx *ctx, HostPipe **out)
{
    Context &c = ctx->c;
    if (!c.host_pipe)
    {
        HostPipe *hp = new HostPipe();
        int rc       = hp->init(c.device);
        if (rc)
        {
            delete hp;
            return rc;
        }
        c.host_pipe = hp;
    }
    *out = c.host_pipe;
    return 0;
}

static int run_host(se_amd_ctx *ctx, bool asym, const float *values, size_t B,
                    const uint8_t *share_seeds, const uint8_t *seeds, uint32_t *c0, uint32_t *c1,
                    uint32_t *ntt_pte, int64_t *pte, uint8_t *status)
{
    if (!ctx || !values || !seeds || !c0 || (asym && !c1) || (!asym && !share_seeds))
        return SE_ERR_INVALD_ARGUMENT;
    if (B == 0) return SE_SUCCESS;
    if (asym ? !ctx->c.have_pk : !ctx->c.have_sk)
    {
        seamd::set_last_error(asym ? "no public key loaded" : "no secret key loaded");
        return SE_ERR_NO_KEY;
    }
    HostPipe *hp;
    int rc = host_pipe(ctx, &hp);
    if (rc) return rc;
    return hp->run(ctx->c, asym, values, B, share_seeds, seeds, c0, c1, ntt_pte, pte, status);
}

int se_amd_set_host_chunk(se_amd_ctx *ctx, size_t ciphertexts)
{
    if (!ctx) return SE_ERR_INVALD_ARGUMENT;
    HostPipe *hp;
    int rc = host_pipe(ctx, &hp);
    if (rc) return rc;
    hp->chunk_override = ciphertexts;
    return SE_SUCCESS;
}

int se_amd_encrypt_sym_host(se_amd_ctx *ctx, const float *values, size_t B,
                            const uint8_t *share_seeds, const uint8_t *seeds, uint32_t *c0,
                            uint32_t *c1, uint32_t *ntt_pte, int64_t *pte, uint8_t *status)
{
    return run_host(ctx, false, values, B, share_seeds, seeds, c0, c1, ntt_pte, pte, status);
}

int se_amd_encrypt_asym_host(se_amd_ctx *ctx, const float *values, size_t B, const uint8_t *seeds,
                             uint32_t *c0, uint32_t *c1, uint32_t *ntt_pte, int64_t *pte,
                             uint8_t *status)
{
    return run_host(ctx, true, values, B, nullptr, seeds, c0, c1, ntt_pte, pte, status);
}

// =============================================================================================
// Layer 1: the reference API.  Static singletons like seal_embedded.c:18-22 -- one parameter set
// per process, not re-entrant.
// =============================================================================================
static Parms g_parms;
static SE_PTRS g_ptrs;
static SE_PARMS g_se_parms;
static std::vector<uint32_t> g_roots;  // [np][n] one-shot NTT roots (SE_PTRS::ntt_roots_ptr per prime)
static se_amd_ctx *g_ctx = nullptr;
// further contexts, one per additional device of $SE_AMD_DEVICES: se_encrypt_batch shards a batch
// over all of them (contiguous blocks, one host thread and one PCIe link per device)
static std::vector<se_amd_ctx *> g_more;

static const char *data_path()
{
    const char *p = getenv("SE_AMD_DATA_PATH");
    return p ? p : "adapter_output_data";  // device/CMakeLists.txt:115,285
}

static void drop_contexts()
{
    if (g_ctx)
    {
        se_amd_destroy(g_ctx);
        g_ctx = nullptr;
    }
    for (se_amd_ctx *x : g_more) se_amd_destroy(x);
    g_more.clear();
}

// $SE_AMD_DEVICES = "all" or a comma list (the batched entry shards over them); otherwise the single
// device $SE_AMD_DEVICE (default 0).  Indices are HIP device ordinals, i.e. positions in the
// process's visible-device list: $HIP_VISIBLE_DEVICES / $ROCR_VISIBLE_DEVICES remap them as usual.
static std::vector<int> device_list()
{
    std::vector<int> devices;
    if (const char *list = getenv("SE_AMD_DEVICES"))
    {
        if (strcmp(list, "all") == 0)
        {
            int count = 0;
            if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
            for (int d = 0; d < count; d++) devices.push_back(d);
        }
        else
        {
            for (const char *p = list; *p;)
            {
                char *end;
                long d = strtol(p, &end, 10);
                if (end == p) break;
                devices.push_back((int)d);
                p = (*end == ',') ? end + 1 : end;
            }
        }
    }
    if (devices.empty()) devices.push_back(getenv("SE_AMD_DEVICE") ? atoi(getenv("SE_AMD_DEVICE")) : 0);
    return devices;
}

SE_PARMS *se_setup_custom(size_t degree, size_t nprimes, const ZZ *modulus_vals, const ZZ *ratios,
                          double scale, EncryptType encrypt_type)
{
    // Same sequence as seal_embedded.c:24-83: flags, pool, pointer carving, ckks_setup, ckks_setup_s
    // -- all through the lower surface of this library (se_lower.cpp).
    if (modulus_vals && ratios)
    {
        // unreachable in the reference (ckks_setup_custom recurses forever, ckks_common.c:90);
        // encrypting under a different chain than the caller asked for would be silent corruption
        fprintf(stderr, "Error! se_setup_custom: custom modulus chains are not supported\n");
        exit(1);
    }
    drop_contexts();
    if (g_ptrs.conj_vals) free(g_ptrs.conj_vals);
    g_ptrs = SE_PTRS();
    delete_parameters(&g_parms);

    const bool asym       = (encrypt_type == SE_ASYM_ENCR);
    const size_t n        = degree;
    g_parms.scale         = scale;  // overwritten by the parameter set (parameters.c:190-226)
    g_parms.is_asymmetric = asym;
    g_parms.pk_from_file  = 1;
    g_parms.sample_s      = 0;
    g_parms.small_u       = 1;
    g_parms.small_s       = 1;
    g_se_parms.parms      = &g_parms;
    g_se_parms.se_ptrs    = &g_ptrs;

    ZZ *mempool = asym ? ckks_mempool_setup_asym(n) : ckks_mempool_setup_sym(n);
    if (asym)
        ckks_set_ptrs_asym(n, mempool, &g_ptrs);
    else
        ckks_set_ptrs_sym(n, mempool, &g_ptrs);
    ckks_setup(n, nprimes, g_ptrs.index_map_ptr, &g_parms);
    if (!asym) ckks_setup_s(&g_parms, NULL, NULL, g_ptrs.ternary);  // sk_<n>.dat -> SE_PTRS::ternary

    const std::vector<int> devices = device_list();
    for (size_t i = 0; i < devices.size(); i++)
    {
        se_amd_ctx *x = nullptr;
        int rc        = se_amd_create(&x, degree, nprimes, devices[i]);
        if (rc == SE_SUCCESS)
            rc = asym ? se_amd_load_keys_from_dir(x, data_path(), 1)
                      : se_amd_set_secret_key(x, reinterpret_cast<const uint8_t *>(g_ptrs.ternary));
        if (rc != SE_SUCCESS)
        {
            // error convention of the reference: print and exit (ckks_sym.c:68-72, fileops.c:60-91)
            fprintf(stderr, "Error! se_setup failed: %s\n", se_amd_last_error());
            exit(1);
        }
        if (i == 0)
            g_ctx = x;
        else
            g_more.push_back(x);
    }
    g_roots.assign(nprimes * n, 0);
    for (size_t j = 0; j < nprimes; j++)
    {
        std::vector<uint32_t> rw;
        seamd::host_ntt_root_pairs(g_ctx->c.hp, j, rw);
        for (size_t i = 0; i < n; i++) g_roots[j * n + i] = rw[2 * i];
    }
    return &g_se_parms;
}

SE_PARMS *se_setup(size_t degree, size_t nprimes, double scale, EncryptType encrypt_type)
{
    return se_setup_custom(degree, nprimes, NULL, NULL, scale, encrypt_type);
}

SE_PARMS *se_setup_default(EncryptType encrypt_type)
{
    return se_setup(4096, 3, pow(2, 25), encrypt_type);  // seal_embedded.c:90-96
}

static void fill_seed(uint8_t *dst, const uint8_t *src)
{
    if (src)
    {
        memcpy(dst, src, SE_PRNG_SEED_BYTE_COUNT);
        return;
    }
    ssize_t got = getrandom(dst, SE_PRNG_SEED_BYTE_COUNT, 0);  // rng.h:45-53
    if (got != SE_PRNG_SEED_BYTE_COUNT)
    {
        fprintf(stderr, "Error! getrandom failed\n");
        exit(1);
    }
}

bool se_encrypt_seeded(uint8_t *shareable_seed, uint8_t *seed, SEND_FNCT_PTR network_send_function,
                       void *v, size_t vlen_bytes, bool print, SE_PARMS *se_parms)
{
    if (!se_parms || !se_parms->parms || !se_parms->se_ptrs || !g_ctx) return false;
    Parms *parms   = se_parms->parms;
    SE_PTRS *ptrs  = se_parms->se_ptrs;
    const size_t n = parms->coeff_count, np = parms->nprimes;

    // seal_embedded.c:108-111: only the first copy_size bytes are cleared/overwritten, so a
    // shorter input leaves the tail of a previous call in place (kept: "stale values" quirk).
    size_t copy = (n / 2) * sizeof(ZZ);
    if (vlen_bytes < copy) copy = vlen_bytes;
    memset(ptrs->values, 0, copy);
    memcpy(ptrs->values, v, copy);

    uint8_t s_share[64], s_priv[64];
    fill_seed(s_share, shareable_seed);
    fill_seed(s_priv, seed);

    // one GPU call for the whole ciphertext (all primes), then the reference's per-prime delivery
    std::vector<uint32_t> c0(np * n), c1(np * n), ntt_pte(np * n);
    std::vector<int64_t> pte(n);
    std::vector<int8_t> codes(parms->is_asymmetric ? n : 0);
    // the private seed, m + e and u do not outlive the call on the host (SE_PTRS keeps what the
    // reference keeps there)
    struct Wipe
    {
        uint8_t *seed;
        std::vector<int64_t> &pte;
        std::vector<int8_t> &codes;
        std::vector<uint32_t> &ntt_pte;
        ~Wipe()
        {
            explicit_bzero(seed, 64);
            explicit_bzero(pte.data(), pte.size() * sizeof(int64_t));
            explicit_bzero(codes.data(), codes.size());
            explicit_bzero(ntt_pte.data(), ntt_pte.size() * sizeof(uint32_t));
        }
    } wipe{s_priv, pte, codes, ntt_pte};
    int rc;
    if (parms->is_asymmetric)
        rc = se_amd_encrypt_asym_host(g_ctx, ptrs->values, 1, s_priv, c0.data(), c1.data(),
                                      ntt_pte.data(), pte.data(), nullptr);
    else
        rc = se_amd_encrypt_sym_host(g_ctx, ptrs->values, 1, s_share, s_priv, c0.data(), c1.data(),
                                     ntt_pte.data(), pte.data(), nullptr);
    if (rc < 0)
    {
        fprintf(stderr, "Error! se_encrypt: %s\n", se_amd_last_error());
        exit(1);
    }
    if (rc > 0) return false;  // encode overflow (seal_embedded.c:115-118)
    // SE_PTRS as the reference leaves it: m + e in the low half of conj_vals; u and e1 of this call
    memcpy(ptrs->conj_vals_int_ptr, pte.data(), n * sizeof(int64_t));
    if (parms->is_asymmetric)
    {
        if (g_ctx->c.fetch_asym_randomness(codes.data(), ptrs->e1_ptr) != 0)
        {
            fprintf(stderr, "Error! se_encrypt: %s\n", se_amd_last_error());
            exit(1);
        }
        se_amd_pack_ternary_host(codes.data(), n, reinterpret_cast<uint8_t *>(ptrs->ternary));
    }

    const char *quirk  = getenv("SE_AMD_REFERENCE_C1_ALIAS");
    const bool alias   = !parms->is_asymmetric && quirk && quirk[0] == '1';
    reset_primes(parms);
    for (size_t j = 0; j < np; j++)
    {
        memcpy(ptrs->ntt_roots_ptr, g_roots.data() + j * n, n * sizeof(ZZ));
        memcpy(ptrs->c0_ptr, c0.data() + j * n, n * sizeof(ZZ));
        if (parms->is_asymmetric) memcpy(ptrs->ntt_pte_ptr, ntt_pte.data() + j * n, n * sizeof(ZZ));
        memcpy(ptrs->c1_ptr, (alias ? ntt_pte.data() : c1.data()) + j * n, n * sizeof(ZZ));
        if (print)
        {
            // print_poly("c0: ", ...) of seal_embedded.c:160-163 with the default SE_PRINT_SMALL
            // (defines.h:49-50, util_print.h:478-488): the first 8 values, then "... }"
            const ZZ *polys[2]   = {ptrs->c0_ptr, ptrs->c1_ptr};
            const char *names[2] = {"c0: ", "c1: "};
            for (int k = 0; k < 2; k++)
            {
                const size_t shown = n < 8 ? n : 8;
                printf("%s : { ", names[k]);
                for (size_t i = 0; i < shown; i++) printf(i + 1 < n ? "%u, " : "%u ", polys[k][i]);
                printf(shown == n ? "}\n" : "... }\n");
            }
        }
        if (network_send_function)
        {
            size_t nbytes = n * sizeof(ZZ);
            if (network_send_function(ptrs->c0_ptr, nbytes) != nbytes) return false;
            if (network_send_function(ptrs->c1_ptr, nbytes) != nbytes) return false;
        }
        if (j + 1 < np) next_modulus(parms);  // ckks_next_prime_* (seal_embedded.c:206-212)
    }
    return true;
}

bool se_encrypt(SEND_FNCT_PTR network_send_function, void *v, size_t vlen_bytes, bool print,
                SE_PARMS *se_parms)
{
    return se_encrypt_seeded(NULL, NULL, network_send_function, v, vlen_bytes, print, se_parms);
}

void se_cleanup(SE_PARMS *se_parms)
{
    drop_contexts();
    g_roots.clear();
    if (se_parms && se_parms->parms) delete_parameters(se_parms->parms);  // seal_embedded.c:227
    // the pool is freed through conj_vals, which points at its start (seal_embedded.c:230-232)
    if (se_parms && se_parms->se_ptrs && se_parms->se_ptrs->conj_vals)
    {
        free(se_parms->se_ptrs->conj_vals);
        *se_parms->se_ptrs = SE_PTRS();
    }
    if (se_parms) se_parms->parms = 0;  // seal_embedded.c:234
}

int se_encrypt_batch(const SE_PARMS *se_parms, const float *values, size_t B,
                     const uint8_t *share_seeds, const uint8_t *seeds, uint32_t *c0, uint32_t *c1)
{
    if (!se_parms || !se_parms->parms || !g_ctx) return SE_ERR_INVALD_ARGUMENT;
    const bool asym = se_parms->parms->is_asymmetric;
    auto run        = [&](se_amd_ctx *ctx, size_t lo, size_t cnt) -> int {
        const size_t n = ctx->c.hp.n, np = ctx->c.hp.nprimes;
        uint32_t *o1   = c1 ? c1 + lo * np * n : nullptr;
        if (asym)
            return se_amd_encrypt_asym_host(ctx, values + lo * (n / 2), cnt, seeds + lo * 64,
                                            c0 + lo * np * n, o1, nullptr, nullptr, nullptr);
        return se_amd_encrypt_sym_host(ctx, values + lo * (n / 2), cnt,
                                       share_seeds ? share_seeds + lo * 64 : nullptr, seeds + lo * 64,
                                       c0 + lo * np * n, o1, nullptr, nullptr, nullptr);
    };
    const size_t ndev = 1 + g_more.size();
    if (ndev == 1 || B < 2 * ndev) return run(g_ctx, 0, B);
    if (!values || !seeds || !c0) return SE_ERR_INVALD_ARGUMENT;

    // contiguous block per device, one host thread each (each drives its own PCIe link)
    std::vector<int> rcs(ndev, 0);
    std::vector<std::string> errs(ndev);
    std::vector<std::thread> workers;
    for (size_t d = 0; d < ndev; d++)
    {
        const size_t lo = B * d / ndev, hi = B * (d + 1) / ndev;
        se_amd_ctx *ctx = d == 0 ? g_ctx : g_more[d - 1];
        workers.emplace_back([&, d, lo, hi, ctx] {
#ifdef SEAMD_TEST_HOOKS
            // fault injection for the re-run path: $SE_AMD_INJECT_SHARD_FAILURE = shard index.  Compiled
            // into the test build of the library only (make testhooks), never into the product.
            const char *inj = getenv("SE_AMD_INJECT_SHARD_FAILURE");
            if (inj && *inj && (size_t)atol(inj) == d)
            {
                rcs[d]  = SE_ERR_NO_DEVICE;
                errs[d] = "injected shard failure";
                return;
            }
#endif
            rcs[d] = run(ctx, lo, hi - lo);
            if (rcs[d] < 0) errs[d] = se_amd_last_error();  // thread-local: carry it to the caller
        });
    }
    for (auto &w : workers) w.join();
    // A shard lost to a DEVICE failure (a HIP error, a device that went away) is re-run on a device whose
    // own shard succeeded in the first pass (units are independent, so a lost device
    // costs time, not results).  Caller errors (invalid argument, missing key) fail on every device alike
    // and are returned as they are.  `healthy` is fixed after the first pass: a device that failed once is
    // never chosen as a fallback, whatever happened to its shard afterwards.
    std::vector<char> healthy(ndev);
    for (size_t d = 0; d < ndev; d++) healthy[d] = rcs[d] >= 0;
    int failed = 0;
    for (size_t d = 0; d < ndev; d++)
    {
        if (rcs[d] < 0)
        {
            const bool device_fault = rcs[d] == SE_ERR_HIP || rcs[d] == SE_ERR_NO_DEVICE;
            const size_t lo = B * d / ndev, hi = B * (d + 1) / ndev;
            int rc = rcs[d];
            for (size_t k = 1; device_fault && k < ndev && rc < 0; k++)
            {
                const size_t h = (d + k) % ndev;
                if (!healthy[h]) continue;
                se_amd_ctx *ctx = h == 0 ? g_ctx : g_more[h - 1];
                fprintf(stderr, "se_encrypt_batch: shard %zu (ciphertexts %zu..%zu) failed (%s); re-running it on "
                                "device slot %zu\n", d, lo, hi, errs[d].c_str(), h);
                rc = run(ctx, lo, hi - lo);
            }
            if (rc < 0)
            {
                seamd::set_last_error(errs[d]);
                return rcs[d];
            }
            rcs[d] = rc;
        }
        failed += rcs[d];
    }
    return failed;
}

}  // extern "C"
