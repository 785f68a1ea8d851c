// se_lower.cpp -- the reference's LOWER surface (include/seal_embedded_amd_lower.h) under its own
// names and prototypes, every operator served by the gfx950 kernels as a host-pointer batch of one.
//
// Mirrors the calling contracts of /root/reference/device/lib:
//   parameters.c:26-230, modulus.c:23-56          set_parms_ckks / next_modulus / set_modulus   (host tables)
//   ckks_common.c:32-274                          index map, ckks_setup, ckks_encode_base, reduce_*
//   fft.c:47-213, ntt.c:24-189, intt.c:26-222     root tables (host) and transforms (GPU)
//   rng.h:40-114, sample.c:39-356                 PRNG + samplers (GPU)
//   ckks_sym.c:29-312, ckks_asym.c:30-299         pool carving, init, per-prime encrypt, gen_pk
//   fileops.c:140-204                             load_sk / load_pki
//
// One GPU context per polynomial degree (with the longest prime chain of that degree: the chain of
// a shorter parameter set is its prefix, parameters.c:129-174) is created on first use.  Error
// convention of the reference: message + exit(1) (ckks_sym.c:68-72, fileops.c:60-91).
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/seal_embedded_amd.h"
#include "se_context.h"

using seamd::Context;

namespace {

std::recursive_mutex g_mu;  // the reference is single-threaded; calls are serialised here

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "Error! %s: %s\n", what, se_amd_last_error());
    exit(1);
}

#define LOWER_HIP(call)                                                             \
    do                                                                              \
    {                                                                               \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess)                                                      \
        {                                                                           \
            fprintf(stderr, "Error! HIP error %d (%s) in %s\n", (int)e__, hipGetErrorString(e__), #call); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

size_t max_primes_for(size_t n)
{
    switch (n)
    {
        case 1024:
        case 2048: return 1;
        case 4096: return 3;
        case 8192: return 6;
        case 16384: return 13;
        default: return 0;
    }
}

// Per-degree GPU state: the context plus a device slab carved into the operand buffers of one call.
struct Lower
{
    se_amd_ctx *h = nullptr;
    size_t n      = 0;
    uint8_t *slab = nullptr;
    size_t slab_bytes = 0;
    double *cplx_in = nullptr, *cplx_out = nullptr;  // [n][2]
    int64_t *i64     = nullptr;                      // [n]
    uint32_t *u32[8] = {};                           // [n] each
    int8_t *i8[2]    = {};                           // [n] each
    uint8_t *packed  = nullptr;                      // [n/4]
    uint8_t *seed    = nullptr;                      // [64]
    uint64_t *ctr    = nullptr;                      // [2]
    uint32_t *fail   = nullptr;                      // [1]
    uint8_t *bytes   = nullptr;                      // growable raw buffer (prng_fill_buffer)
    size_t bytes_cap = 0;
    std::vector<uint32_t> roots_cache[seamd::kMaxPrimes];   // ntt_roots_initialize output per prime (host table)
    Context &c() { return h->c; }

    // ---- symmetric fast path (round 5) -------------------------------------------------------------------------
    // ckks_sym_init (ckks_sym.c:181-197) receives the shareable seed: from there on a_j of EVERY prime is a function of
    // (seed, start counter of prime j), and the start counter of prime j >= 1 lies in a narrow window (se_context.cpp,
    // small_batch_plan).  So the init call launches, asynchronously, the uniform sampler of prime 0 at counter 0 and of
    // every prime j >= 1 for every start counter of its window ("virtual ciphertexts", one wave each, ONE launch) -- the
    // primes of a ciphertext run side by side instead of as np sequential 121-permutation chains, as se_encrypt_seeded
    // does.  The first ckks_encode_encrypt_sym call (ckks_sym.c:199) that finds its PRNG state in that table follows
    // the counter chain through it on the host, launches the per-prime kernel for this AND the remaining primes, and
    // every per-prime call returns its prime from the precomputed results WHEN ITS INPUTS MATCH what they were computed
    // from: shareable seed and counter, plaintext (m + e) and packed key compared byte for byte, no ep_small.  Anything
    // else takes the per-prime chain as before.  PRNG counters are left exactly as the reference leaves them.
    struct SymSpec
    {
        bool armed = false, fetched = false;
        uint8_t seed[64] = {};
        // The chain follows the caller's prime order: it starts at the prime the Parms stood at when ckks_sym_init was
        // called (the reference's bench_sym never rewinds: its iterations start at primes 0, 2, 1, 0, ...) and walks
        // next_modulus's wrap-around order.  Step k of the chain = prime (first + k) mod chain; base / count / offset
        // are indexed by STEP.
        uint32_t first = 0, chain = 0;                   // prime of step 0; primes in the caller's chain
        uint32_t nprimes = 0, total = 0;                 // steps covered; virtual ciphertexts (index 0: step 0 @ 0)
        uint64_t base[seamd::kMaxPrimes]  = {};
        uint32_t count[seamd::kMaxPrimes] = {}, offset[seamd::kMaxPrimes] = {};
        uint32_t prime_of_step(uint32_t k) const { return (first + k) % chain; }
        uint32_t step_of_prime(uint32_t j) const { return (j + chain - first) % chain; }
        size_t cap = 0;                                  // virtual ciphertexts the device buffers hold
        uint8_t *d_meta = nullptr;                       // [cap] x (64 seed + 8 ctr + 8 ctrout + 1 prime), carved below
        uint8_t *d_seeds = nullptr, *d_prime = nullptr;
        uint64_t *d_ctr = nullptr, *d_ctrout = nullptr;
        uint32_t *d_rows = nullptr;                      // [cap][n]
        std::vector<uint64_t> h_ctrout;
        hipStream_t st = nullptr, cp = nullptr;
        hipEvent_t ev_sampled = nullptr, ev_kernel[seamd::kMaxPrimes] = {}, ev_copied[seamd::kMaxPrimes] = {};
        // precomputed primes
        bool pre[seamd::kMaxPrimes]          = {};
        uint64_t pre_start[seamd::kMaxPrimes] = {}, pre_end[seamd::kMaxPrimes] = {};
        std::vector<int64_t> h_pte;                      // the plaintext d_pte holds (empty: none)
        std::vector<uint8_t> h_key;                      // the packed key d_key holds
        // one slab (one memset wipes it): d_pte [n] int64 | d_out [np][3][n]: c0 | ntt_pte | s_save | d_key [n/4]
        uint8_t *d_slab = nullptr;
        size_t slab_bytes = 0;
        int64_t *d_pte  = nullptr;
        uint8_t *d_key  = nullptr;
        uint32_t *d_out = nullptr;
        bool dirty = false;                              // secret-bearing copies exist since the last wipe
        uint32_t *h_stage = nullptr;                     // pinned [np][4][n]: a | c0 | ntt_pte | s_save
    } sym;
};

std::map<size_t, Lower *> g_lower;

// prng_fill_buffer's own device scratch (a pure PRNG call needs no parameter set)
struct PrngScratch
{
    int device     = 0;
    bool ready     = false;
    uint8_t *seed  = nullptr;   // [64]
    uint64_t *ctr  = nullptr;   // [1]
    uint8_t *bytes = nullptr;
    size_t cap     = 0;
} g_prng;

// The operand slabs hold secrets between calls (packed secret key, u, e / e1, PRNG seeds, m + e):
// se_amd_lower_shutdown() -- also run at process exit -- wipes and frees them and destroys the per-degree
// contexts.  The surface stays usable afterwards (state is rebuilt on the next call).
void lower_shutdown()
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    for (auto &kv : g_lower)
    {
        Lower *L = kv.second;
        if (hipSetDevice(L->c().device) == hipSuccess)
        {
            (void)hipDeviceSynchronize();
            if (L->slab) (void)hipMemset(L->slab, 0, L->slab_bytes);
            if (L->bytes) (void)hipMemset(L->bytes, 0, L->bytes_cap);
            (void)hipDeviceSynchronize();
            if (L->slab) (void)hipFree(L->slab);
            if (L->bytes) (void)hipFree(L->bytes);
            Lower::SymSpec &S = L->sym;
            const size_t n = L->n, np = L->c().hp.nprimes;
            if (S.d_rows) (void)hipMemset(S.d_rows, 0, S.cap * n * sizeof(uint32_t));
            if (S.d_meta) (void)hipMemset(S.d_meta, 0, S.cap * 88);
            if (S.d_slab) (void)hipMemset(S.d_slab, 0, S.slab_bytes);
            (void)hipDeviceSynchronize();
            void *dev[] = {S.d_rows, S.d_meta, S.d_slab};
            for (void *q : dev)
                if (q) (void)hipFree(q);
            if (S.h_stage)
            {
                explicit_bzero(S.h_stage, np * 4 * n * sizeof(uint32_t));
                (void)hipHostFree(S.h_stage);
            }
            if (!S.h_pte.empty()) explicit_bzero(S.h_pte.data(), S.h_pte.size() * 8);
            if (!S.h_key.empty()) explicit_bzero(S.h_key.data(), S.h_key.size());
            if (S.st) (void)hipStreamDestroy(S.st);
            if (S.cp) (void)hipStreamDestroy(S.cp);
            if (S.ev_sampled) (void)hipEventDestroy(S.ev_sampled);
            for (auto &e : S.ev_kernel)
                if (e) (void)hipEventDestroy(e);
            for (auto &e : S.ev_copied)
                if (e) (void)hipEventDestroy(e);
        }
        se_amd_destroy(L->h);
        delete L;
    }
    g_lower.clear();
    if (g_prng.ready && hipSetDevice(g_prng.device) == hipSuccess)
    {
        (void)hipMemset(g_prng.seed, 0, 64);
        (void)hipMemset(g_prng.ctr, 0, 8);
        if (g_prng.bytes) (void)hipMemset(g_prng.bytes, 0, g_prng.cap);
        (void)hipDeviceSynchronize();
        (void)hipFree(g_prng.seed), (void)hipFree(g_prng.ctr);
        if (g_prng.bytes) (void)hipFree(g_prng.bytes);
    }
    g_prng = PrngScratch();
}

void register_shutdown()
{
    static bool registered = false;
    if (!registered)
    {
        registered = true;
        atexit(lower_shutdown);
    }
}

Lower &lower_for_degree(size_t n)
{
    auto it = g_lower.find(n);
    if (it != g_lower.end()) return *it->second;
    const size_t np = max_primes_for(n);
    if (!np)
    {
        fprintf(stderr, "Error! unsupported polynomial degree %zu (parameters.c:176-230)\n", n);
        exit(1);
    }
    register_shutdown();
    Lower *L = new Lower();
    L->n     = n;
    int dev  = getenv("SE_AMD_DEVICE") ? atoi(getenv("SE_AMD_DEVICE")) : 0;
    if (se_amd_create(&L->h, n, np, dev) != SE_SUCCESS) die("GPU context");
    if (L->c().ensure_scratch(1) != 0) die("GPU scratch");
    size_t total = 16 * n * 2 + 8 * n + 8 * 4 * n + 2 * n + n / 4 + 64 + 16 + 16;
    LOWER_HIP(hipSetDevice(L->c().device));
    LOWER_HIP(hipMalloc((void **)&L->slab, total));
    L->slab_bytes = total;
    uint8_t *p  = L->slab;
    L->cplx_in  = (double *)p, p += 16 * n;
    L->cplx_out = (double *)p, p += 16 * n;
    L->i64      = (int64_t *)p, p += 8 * n;
    for (auto &u : L->u32) u = (uint32_t *)p, p += 4 * n;
    for (auto &e : L->i8) e = (int8_t *)p, p += n;
    L->packed = p, p += n / 4;
    L->seed   = p, p += 64;
    L->ctr    = (uint64_t *)p, p += 16;
    L->fail   = (uint32_t *)p;
    g_lower[n] = L;
    return *L;
}

// context for a Parms; checks that the caller's chain is the default one the tables were built for
Lower &lower_for(const Parms *parms)
{
    if (!parms || !parms->moduli)
    {
        fprintf(stderr, "Error! Parms not set up (call ckks_setup first)\n");
        exit(1);
    }
    Lower &L = lower_for_degree(parms->coeff_count);
    if (parms->nprimes > L.c().hp.nprimes || parms->curr_modulus_idx >= parms->nprimes ||
        parms->moduli[parms->curr_modulus_idx].value != L.c().hp.q[parms->curr_modulus_idx])
    {
        fprintf(stderr, "Error! modulus chain is not a default parameter set (parameters.c:129-230)\n");
        exit(1);
    }
    return L;
}

inline int prime_of(const Parms *parms) { return (int)parms->curr_modulus_idx; }

void up(void *dst, const void *src, size_t bytes) { LOWER_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); }
void down(void *dst, const void *src, size_t bytes) { LOWER_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); }

void put_prng(Lower &L, const SE_PRNG *prng)
{
    up(L.seed, prng->seed, 64);
    up(L.ctr, &prng->counter, 8);
}

const char *data_path()
{
    const char *p = getenv("SE_AMD_DATA_PATH");
    return p ? p : "adapter_output_data";  // device/CMakeLists.txt:115,285
}

void read_image(const char *path, size_t bytes, void *dst)
{
    FILE *f = fopen(path, "rb");
    if (!f)
    {
        fprintf(stderr, "Error! cannot open %s: %s\n", path, strerror(errno));  // fileops.c:60-91
        exit(1);
    }
    size_t got = fread(dst, 1, bytes, f);
    fclose(f);
    if (got != bytes)
    {
        fprintf(stderr, "Error! short read on %s (%zu of %zu bytes)\n", path, got, bytes);
        exit(1);
    }
}

// counter wrap of prng_fill_buffer (rng.h:85-90)
void after_draws(SE_PRNG *prng, uint64_t before)
{
    if (prng->counter < before)
    {
        printf("PRNG counter overflowed.");
        printf("Re-randomizing seed and resetting counter to 0.\n");
        prng_randomize_reset(prng, NULL);
    }
}

void ternary_codes_to_packed(const int8_t *codes, size_t n, void *packed)
{
    se_amd_pack_ternary_host(codes, n, (uint8_t *)packed);  // sample.c:61-87
}

}  // namespace

extern "C" {

// ---- rng.h -------------------------------------------------------------------------------------
void prng_randomize_reset(SE_PRNG *prng, uint8_t *seed_in)
{
    prng->counter = 0;
    if (seed_in)
    {
        memcpy(prng->seed, seed_in, SE_PRNG_SEED_BYTE_COUNT);
        return;
    }
    ssize_t got = getrandom(prng->seed, SE_PRNG_SEED_BYTE_COUNT, 0);
    if (got != SE_PRNG_SEED_BYTE_COUNT)
    {
        fprintf(stderr, "Error! getrandom failed\n");
        exit(1);
    }
}

void prng_clear(SE_PRNG *prng)
{
    memset(prng->seed, 0, SE_PRNG_SEED_BYTE_COUNT);
    prng->counter = 0;
}

void se_amd_lower_shutdown(void) { lower_shutdown(); }

// A pure PRNG call needs no parameter set: it has device scratch of its own (seed, counter, growable
// output buffer on $SE_AMD_DEVICE) instead of borrowing -- or creating -- a per-degree context.
void prng_fill_buffer(size_t byte_count, SE_PRNG *prng, void *buffer)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (byte_count > 0xFFFFFFFFu)
    {
        // one call = ONE SHAKE256 squeeze (rng.h:78-91); the kernel counts output bytes in 32 bits
        fprintf(stderr, "Error! prng_fill_buffer: %zu bytes in one call (limit 4 GiB - 1)\n", byte_count);
        exit(1);
    }
    if (!g_prng.ready)
    {
        register_shutdown();
        g_prng.device = getenv("SE_AMD_DEVICE") ? atoi(getenv("SE_AMD_DEVICE")) : 0;
        LOWER_HIP(hipSetDevice(g_prng.device));
        LOWER_HIP(hipMalloc((void **)&g_prng.seed, 64));
        LOWER_HIP(hipMalloc((void **)&g_prng.ctr, 8));
        g_prng.ready = true;
    }
    LOWER_HIP(hipSetDevice(g_prng.device));
    if (byte_count > g_prng.cap)
    {
        if (g_prng.bytes)
        {
            (void)hipMemset(g_prng.bytes, 0, g_prng.cap);   // PRNG output of a secret seed
            (void)hipFree(g_prng.bytes);
        }
        g_prng.bytes = nullptr, g_prng.cap = 0;
        LOWER_HIP(hipMalloc((void **)&g_prng.bytes, byte_count ? byte_count : 1));
        g_prng.cap = byte_count ? byte_count : 1;
    }
    up(g_prng.seed, prng->seed, 64);
    up(g_prng.ctr, &prng->counter, 8);
    if (byte_count)
    {
        LOWER_HIP(seamd::launch_prng_blocks(g_prng.seed, g_prng.ctr, g_prng.bytes, (uint32_t)byte_count, 1, nullptr));
        down(buffer, g_prng.bytes, byte_count);
    }
    const uint64_t before = prng->counter;
    prng->counter++;
    after_draws(prng, before);
}

// ---- parameters.h / modulus.h (host tables, A1) ------------------------------------------------
bool set_modulus(const ZZ q, Modulus *mod)
{
    // modulus.c:23-56 is a table of floor(2^64/q) for the tabulated primes (0 = not in the table);
    // for odd q > 1, floor(2^64/q) == floor((2^64 - 1)/q)
    if (!seamd::host_known_prime(q)) return 0;
    const uint64_t ratio = ~(uint64_t)0 / q;
    set_modulus_custom(q, (ZZ)(ratio >> 32), (ZZ)ratio, mod);
    return 1;
}

void set_modulus_custom(const ZZ q, ZZ hw, ZZ lw, Modulus *mod)
{
    mod->value          = q;
    mod->const_ratio[1] = hw;
    mod->const_ratio[0] = lw;
}

void set_parms_ckks(size_t degree, size_t nprimes, Parms *parms)
{
    seamd::HostParams hp;
    if (seamd::host_params_init(hp, degree, nprimes) != 0)
    {
        fprintf(stderr, "Error! unsupported parameter set (degree %zu, %zu primes)\n", degree, nprimes);
        exit(1);
    }
    parms->coeff_count = degree;
    parms->logn        = hp.logn;
    parms->nprimes     = nprimes;
    parms->moduli      = (Modulus *)calloc(nprimes, sizeof(Modulus));  // SE_USE_MALLOC (parameters.c:108-112)
    if (!parms->moduli)
    {
        fprintf(stderr, "Error! Allocation failed. Exiting...\n");
        exit(1);
    }
    for (size_t j = 0; j < nprimes; j++) set_modulus_custom(hp.q[j], hp.cr_hi[j], hp.cr_lo[j], &parms->moduli[j]);
    parms->curr_modulus_idx = 0;
    parms->curr_modulus     = &parms->moduli[0];
    parms->scale            = hp.scale;
}

void delete_parameters(Parms *parms)
{
    if (parms && parms->moduli)
    {
        free(parms->moduli);
        parms->moduli = 0;
    }
}

void reset_primes(Parms *parms)
{
    parms->curr_modulus_idx = 0;
    parms->curr_modulus     = &parms->moduli[0];
}

bool next_modulus(Parms *parms)
{
    bool ret = 1;
    if (parms->curr_modulus_idx + 1 >= parms->nprimes)
    {
        parms->curr_modulus_idx = 0;  // parameters.c:78-84: wraps and reports the end of the chain
        ret                     = 0;
    }
    else
        parms->curr_modulus_idx++;
    parms->curr_modulus = &parms->moduli[parms->curr_modulus_idx];
    return ret;
}

// ---- ckks_common.h -----------------------------------------------------------------------------
void ckks_calc_index_map(const Parms *parms, uint16_t *index_map)
{
    seamd::HostParams hp;
    hp.n    = parms->coeff_count;
    hp.logn = parms->logn;
    std::vector<uint16_t> map, inv;
    seamd::host_index_map(hp, map, inv);
    memcpy(index_map, map.data(), hp.n * sizeof(uint16_t));
}

void ckks_setup(size_t degree, size_t nprimes, uint16_t *index_map, Parms *parms)
{
    set_parms_ckks(degree, nprimes, parms);
    if (index_map) ckks_calc_index_map(parms, index_map);  // SE_INDEX_MAP_PERSIST (user_defines.h:94)
}

void ckks_setup_custom(size_t degree, size_t nprimes, const ZZ *modulus_vals, const ZZ *ratios,
                       uint16_t *index_map, Parms *parms)
{
    if (!modulus_vals || !ratios)
    {
        ckks_setup(degree, nprimes, index_map, parms);
        return;
    }
    fprintf(stderr, "Error! custom modulus chains are not supported (the reference's ckks_setup_custom "
                    "recurses forever, ckks_common.c:90)\n");
    exit(1);
}

void ckks_reset_primes(Parms *parms) { reset_primes(parms); }

bool ckks_encode_base(const Parms *parms, const flpt *values, size_t values_len, uint16_t *index_map,
                      se_complex *ifft_roots, se_complex *conj_vals)
{
    (void)ifft_roots;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L       = lower_for_degree(parms->coeff_count);
    const size_t n = L.n;
    // the scatter of ckks_common.c:139-153 (data movement through the CALLER's index map; untouched
    // slots keep their previous contents, as in the reference)
    double *cv = reinterpret_cast<double *>(conj_vals);
    std::vector<uint16_t> own;
    if (!index_map)
    {
        own.resize(n);
        ckks_calc_index_map(parms, own.data());
        index_map = own.data();
    }
    for (size_t i = 0; i < values_len; i++)
    {
        const uint16_t i1 = index_map[i], i2 = index_map[i + n / 2];
        cv[2 * i1] = (double)values[i], cv[2 * i1 + 1] = 0.0;
        cv[2 * i2] = (double)values[i], cv[2 * i2 + 1] = 0.0;
    }
    LOWER_HIP(hipSetDevice(L.c().device));
    up(L.cplx_in, cv, 16 * n);
    const uint32_t none = 0xFFFFFFFFu;
    up(L.fail, &none, 4);
    seamd::FftArgs fa{L.cplx_in, L.cplx_out, L.i64, L.fail, 1};
    seamd::DevParams dp = L.c().dp;
    dp.n_inv            = parms->scale / (double)n;  // ckks_common.c:183 (the caller's scale)
    LOWER_HIP(seamd::launch_fft_polys(dp, L.c().dt, fa, 1, nullptr));
    uint32_t bad = none;
    down(&bad, L.fail, 4);
    // the reference converts in place, index by index, and stops at the first value that does not
    // fit (ckks_common.c:187-206): int64 results below that index, IFFT output above it
    down(cv, L.cplx_out, 16 * n);
    const size_t upto = bad == none ? n : (size_t)bad;
    if (upto) down(cv, L.i64, 8 * upto);
    if (bad != none)
    {
        printf("Error! Value at index %u is possibly too large.\n", bad);
        return false;
    }
    return true;
}

static void reduce_generic(const Parms *parms, const int64_t *pte, const int8_t *e, ZZ *out, bool add)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L       = lower_for(parms);
    const size_t n = L.n;
    LOWER_HIP(hipSetDevice(L.c().device));
    if (pte) up(L.i64, pte, 8 * n);
    if (e) up(L.i8[0], e, n);
    if (add) up(L.u32[0], out, 4 * n);
    LOWER_HIP(seamd::launch_reduce_poly(L.c().dp, prime_of(parms), pte ? L.i64 : nullptr, e ? L.i8[0] : nullptr,
                                        L.u32[0], add, n, nullptr));
    down(out, L.u32[0], 4 * n);
}

void print_ckks_mempool_size(size_t n, bool sym)
{
    // same text as ckks_common.c:336-380 for the default configuration (values buffer inside the pool)
    size_t pool = sym ? ckks_get_mempool_size_sym(n) : ckks_get_mempool_size_asym(n);
    const char *txt[2] = {"\nTotal memory requirement (incl. values buffer)  :",
                          "\nTotal memory requirement (without values buffer):"};
    for (int i = 0; i < 2; i++)
    {
        const size_t bytes = pool * sizeof(ZZ);
        if (bytes / 1024)
            printf("%s %zu KB\n", txt[i], bytes / 1024);
        else
            printf("%s %zu bytes\n", txt[i], bytes);
        printf("\t( i.e. [(degree = %zu) * (sizeof(ZZ) = %zu bytes) = ", n, sizeof(ZZ));
        if (n * sizeof(ZZ) / 1024)
            printf("%zu KB] * %0.4f )\n\n", n * sizeof(ZZ) / 1024, pool / (double)n);
        else
            printf("%zu bytes] * %0.4f )\n\n", n * sizeof(ZZ), pool / (double)n);
        pool -= n / 2;
    }
}

void reduce_set_pte(const Parms *parms, const int64_t *conj_vals_int, ZZ *out)
{
    reduce_generic(parms, conj_vals_int, nullptr, out, false);
}
void reduce_add_pte(const Parms *parms, const int64_t *conj_vals_int, ZZ *out)
{
    reduce_generic(parms, conj_vals_int, nullptr, out, true);
}
void reduce_set_e_small(const Parms *parms, const int8_t *e, ZZ *out)
{
    reduce_generic(parms, nullptr, e, out, false);
}
void reduce_add_e_small(const Parms *parms, const int8_t *e, ZZ *out)
{
    reduce_generic(parms, nullptr, e, out, true);
}

// ---- fft.h -------------------------------------------------------------------------------------
void calc_ifft_roots(size_t n, size_t logn, se_complex *ifft_roots)
{
    seamd::HostParams hp;
    hp.n = n, hp.logn = logn;
    std::vector<double> w;
    seamd::host_ifft_twiddles(hp, w);  // (cos, -sin) of 2 pi bitrev(i) / 2n: fft.c:59-67
    memcpy(ifft_roots, w.data(), 16 * n);
}

void calc_fft_roots(size_t n, size_t logn, se_complex *roots)
{
    calc_ifft_roots(n, logn, roots);
    double *r = reinterpret_cast<double *>(roots);
    for (size_t i = 0; i < n; i++) r[2 * i + 1] = -r[2 * i + 1];  // fft.c:47-57: the conjugates
}

static void fft_generic(se_complex *vec, size_t n, int mode)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for_degree(n);
    LOWER_HIP(hipSetDevice(L.c().device));
    up(L.cplx_in, vec, 16 * n);
    seamd::FftArgs fa{L.cplx_in, L.cplx_out, nullptr, nullptr, mode};
    LOWER_HIP(seamd::launch_fft_polys(L.c().dp, L.c().dt, fa, 1, nullptr));
    down(vec, L.cplx_out, 16 * n);
}

void ifft_inpl(se_complex *vec, size_t n, size_t logn, const se_complex *roots)
{
    (void)logn;
    (void)roots;
    fft_generic(vec, n, 0);
}

void fft_inpl(se_complex *vec, size_t n, size_t logn, const se_complex *roots)
{
    (void)logn;
    (void)roots;
    fft_generic(vec, n, 2);
}

// ---- ntt.h / intt.h ----------------------------------------------------------------------------
static void roots_generic(const Parms *parms, ZZ *roots, bool inverse)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for(parms);
    std::vector<uint32_t> rw;
    if (inverse)
        seamd::host_intt_root_pairs(L.c().hp, parms->curr_modulus_idx, rw);
    else
    {
        // the per-prime encrypt calls hand this table out every time (ckks_sym.c:270, one-shot roots): keep it
        std::vector<uint32_t> &cache = L.roots_cache[parms->curr_modulus_idx];
        if (cache.empty())
        {
            seamd::host_ntt_root_pairs(L.c().hp, parms->curr_modulus_idx, rw);
            cache.resize(L.n);
            for (size_t i = 0; i < L.n; i++) cache[i] = rw[2 * i];
        }
        memcpy(roots, cache.data(), L.n * sizeof(ZZ));
        return;
    }
    for (size_t i = 0; i < L.n; i++) roots[i] = rw[2 * i];
}

void ntt_roots_initialize(const Parms *parms, ZZ *ntt_roots)
{
    if (ntt_roots) roots_generic(parms, ntt_roots, false);
}
void intt_roots_initialize(const Parms *parms, ZZ *intt_roots)
{
    if (intt_roots) roots_generic(parms, intt_roots, true);
}

void ntt_inpl(const Parms *parms, const ZZ *ntt_roots, ZZ *vec)
{
    (void)ntt_roots;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for(parms);
    LOWER_HIP(hipSetDevice(L.c().device));
    up(L.u32[0], vec, 4 * L.n);
    LOWER_HIP(seamd::launch_ntt_polys(L.c().dp, L.c().dt, prime_of(parms), L.u32[0], nullptr, 1, nullptr));
    down(vec, L.u32[0], 4 * L.n);
}

void intt_inpl(const Parms *parms, const ZZ *intt_roots, ZZ *vec)
{
    (void)intt_roots;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for(parms);
    LOWER_HIP(hipSetDevice(L.c().device));
    up(L.u32[0], vec, 4 * L.n);
    LOWER_HIP(seamd::launch_decrypt_decode(L.c().dp, L.c().dt, L.u32[0], nullptr, 1, prime_of(parms), nullptr,
                                           L.u32[0], nullptr, 1, nullptr));
    down(vec, L.u32[0], 4 * L.n);
}

// ---- sample.h ----------------------------------------------------------------------------------
// a for the current prime from (seed, counter); returns with the device copy in L.u32[0]
static void uniform_on_device(Lower &L, const Parms *parms, SE_PRNG *prng)
{
    Context &c       = L.c();
    if (L.sym.armed && L.sym.st) LOWER_HIP(hipStreamSynchronize(L.sym.st));   // the speculation shares c.d_rej / c.d_spec
    const uint32_t j = (uint32_t)prime_of(parms);
    put_prng(L, prng);
    seamd::UniformArgs ua{L.seed, L.ctr, L.ctr + 1, L.u32[0], c.d_rej, c.rej_cap, 1, j, j + 1, 1,
                          c.d_spec, c.spec_cap, 0, c.debug_flags, nullptr, j, 0};
    LOWER_HIP(seamd::launch_sample_uniform(c.dp, ua, nullptr));
    const uint64_t before = prng->counter;
    down(&prng->counter, L.ctr + 1, 8);
    after_draws(prng, before);
}

void sample_poly_uniform(const Parms *parms, SE_PRNG *prng, ZZ *poly)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for(parms);
    LOWER_HIP(hipSetDevice(L.c().device));
    uniform_on_device(L, parms, prng);
    down(poly, L.u32[0], 4 * L.n);
}

void expand_poly_ternary(const ZZ *src, const Parms *parms, ZZ *dest)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for(parms);
    LOWER_HIP(hipSetDevice(L.c().device));
    up(L.packed, src, L.n / 4);
    LOWER_HIP(seamd::launch_expand_ternary(L.packed, L.u32[0], parms->curr_modulus->value, (uint32_t)L.n, nullptr));
    down(dest, L.u32[0], 4 * L.n);
}

void expand_poly_ternary_inpl(ZZ *poly, const Parms *parms) { expand_poly_ternary(poly, parms, poly); }

// sample.c:61-111: accessors of the packed form (data layout, not arithmetic)
void set_small_poly_idx(size_t idx, uint8_t val_in, ZZ *poly)
{
    uint8_t *b     = reinterpret_cast<uint8_t *>(poly);
    const int sh   = 6 - 2 * (int)(idx % 4);
    b[idx / 4]     = (uint8_t)((b[idx / 4] & ~(0x3 << sh)) | ((val_in & 0x3) << sh));
}

uint8_t get_small_poly_idx(const ZZ *poly, size_t idx)
{
    return (reinterpret_cast<const uint8_t *>(poly)[idx / 4] >> (6 - 2 * (idx % 4))) & 0x3;
}

ZZ get_small_poly_idx_expanded(const ZZ *poly, size_t idx, ZZ q)
{
    const ZZ v = get_small_poly_idx(poly, idx);
    return v + (v == 0 ? q : 0) - 1;
}

void convert_poly_ternary(const ZZ *src, const Parms *parms, ZZ *dest)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for(parms);
    LOWER_HIP(hipSetDevice(L.c().device));
    up(L.u32[0], src, 4 * L.n);
    LOWER_HIP(seamd::launch_ternary_words(L.u32[0], L.u32[1], nullptr, parms->curr_modulus->value, (uint32_t)L.n, 0,
                                          nullptr));
    down(dest, L.u32[1], 4 * L.n);
}

void convert_poly_ternary_inpl(ZZ *poly, const Parms *parms) { convert_poly_ternary(poly, parms, poly); }

void sample_poly_ternary(const Parms *parms, SE_PRNG *prng, ZZ *poly)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L       = lower_for(parms);
    const size_t n = L.n;
    const ZZ q     = parms->curr_modulus->value;
    LOWER_HIP(hipSetDevice(L.c().device));
    // one 4n-byte block, then one 4-byte block per rejected word (>= 0xFFFFFFFE: 2^-31 per word)
    put_prng(L, prng);
    const uint32_t zero = 0;
    up(L.fail, &zero, 4);
    LOWER_HIP(seamd::launch_prng_blocks(L.seed, L.ctr, (uint8_t *)L.u32[0], (uint32_t)(4 * n), 1, nullptr));
    LOWER_HIP(seamd::launch_ternary_words(L.u32[0], L.u32[1], L.fail, q, (uint32_t)n, 1, nullptr));
    uint32_t nrej = 0;
    down(&nrej, L.fail, 4);
    down(poly, L.u32[1], 4 * n);
    uint64_t before = prng->counter;
    prng->counter++;
    after_draws(prng, before);
    if (nrej)
    {
        std::vector<uint32_t> words(n);
        down(words.data(), L.u32[0], 4 * n);
        for (size_t i = 0; i < n; i++)
        {
            uint32_t w = words[i];
            if (w < 0xFFFFFFFEu) continue;
            while (w >= 0xFFFFFFFEu) prng_fill_buffer(4, prng, &w);
            up(L.u32[2], &w, 4);
            LOWER_HIP(seamd::launch_ternary_words(L.u32[2], L.u32[3], L.fail, q, 1, 1, nullptr));
            down(&poly[i], L.u32[3], 4);
        }
    }
}

void sample_small_poly_ternary_prng_96(PolySizeType n, SE_PRNG *prng, ZZ *poly)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for_degree(n);
    LOWER_HIP(hipSetDevice(L.c().device));
    put_prng(L, prng);
    seamd::TernaryArgs ta{L.seed, L.i8[0], L.ctr + 1, (uint32_t)n, 1, L.ctr};
    LOWER_HIP(seamd::launch_sample_ternary(ta, nullptr));
    std::vector<int8_t> codes(n);
    down(codes.data(), L.i8[0], n);
    const uint64_t before = prng->counter;
    down(&prng->counter, L.ctr + 1, 8);
    after_draws(prng, before);
    ternary_codes_to_packed(codes.data(), n, poly);
}

void sample_poly_cbd_generic_prng_16(PolySizeType n, SE_PRNG *prng, int8_t *poly)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for_degree(n);
    LOWER_HIP(hipSetDevice(L.c().device));
    put_prng(L, prng);
    seamd::CbdArgs ca{L.seed, L.ctr, L.i8[0], (uint32_t)(n / 16), 1};
    LOWER_HIP(seamd::launch_sample_cbd(ca, nullptr));
    down(poly, L.i8[0], n);
    const uint64_t before = prng->counter;
    prng->counter += n / 16;
    after_draws(prng, before);
}

void sample_add_poly_cbd_generic_inpl_prng_16(int64_t *poly, PolySizeType n, SE_PRNG *prng)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for_degree(n);
    LOWER_HIP(hipSetDevice(L.c().device));
    put_prng(L, prng);
    up(L.i64, poly, 8 * n);
    seamd::CbdArgs ca{L.seed, L.ctr, L.i8[0], (uint32_t)(n / 16), 1};
    LOWER_HIP(seamd::launch_sample_cbd(ca, nullptr));
    LOWER_HIP(seamd::launch_add_small(L.i64, L.i8[0], n, nullptr));
    down(poly, L.i64, 8 * n);
    const uint64_t before = prng->counter;
    prng->counter += n / 16;
    after_draws(prng, before);
}

// ---- fileops.h ---------------------------------------------------------------------------------
void load_sk(const Parms *parms, ZZ *s)
{
    char path[512];
    snprintf(path, sizeof(path), "%s/sk_%zu.dat", data_path(), parms->coeff_count);
    read_image(path, parms->coeff_count / 4, s);
}

void load_pki(size_t i, const Parms *parms, ZZ *pki)
{
    char path[512];
    snprintf(path, sizeof(path), "%s/pk%zu_ntt_%zu_%u.dat", data_path(), i, parms->coeff_count,
             parms->curr_modulus->value);
    read_image(path, parms->coeff_count * sizeof(ZZ), pki);
}

// ---- ckks_sym.h --------------------------------------------------------------------------------
// default configuration (user_defines.h:68-129): IFFT on the fly, one-shot NTT roots, persistent
// index map, persistent s, values inside the pool
size_t ckks_get_mempool_size_sym(size_t degree)
{
    const size_t n = degree;
    return 4 * n + n + n / 2 + n / 16 + n / 2;
}

static ZZ *pool_alloc(size_t zz)
{
    ZZ *mempool = (ZZ *)calloc(zz, sizeof(ZZ));
    if (!mempool)
    {
        printf("Error! Allocation failed. Exiting...\n");
        exit(1);
    }
    return mempool;
}

ZZ *ckks_mempool_setup_sym(size_t degree) { return pool_alloc(ckks_get_mempool_size_sym(degree)); }

void ckks_set_ptrs_sym(size_t degree, ZZ *mempool, SE_PTRS *se_ptrs)
{
    const size_t n             = degree;
    se_ptrs->conj_vals         = (se_complex *)mempool;
    se_ptrs->conj_vals_int_ptr = (int64_t *)mempool;
    se_ptrs->c1_ptr            = &mempool[2 * n];
    se_ptrs->c0_ptr            = &mempool[3 * n];
    se_ptrs->ntt_pte_ptr       = &mempool[2 * n];  // the reference's alias (ckks_sym.c:86-88)
    se_ptrs->ifft_roots        = 0;
    se_ptrs->ntt_roots_ptr     = &mempool[4 * n];
    se_ptrs->index_map_ptr     = (uint16_t *)&mempool[5 * n];
    se_ptrs->ternary           = &mempool[5 * n + n / 2];
    se_ptrs->values            = (flpt *)&mempool[5 * n + n / 2 + n / 16];
    se_ptrs->e1_ptr            = 0;
}

void ckks_setup_s(const Parms *parms, uint8_t *seed, SE_PRNG *prng, ZZ *s)
{
    if (parms->sample_s)
    {
        prng_randomize_reset(prng, seed);
        sample_small_poly_ternary_prng_96(parms->coeff_count, prng, s);
    }
    else
        load_sk(parms, s);
}

// ---- symmetric fast path (Lower::SymSpec) ----
static const size_t kSymSpecMaxBytes = (size_t)128 << 20;   // rows of the virtual ciphertexts

// The fast path keeps secret-bearing copies between the calls of ONE ciphertext: m + e and the packed key on the host
// (the byte-for-byte guards) and on the device, the per-prime results in d_out and in the pinned staging.  They are
// wiped as soon as the chain they belong to is over -- when its last precomputed prime has been delivered, and when
// ckks_sym_init arms the next ciphertext (a chain abandoned half-way) -- not only at lower_shutdown.  Precondition:
// nothing on S.st reads them any more (the caller waited for the last kernel); the device memsets are ordered on S.cp
// behind the staging copies.
static void sym_spec_wipe(Lower &L)
{
    Lower::SymSpec &S = L.sym;
    const size_t n = L.n, np = L.c().hp.nprimes;
    for (auto &p : S.pre) p = false;
    if (!S.dirty) return;                          // nothing uploaded or computed since the last wipe
    if (!S.h_pte.empty()) explicit_bzero(S.h_pte.data(), S.h_pte.size() * 8);
    if (!S.h_key.empty()) explicit_bzero(S.h_key.data(), S.h_key.size());
    S.h_pte.clear(), S.h_key.clear();
    if (S.cp && S.d_slab) (void)hipMemsetAsync(S.d_slab, 0, S.slab_bytes, S.cp);
    // staging rows per prime: a | c0 (the ciphertext: public) | ntt_pte | s_save (NTT(m + e), NTT(s): wiped)
    if (S.h_stage)
        for (size_t pr = 0; pr < np; pr++) explicit_bzero(S.h_stage + (pr * 4 + 2) * n, 2 * n * sizeof(uint32_t));
    S.dirty = false;
}

// windows of the start counters of primes 1 .. np-1 (the estimate of Context::small_batch_plan), as many primes as the
// row budget allows; launches the samplers on S.st.  The caller holds g_mu.
static void sym_spec_arm(Lower &L, const Parms *parms, const SE_PRNG *shareable)
{
    Lower::SymSpec &S = L.sym;
    Context &c        = L.c();
    const size_t n    = L.n;
    S.armed = S.fetched = false;
    for (auto &p : S.pre) p = false;
    if (getenv("SE_AMD_LOWER_SPECULATION") && atoi(getenv("SE_AMD_LOWER_SPECULATION")) == 0) return;
    if (parms->nprimes < 1 || parms->nprimes > c.hp.nprimes) return;
    LOWER_HIP(hipSetDevice(c.device));
    if (!S.st)
    {
        LOWER_HIP(hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking));
        LOWER_HIP(hipStreamCreateWithFlags(&S.cp, hipStreamNonBlocking));
        LOWER_HIP(hipEventCreateWithFlags(&S.ev_sampled, hipEventDisableTiming));
        for (auto &e : S.ev_kernel) LOWER_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : S.ev_copied) LOWER_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const size_t np = c.hp.nprimes;
        S.slab_bytes = 8 * n + np * 3 * n * sizeof(uint32_t) + n / 4;
        LOWER_HIP(hipMalloc((void **)&S.d_slab, S.slab_bytes));
        S.d_pte = (int64_t *)S.d_slab;
        S.d_out = (uint32_t *)(S.d_slab + 8 * n);
        S.d_key = S.d_slab + 8 * n + np * 3 * n * sizeof(uint32_t);
        LOWER_HIP(hipHostMalloc((void **)&S.h_stage, np * 4 * n * sizeof(uint32_t), hipHostMallocDefault));
    }
    else
    {
        LOWER_HIP(hipStreamSynchronize(S.st));   // an earlier speculation nobody consumed
        LOWER_HIP(hipStreamSynchronize(S.cp));
        if (S.dirty)
        {
            sym_spec_wipe(L);                        // what a chain abandoned half-way left behind
            LOWER_HIP(hipStreamSynchronize(S.cp));   // (its memset runs on S.cp; the uploads below do not order against it)
        }
    }
    // plan
    double mu = 0.0, var = 0.0;
    uint32_t total = 1, covered = 1;
    S.first = (uint32_t)parms->curr_modulus_idx, S.chain = (uint32_t)parms->nprimes;
    S.base[0] = 0, S.count[0] = 1, S.offset[0] = 0;
    for (uint32_t j = 1; j < parms->nprimes; j++)
    {
        const double p = (double)(0u - c.dp.bound[S.prime_of_step(j - 1)]) / 4294967296.0;
        mu += 1.0 + (double)n * p / (1.0 - p);
        var += (double)n * p * 1.06;
        const uint64_t h   = (uint64_t)(5.5 * sqrt(var)) + 2;
        const uint64_t mid = (uint64_t)(mu + 0.5);
        const uint32_t cnt = (uint32_t)(2 * h + 1);
        if (((size_t)total + cnt) * n * sizeof(uint32_t) > kSymSpecMaxBytes) break;   // later primes: the plain chain
        S.base[j]   = mid > h ? mid - h : 0;
        S.count[j]  = cnt;
        S.offset[j] = total;
        total += cnt;
        covered = j + 1;
    }
    if (total > S.cap)
    {
        if (S.d_rows) (void)hipFree(S.d_rows);
        if (S.d_meta) (void)hipFree(S.d_meta);
        S.d_rows = nullptr, S.d_meta = nullptr, S.cap = 0;
        LOWER_HIP(hipMalloc((void **)&S.d_rows, (size_t)total * n * sizeof(uint32_t)));
        LOWER_HIP(hipMalloc((void **)&S.d_meta, (size_t)total * 88));
        S.cap      = total;
    }
    // carved for THIS plan's `total` (<= cap), so that one upload of total x 88 bytes covers it
    S.d_seeds  = S.d_meta;
    S.d_ctr    = (uint64_t *)(S.d_meta + (size_t)total * 64);
    S.d_ctrout = S.d_ctr + total;
    S.d_prime  = (uint8_t *)(S.d_ctrout + total);
    if (c.ensure_scratch(1, (size_t)1 + total) != 0) die("GPU scratch");
    // seeds | counters | (end counters) | primes of the virtual ciphertexts: one upload
    std::vector<uint8_t> meta((size_t)total * 88, 0);
    uint64_t *hc = (uint64_t *)(meta.data() + (size_t)total * 64);
    uint8_t *hp  = meta.data() + (size_t)total * 80;
    for (uint32_t j = 0; j < covered; j++)
        for (uint32_t g = 0; g < S.count[j]; g++)
        {
            const uint32_t v = S.offset[j] + g;
            memcpy(meta.data() + (size_t)v * 64, shareable->seed, 64);
            hc[v] = S.base[j] + g;
            hp[v] = (uint8_t)S.prime_of_step(j);
        }
    up(S.d_meta, meta.data(), meta.size());
    explicit_bzero(meta.data(), meta.size());
    seamd::UniformArgs ug{S.d_seeds, S.d_ctr, S.d_ctrout, S.d_rows, c.d_rej + c.rej_cap, c.rej_cap, total,
                          0,         0,       1,          c.d_spec + c.spec_cap, c.spec_cap, 0, c.debug_flags,
                          nullptr,   0,       0,          S.d_prime};
    LOWER_HIP(seamd::launch_sample_uniform(c.dp, ug, S.st));
    LOWER_HIP(hipEventRecord(S.ev_sampled, S.st));
    memcpy(S.seed, shareable->seed, 64);
    S.nprimes = covered;
    S.total   = total;
    S.armed   = true;
}

// row of the table that holds a_j for (this PRNG state, prime j), or -1
static long sym_spec_row(Lower &L, const SE_PRNG *shareable, uint32_t j)
{
    Lower::SymSpec &S = L.sym;
    if (!S.armed || j >= S.chain || memcmp(S.seed, shareable->seed, 64) != 0) return -1;
    const uint32_t k = S.step_of_prime(j);
    if (k >= S.nprimes) return -1;
    const uint64_t g = shareable->counter - S.base[k];   // wraps to a huge value below the window
    if (g >= S.count[k]) return -1;
    return (long)(S.offset[k] + g);
}

void ckks_sym_init(const Parms *parms, uint8_t *share_seed_in, uint8_t *seed_in, SE_PRNG *shareable_prng,
                   SE_PRNG *prng, int64_t *conj_vals_int)
{
    prng_randomize_reset(shareable_prng, share_seed_in);
    prng_randomize_reset(prng, seed_in);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L = lower_for_degree(parms->coeff_count);
    if (parms->moduli && parms->nprimes <= L.c().hp.nprimes && parms->curr_modulus_idx < parms->nprimes)
        sym_spec_arm(L, parms, shareable_prng);
    Lower::SymSpec &S = L.sym;
    if (!S.armed)
    {
        sample_add_poly_cbd_generic_inpl_prng_16(conj_vals_int, parms->coeff_count, prng);
        return;
    }
    // sample_add_poly_cbd_generic_inpl_prng_16 with the sum left in a buffer of its own: m + e stays on the device
    // for the per-prime calls (L.i64 is every operator's scratch)
    const size_t n = L.n;
    put_prng(L, prng);
    S.dirty = true;
    up(S.d_pte, conj_vals_int, 8 * n);
    seamd::CbdArgs ca{L.seed, L.ctr, L.i8[0], (uint32_t)(n / 16), 1};
    LOWER_HIP(seamd::launch_sample_cbd(ca, nullptr));
    LOWER_HIP(seamd::launch_add_small(S.d_pte, L.i8[0], n, nullptr));
    down(conj_vals_int, S.d_pte, 8 * n);   // host-synchronous: the sum is complete before anything on S.st reads it
    S.h_pte.assign(conj_vals_int, conj_vals_int + n);
    const uint64_t before = prng->counter;
    prng->counter += n / 16;
    after_draws(prng, before);
}

void ckks_encode_encrypt_sym(const Parms *parms, const int64_t *conj_vals_int, const int8_t *ep_small,
                             SE_PRNG *shareable_prng, ZZ *s_small, ZZ *ntt_pte, ZZ *ntt_roots, ZZ *c0_s,
                             ZZ *c1, ZZ *s_save, ZZ *c1_save)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L       = lower_for(parms);
    const size_t n = L.n;
    if (!conj_vals_int && !ep_small)
    {
        fprintf(stderr, "Error! ckks_encode_encrypt_sym needs conj_vals_int or ep_small\n");
        exit(1);
    }
    LOWER_HIP(hipSetDevice(L.c().device));
    Lower::SymSpec &S = L.sym;
    const uint32_t j  = (uint32_t)prime_of(parms);
    long row          = ep_small ? -1 : sym_spec_row(L, shareable_prng, j);
    if (row >= 0)
    {
        // ---- fast path: a_j comes from the table the init call started ----
        if (!S.fetched)
        {
            LOWER_HIP(hipEventSynchronize(S.ev_sampled));
            S.h_ctrout.resize(S.total);
            down(S.h_ctrout.data(), S.d_ctrout, (size_t)S.total * 8);
            S.fetched = true;
        }
        const bool same_pte = S.h_pte.size() == n && memcmp(S.h_pte.data(), conj_vals_int, 8 * n) == 0;
        const bool same_key = S.h_key.size() == n / 4 && memcmp(S.h_key.data(), s_small, n / 4) == 0;
        if (!(S.pre[j] && S.pre_start[j] == shareable_prng->counter && same_pte && same_key))
        {
            // (re)compute this prime and the ones the counter chain leads to, from the inputs of THIS call
            LOWER_HIP(hipStreamSynchronize(S.cp));   // staging buffers of an earlier chain may still be in flight
            for (auto &p : S.pre) p = false;
            S.dirty = true;
            if (!same_pte)
            {
                up(S.d_pte, conj_vals_int, 8 * n);
                S.h_pte.assign(conj_vals_int, conj_vals_int + n);
            }
            if (!same_key)
            {
                up(S.d_key, s_small, n / 4);
                S.h_key.assign((const uint8_t *)s_small, (const uint8_t *)s_small + n / 4);
            }
            uint64_t ctr = shareable_prng->counter;
            long r       = row;
            // steps of the chain from this prime on; results are kept per PRIME (pr)
            for (uint32_t k = S.step_of_prime(j); k < S.nprimes && r >= 0; k++)
            {
                const uint32_t pr = S.prime_of_step(k);
                uint32_t *out = S.d_out + (size_t)pr * 3 * n, *stage = S.h_stage + (size_t)pr * 4 * n;
                const uint32_t *a = S.d_rows + (size_t)r * n;
                seamd::LowerSymArgs sa{S.d_key, S.d_pte, nullptr, a, out, out + n, out + 2 * n, (int)pr, 0, 0, 0};
                LOWER_HIP(seamd::launch_lower_sym_prime(L.c().dp, L.c().dt, sa, 1, S.st));
                LOWER_HIP(hipEventRecord(S.ev_kernel[pr], S.st));
                LOWER_HIP(hipStreamWaitEvent(S.cp, S.ev_kernel[pr], 0));
                LOWER_HIP(hipMemcpyAsync(stage, a, 4 * n, hipMemcpyDeviceToHost, S.cp));
                LOWER_HIP(hipMemcpyAsync(stage + n, out, 3 * 4 * n, hipMemcpyDeviceToHost, S.cp));
                LOWER_HIP(hipEventRecord(S.ev_copied[pr], S.cp));
                S.pre[pr]       = true;
                S.pre_start[pr] = ctr;
                S.pre_end[pr]   = S.h_ctrout[r];
                // the next prime starts where this one's redraws stopped
                ctr = S.h_ctrout[r];
                r   = -1;
                if (k + 1 < S.nprimes)
                {
                    const uint64_t g = ctr - S.base[k + 1];
                    if (g < S.count[k + 1]) r = (long)(S.offset[k + 1] + g);
                }
            }
        }
        LOWER_HIP(hipEventSynchronize(S.ev_copied[j]));
        const uint32_t *stage = S.h_stage + (size_t)j * 4 * n;
        // deliveries in the reference's write order, so that aliased buffers end up the same
        memcpy(c1, stage, 4 * n);
        if (c1_save) memcpy(c1_save, stage, 4 * n);
        if (ntt_roots) roots_generic(parms, ntt_roots, false);
        if (s_save) memcpy(s_save, stage + 3 * n, 4 * n);
        memcpy(ntt_pte, stage + 2 * n, 4 * n);
        memcpy(c0_s, stage + n, 4 * n);
        const uint64_t before   = shareable_prng->counter;
        shareable_prng->counter = S.pre_end[j];
        after_draws(shareable_prng, before);
        // the last precomputed prime of the chain has been delivered (its kernel and copies are done, and S.st / S.cp
        // are in order, so every earlier prime's are too): nothing of this ciphertext stays behind
        if (S.step_of_prime(j) + 1 == S.nprimes) sym_spec_wipe(L);
        return;
    }
    // c1 = a <- U (ckks_sym.c:220); the counter moves exactly as the reference's rejection loop
    uniform_on_device(L, parms, shareable_prng);
    up(L.packed, s_small, n / 4);
    // ep_small wins over conj_vals_int, as in the reference (ckks_sym.c:279-283)
    if (ep_small)
        up(L.i8[0], ep_small, n);
    else
        up(L.i64, conj_vals_int, 8 * n);
    seamd::LowerSymArgs sa{L.packed, ep_small ? nullptr : L.i64, ep_small ? L.i8[0] : nullptr, L.u32[0],
                           L.u32[1], L.u32[2], L.u32[3], prime_of(parms), 0, 0, 0};
    LOWER_HIP(seamd::launch_lower_sym_prime(L.c().dp, L.c().dt, sa, 1, nullptr));
    // deliveries in the reference's write order, so that aliased buffers end up the same
    down(c1, L.u32[0], 4 * n);
    if (c1_save) down(c1_save, L.u32[0], 4 * n);
    if (ntt_roots) roots_generic(parms, ntt_roots, false);
    if (s_save) down(s_save, L.u32[3], 4 * n);
    down(ntt_pte, L.u32[2], 4 * n);
    down(c0_s, L.u32[1], 4 * n);
}

bool ckks_next_prime_sym(Parms *parms, ZZ *s)
{
    (void)s;  // small_s: the 2-bit form needs no per-prime conversion (ckks_sym.c:307)
    return next_modulus(parms);
}

// ---- ckks_asym.h -------------------------------------------------------------------------------
size_t ckks_get_mempool_size_asym(size_t degree)
{
    const size_t n = degree;
    return 4 * n + (n + n / 4 + n / 16) + n + n / 2 + n / 2;
}

ZZ *ckks_mempool_setup_asym(size_t degree) { return pool_alloc(ckks_get_mempool_size_asym(degree)); }

void ckks_set_ptrs_asym(size_t degree, ZZ *mempool, SE_PTRS *se_ptrs)
{
    const size_t n             = degree;
    se_ptrs->conj_vals         = (se_complex *)mempool;
    se_ptrs->conj_vals_int_ptr = (int64_t *)mempool;
    se_ptrs->c1_ptr            = &mempool[2 * n];
    se_ptrs->c0_ptr            = &mempool[3 * n];
    se_ptrs->ifft_roots        = 0;
    se_ptrs->ntt_roots_ptr     = &mempool[4 * n];
    se_ptrs->ntt_pte_ptr       = &mempool[5 * n];
    se_ptrs->index_map_ptr     = (uint16_t *)&mempool[6 * n];
    se_ptrs->e1_ptr            = (int8_t *)&mempool[6 * n + n / 2];
    se_ptrs->ternary           = &mempool[6 * n + n / 2 + n / 4];
    se_ptrs->values            = (flpt *)&mempool[6 * n + n / 2 + n / 4 + n / 16];
}

void gen_pk(const Parms *parms, ZZ *s_small, ZZ *ntt_roots, uint8_t *seed, SE_PRNG *shareable_prng,
            ZZ *s_save, int8_t *ep_small, ZZ *ntt_ep, ZZ *pk_c0, ZZ *pk_c1)
{
    prng_randomize_reset(shareable_prng, seed);
    ckks_encode_encrypt_sym(parms, 0, ep_small, shareable_prng, s_small, ntt_ep, ntt_roots, pk_c0, pk_c1,
                            s_save, 0);
}

void ckks_asym_init(const Parms *parms, uint8_t *seed, SE_PRNG *prng, int64_t *conj_vals_int, ZZ *u,
                    int8_t *e1)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const size_t n = parms->coeff_count;
    Lower &L       = lower_for_degree(n);
    LOWER_HIP(hipSetDevice(L.c().device));
    prng_randomize_reset(prng, seed);
    // u, then e0 (added to the plaintext) and e1 from the counters behind u (ckks_asym.c:188-201)
    put_prng(L, prng);
    up(L.i64, conj_vals_int, 8 * n);
    seamd::TernaryArgs ta{L.seed, L.i8[0], L.ctr + 1, (uint32_t)n, 1, L.ctr};
    LOWER_HIP(seamd::launch_sample_ternary(ta, nullptr));
    int8_t *d_err = (int8_t *)L.u32[0];  // 2n bytes: e0 | e1
    seamd::CbdArgs ca{L.seed, L.ctr + 1, d_err, (uint32_t)(2 * (n / 16)), 1};
    LOWER_HIP(seamd::launch_sample_cbd(ca, nullptr));
    LOWER_HIP(seamd::launch_add_small(L.i64, d_err, n, nullptr));
    std::vector<int8_t> codes(n);
    down(codes.data(), L.i8[0], n);
    down(conj_vals_int, L.i64, 8 * n);
    down(e1, d_err + n, n);
    const uint64_t before = prng->counter;
    down(&prng->counter, L.ctr + 1, 8);
    prng->counter += 2 * (n / 16);
    after_draws(prng, before);
    ternary_codes_to_packed(codes.data(), n, u);
}

void ckks_encode_encrypt_asym(const Parms *parms, const int64_t *conj_vals_int, const ZZ *u,
                              const int8_t *e1, ZZ *ntt_roots, ZZ *ntt_u_e1_pte, ZZ *ntt_u_save,
                              ZZ *ntt_e1_save, ZZ *pk_c0, ZZ *pk_c1)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Lower &L       = lower_for(parms);
    const size_t n = L.n;
    if (parms->pk_from_file)
    {
        load_pki(1, parms, pk_c1);
        load_pki(0, parms, pk_c0);
    }
    LOWER_HIP(hipSetDevice(L.c().device));
    up(L.packed, u, n / 4);
    up(L.i8[0], e1, n);
    up(L.i64, conj_vals_int, 8 * n);
    up(L.u32[0], pk_c0, 4 * n);
    up(L.u32[1], pk_c1, 4 * n);
    seamd::LowerAsymArgs aa{L.packed, L.i8[0], L.i64, L.u32[0], L.u32[1], L.u32[2], L.u32[3], L.u32[4],
                            ntt_u_save ? L.u32[5] : nullptr, ntt_e1_save ? L.u32[6] : nullptr, prime_of(parms)};
    LOWER_HIP(seamd::launch_lower_asym_prime(L.c().dp, L.c().dt, aa, 1, nullptr));
    if (ntt_roots) roots_generic(parms, ntt_roots, false);
    if (ntt_u_save) down(ntt_u_save, L.u32[5], 4 * n);
    if (ntt_e1_save) down(ntt_e1_save, L.u32[6], 4 * n);
    down(ntt_u_e1_pte, L.u32[4], 4 * n);
    down(pk_c1, L.u32[3], 4 * n);
    down(pk_c0, L.u32[2], 4 * n);
}

bool ckks_next_prime_asym(Parms *parms, ZZ *u)
{
    (void)u;  // small_u: nothing to convert (ckks_asym.c:296-297)
    return next_modulus(parms);
}

}  // extern "C"
