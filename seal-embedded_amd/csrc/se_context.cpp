// se_context.cpp -- GPU context: table upload, key preparation, scratch, and the kernel chains of
// the whole-path entry points.
//
// Kernel chains (no host synchronisation inside; an auxiliary stream of the context runs the kernels
// that do not depend on each other side by side and is joined back into the caller's stream):
//   symmetric  (ckks_sym.c:181-301):  [k_sample_cbd || k_sample_uniform(a -> c1)] -> k_encode_encrypt,
//                                      or the per-prime software pipeline k_encode_rns / k_ntt_fuse
//                                      beside the per-prime uniform sampler (encrypt_sym below)
//   asymmetric (ckks_asym.c:173-286): k_sample_ternary(u, counter) -> k_sample_cbd(e0|e1 at
//                                      counter base) -> k_encode_encrypt
//   encode-only (BASELINE config 5):  k_encode_encrypt<EncodeOnly>
#include "se_context.h"
#include "se_hostpipe.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

namespace seamd {

static thread_local std::string g_last_error;

void set_last_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error() { return g_last_error; }

namespace {
// temporary device allocation, wiped and freed on every path out of its scope
struct DevTemp
{
    void *p      = nullptr;
    size_t bytes = 0;
    bool wipe    = false;
    hipError_t alloc(size_t n, bool secret = false)
    {
        bytes = n;
        wipe  = secret;
        return hipMalloc(&p, n);
    }
    ~DevTemp()
    {
        if (!p) return;
        if (wipe) (void)hipMemset(p, 0, bytes);
        (void)hipFree(p);
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};
}  // namespace

// Zero a device slab that held secret-dependent data before it is freed (best effort: errors are ignored, the
// free follows either way).
static void wipe_device(void *p, size_t bytes)
{
    if (p && bytes) (void)hipMemset(p, 0, bytes);
}

int hip_fail(hipError_t e, const char *what)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    set_last_error(buf);
    return kErrHip;
}

Context::~Context()
{
    (void)hipSetDevice(device);
    delete host_pipe;
    for (auto &ev : events)
    {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    if (aux_stream) (void)hipStreamDestroy(aux_stream);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_cbd) (void)hipEventDestroy(ev_cbd);
    if (ev_enc) (void)hipEventDestroy(ev_enc);
    if (ev_done) (void)hipEventDestroy(ev_done);
    for (auto &e : ev_prime)
        if (e) (void)hipEventDestroy(e);
    if (spec_stream) (void)hipStreamDestroy(spec_stream);
    if (cand_stream) (void)hipStreamDestroy(cand_stream);
    for (auto &e : ev_cand)
        if (e) (void)hipEventDestroy(e);
    // nothing may still be writing the scratch when it is wiped (the streams above were non-blocking)
    (void)hipDeviceSynchronize();
    // secret-bearing slabs are zeroed before they go back to the allocator: NTT(s), the error polynomials
    // e / e0|e1, the ternary u, the per-ciphertext seeds of the speculation path and `a` (recomputable from the
    // shareable seed, kept out of freed memory all the same)
    const size_t n = hp.n, np = hp.nprimes;
    wipe_device(d_s_hat, 2 * np * n * sizeof(uint32_t));
    wipe_device(d_err, scratch_cap * 2 * n);
    wipe_device(d_ucodes, scratch_cap * n);
    wipe_device(d_sp_seeds, sp_cap * 64);
    wipe_device(d_a, a_cap * np * n * sizeof(uint32_t));
    void *ptrs[] = {d_inv_map, d_ifft_w, d_ntt_rw, d_s_hat, d_pk0, d_pk1, d_intt_rw, d_map, d_gather,
                    d_err,     d_ucodes, d_ctr,    d_rej, d_a,   d_spec, d_general, d_compact,
                    d_sp_seeds, d_sp_ctr, d_sp_ctrout, d_sp_rows, d_sp_fail, d_sp_prime, d_nrej, d_flagged};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
}

int Context::init(size_t n, size_t nprimes, int dev)
{
    int rc = host_params_init(hp, n, nprimes);
    if (rc != 0)
    {
        set_last_error("unsupported parameter set (degree, nprimes)");
        return kErrInvalid;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    {
        set_last_error("no HIP device available: this library has no CPU path");
        return kErrNoDevice;
    }
    if (dev < 0 || dev >= count)
    {
        set_last_error("device index out of range");
        return kErrInvalid;
    }
    device = dev;
    SEAMD_HIP(hipSetDevice(device));
    if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
        num_cus <= 0)
        num_cus = 256;
    // SE_AMD_NUM_CUS=<k>: plan launches as if the device had k CUs (a CPX / partial partition seen from the
    // dispatch logic; tests of the small-device limits).  Never more than the device has.
    if (const char *e = getenv("SE_AMD_NUM_CUS"))
    {
        const int k = atoi(e);
        if (k > 0 && k < num_cus) num_cus = k;
    }
    // SE_AMD_STAGED=0|1 / SE_AMD_SPECULATION=0|1: override the form the dispatch would pick (the thresholds were
    // measured on one 256-CU MI355X and are scaled by the CU count; results are bit-identical either way)
    // (kept in members of their own: se_amd_set_debug_flags assigns the whole debug_flags field)
    if (const char *e = getenv("SE_AMD_STAGED")) staged_mode = atoi(e) ? 1 : 0;
    if (const char *e = getenv("SE_AMD_SPECULATION")) spec_mode = atoi(e) ? 1 : 0;
    dp         = to_dev_params(hp);
    dp.num_cus = (uint32_t)num_cus;
    rej_cap = (uint32_t)(n / 16 > 256 ? n / 16 : 256);
    // rej_cap >= 3x the expected rejections per polynomial; spec_cap ~ mean + >5 sigma of the draws
    // speculation capacity: the helper waves compute this many candidates per polynomial WHILE the
    // chains squeeze n*4/136 blocks, and the chains wait for them -- more than the chains need is
    // pure critical path.  mean + 4 sigma of the draws per polynomial (30-bit primes: reject
    // probability 0.0186; draws beyond the capacity go through the pooled loop), multiple of 16.
    {
        const double mean = (double)n * 0.0186 * 1.02;
        uint32_t cap      = (uint32_t)(mean + 4.0 * sqrt((double)n * 0.0186) + 15.0) & ~15u;
        spec_cap          = n <= 2048 ? 32u : cap;
    }

    std::vector<uint16_t> inv;
    host_index_map(hp, index_map, inv);
    std::vector<double> w;
    host_ifft_twiddles(hp, w);
    // every table is followed by the thread-major copy of its window-0 entries (se_types.h)
    const size_t tl = xform_table_len(n), th = n / 16;
    auto append_thread_major = [&](auto *tab) {   // tab: [tl][2], first n pairs filled
        for (int b = 0; b < 4; b++)
            for (size_t g = 0; g < ((size_t)1 << (3 - b)); g++)
                for (size_t t = 0; t < th; t++)
                {
                    const size_t src = (n >> (b + 1)) + (t << (3 - b)) + g;
                    const size_t dst = n + ((8u >> b) - 1 + g) * th + t;
                    tab[2 * dst] = tab[2 * src], tab[2 * dst + 1] = tab[2 * src + 1];
                }
    };
    w.resize(2 * tl);
    append_thread_major(w.data());
    std::vector<uint32_t> rw_all(2 * tl * nprimes), rw;
    for (size_t j = 0; j < nprimes; j++)
    {
        host_ntt_root_pairs(hp, j, rw);
        // the device table holds (-root mod 2^32, shoup(root)): the forward butterfly then needs no
        // separate negation (ct_butterfly, modarith.cuh)
        for (size_t i = 0; i < n; i++) rw[2 * i] = 0u - rw[2 * i];
        memcpy(rw_all.data() + 2 * tl * j, rw.data(), 2 * n * sizeof(uint32_t));
        append_thread_major(rw_all.data() + 2 * tl * j);
    }
    SEAMD_HIP(hipMalloc((void **)&d_inv_map, n * sizeof(uint16_t)));
    SEAMD_HIP(hipMalloc((void **)&d_ifft_w, w.size() * sizeof(double)));
    SEAMD_HIP(hipMalloc((void **)&d_ntt_rw, rw_all.size() * sizeof(uint32_t)));
    SEAMD_HIP(hipMemcpy(d_inv_map, inv.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
    SEAMD_HIP(hipMemcpy(d_ifft_w, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    SEAMD_HIP(hipMemcpy(d_ntt_rw, rw_all.data(), rw_all.size() * sizeof(uint32_t),
                        hipMemcpyHostToDevice));
    SEAMD_HIP(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    SEAMD_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    SEAMD_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    SEAMD_HIP(hipEventCreateWithFlags(&ev_cbd, hipEventDisableTiming));
    SEAMD_HIP(hipEventCreateWithFlags(&ev_enc, hipEventDisableTiming));
    SEAMD_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    for (size_t j = 0; j < (size_t)kMaxPrimes; j++)
        SEAMD_HIP(hipEventCreateWithFlags(&ev_prime[j], hipEventDisableTiming));
    {
        std::vector<uint32_t> irw_all(2 * n * nprimes), irw;
        for (size_t j = 0; j < nprimes; j++)
        {
            host_intt_root_pairs(hp, j, irw);
            memcpy(irw_all.data() + 2 * n * j, irw.data(), 2 * n * sizeof(uint32_t));
        }
        SEAMD_HIP(hipMalloc((void **)&d_intt_rw, irw_all.size() * sizeof(uint32_t)));
        SEAMD_HIP(hipMemcpy(d_intt_rw, irw_all.data(), irw_all.size() * sizeof(uint32_t),
                            hipMemcpyHostToDevice));
        SEAMD_HIP(hipMalloc((void **)&d_map, n * sizeof(uint16_t)));
        SEAMD_HIP(hipMemcpy(d_map, index_map.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
        dt.intt_rw   = d_intt_rw;
        dt.index_map = d_map;
    }
    {
        std::vector<uint16_t> gather(n);
        // point k = 16 t + e is entry e % 8 of the uint4 at [e / 8][t]
        for (size_t k = 0; k < n; k++)
        {
            const size_t t = k >> 4, e = k & 15;
            gather[((e >> 3) * (n / 16) + t) * 8 + (e & 7)] =
                (uint16_t)sv_slot((uint32_t)(inv[k] & (n / 2 - 1)), (uint32_t)hp.logn);
        }
        SEAMD_HIP(hipMalloc((void **)&d_gather, n * sizeof(uint16_t)));
        SEAMD_HIP(hipMemcpy(d_gather, gather.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
        dt.gather_map = d_gather;
    }
    dt.inv_map = d_inv_map;
    dt.ifft_w  = d_ifft_w;
    dt.ntt_rw  = d_ntt_rw;
    return 0;
}

// The fused kernel's list of declined plaintexts (kernel_args.h, EncArgs::general): 4 bytes per plaintext.
int Context::ensure_general(size_t B)
{
    if (B <= general_cap && d_general) return 0;
    SEAMD_HIP(hipSetDevice(device));
    SEAMD_HIP(hipDeviceSynchronize());
    if (d_general) (void)hipFree(d_general);
    d_general = nullptr, general_cap = 0;
    SEAMD_HIP(hipMalloc((void **)&d_general, (B + 1) * sizeof(uint32_t)));
    general_cap = B;
    return 0;
}

int Context::ensure_scratch(size_t B, size_t rows)
{
    if (int rc = ensure_general(B)) return rc;
    // d_err / d_ucodes / d_ctr are per real ciphertext; the reject lists and speculation rows are
    // also needed by the virtual ciphertexts of the small-batch path, which need nothing else
    if (rows < B) rows = B;
    if (B <= scratch_cap && rows <= rows_cap) return 0;
    SEAMD_HIP(hipSetDevice(device));
    SEAMD_HIP(hipDeviceSynchronize());
    const size_t n = hp.n;
    if (B > scratch_cap)
    {
        wipe_device(d_err, scratch_cap * 2 * n);      // e / e0|e1 and u of earlier calls
        wipe_device(d_ucodes, scratch_cap * n);
        void *old[] = {d_err, d_ucodes, d_ctr, d_compact, d_nrej, d_flagged};
        for (void *p : old)
            if (p) (void)hipFree(p);
        d_err = nullptr, d_ucodes = nullptr, d_ctr = nullptr, d_compact = nullptr, d_nrej = nullptr, d_flagged = nullptr;
        scratch_cap = 0;
        SEAMD_HIP(hipMalloc((void **)&d_err, B * 2 * n));
        SEAMD_HIP(hipMalloc((void **)&d_ucodes, B * n));
        SEAMD_HIP(hipMalloc((void **)&d_ctr, B * sizeof(uint64_t)));
        SEAMD_HIP(hipMalloc((void **)&d_compact, B));
        SEAMD_HIP(hipMalloc((void **)&d_nrej, B * sizeof(uint32_t)));
        SEAMD_HIP(hipMalloc((void **)&d_flagged, (B + 1) * sizeof(uint32_t)));
        scratch_cap = B;
    }
    if (rows > rows_cap)
    {
        if (d_rej) (void)hipFree(d_rej);
        if (d_spec) (void)hipFree(d_spec);
        d_rej = nullptr, d_spec = nullptr;
        rows_cap = 0;
        SEAMD_HIP(hipMalloc((void **)&d_rej, rows * (size_t)(rej_cap ? rej_cap : 1) * sizeof(uint32_t)));
        SEAMD_HIP(hipMalloc((void **)&d_spec, rows * (size_t)spec_cap * sizeof(uint32_t)));
        rows_cap = rows;
    }
    return 0;
}

// Successive calls share the context's scratch and auxiliary streams: order them.  The caller
// holds `mu`.
int Context::begin_call(hipStream_t st)
{
    SEAMD_HIP(hipSetDevice(device));
    if (have_done) SEAMD_HIP(hipStreamWaitEvent(st, ev_done, 0));
    return 0;
}

int Context::end_call(hipStream_t st, int rc)
{
    if (rc != 0)
    {
        // a launch failed part-way: auxiliary streams may be forked and un-joined; drain the device
        // so nothing still refers to the scratch, keep the first error
        const std::string first = last_error();
        (void)hipDeviceSynchronize();
        set_last_error(first);
        return rc;
    }
    SEAMD_HIP(hipEventRecord(ev_done, st));
    have_done = true;
    return 0;
}

int Context::fetch_asym_randomness(int8_t *ucodes, int8_t *e1)
{
    std::lock_guard<std::mutex> lk(mu);
    if (!d_ucodes || !d_err) return kErrInvalid;
    SEAMD_HIP(hipSetDevice(device));
    SEAMD_HIP(hipDeviceSynchronize());
    const size_t n = hp.n;
    SEAMD_HIP(hipMemcpy(ucodes, d_ucodes, n, hipMemcpyDeviceToHost));
    SEAMD_HIP(hipMemcpy(e1, d_err + n, n, hipMemcpyDeviceToHost));
    return 0;
}

// sk arrives 2-bit packed (sk_<n>.dat, fileops.c:140-170).  Expand per prime (sample.c:98-129),
// NTT on the device, keep (NTT(s), shoup) pairs -- ckks_sym.c:255-266 hoisted out of the
// per-ciphertext path.
int Context::set_secret_key(const uint8_t *sk_packed)
{
    std::lock_guard<std::mutex> lk(mu);
    return set_secret_key_impl(sk_packed);
}

// the caller holds `mu`
int Context::set_secret_key_impl(const uint8_t *sk_packed)
{
    const size_t n = hp.n, np = hp.nprimes;
    SEAMD_HIP(hipSetDevice(device));
    std::vector<uint32_t> expanded(np * n);
    // the expanded key never outlives this call, on either side of the bus
    struct Wipe
    {
        std::vector<uint32_t> &v;
        ~Wipe() { explicit_bzero(v.data(), v.size() * sizeof(uint32_t)); }
    } wipe{expanded};
    for (size_t j = 0; j < np; j++)
        for (size_t i = 0; i < n; i++)
        {
            uint32_t code = (sk_packed[i / 4] >> (6 - 2 * (i % 4))) & 3u;
            if (code > 2)
            {
                set_last_error("secret key holds an invalid 2-bit code (3)");
                return kErrInvalid;
            }
            expanded[j * n + i] = code + (code == 0 ? hp.q[j] : 0u) - 1u;
        }
    DevTemp tmp;
    SEAMD_HIP(tmp.alloc(np * n * sizeof(uint32_t), true));
    uint32_t *d_tmp = tmp.as<uint32_t>();
    if (!d_s_hat) SEAMD_HIP(hipMalloc((void **)&d_s_hat, 2 * np * n * sizeof(uint32_t)));
    SEAMD_HIP(hipMemcpy(d_tmp, expanded.data(), np * n * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (size_t j = 0; j < np; j++)
        SEAMD_HIP(launch_ntt_polys(dp, dt, (int)j, d_tmp + j * n, d_s_hat + 2 * j * n, 1, nullptr));
    SEAMD_HIP(hipDeviceSynchronize());
    dt.s_hat = d_s_hat;
    have_sk  = true;
    return 0;
}

// pk slabs are already NTT-form residues (pk{0,1}_ntt_<n>_<q>.dat, fileops.c:172-204); add the
// Shoup companions once.
int Context::set_public_key(const uint32_t *pk0, const uint32_t *pk1)
{
    const size_t n = hp.n, np = hp.nprimes;
    SEAMD_HIP(hipSetDevice(device));
    for (size_t j = 0; j < np; j++)
        for (size_t i = 0; i < n; i++)
            if (pk0[j * n + i] >= hp.q[j] || pk1[j * n + i] >= hp.q[j])
            {
                set_last_error("public key coefficient not reduced modulo its prime");
                return kErrInvalid;
            }
    std::lock_guard<std::mutex> lk(mu);
    DevTemp tmp;
    SEAMD_HIP(tmp.alloc(2 * np * n * sizeof(uint32_t)));
    uint32_t *d_tmp = tmp.as<uint32_t>();
    if (!d_pk0) SEAMD_HIP(hipMalloc((void **)&d_pk0, 2 * np * n * sizeof(uint32_t)));
    if (!d_pk1) SEAMD_HIP(hipMalloc((void **)&d_pk1, 2 * np * n * sizeof(uint32_t)));
    SEAMD_HIP(hipMemcpy(d_tmp, pk0, np * n * sizeof(uint32_t), hipMemcpyHostToDevice));
    SEAMD_HIP(hipMemcpy(d_tmp + np * n, pk1, np * n * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (size_t j = 0; j < np; j++)
    {
        SEAMD_HIP(launch_make_pairs(d_tmp + j * n, d_pk0 + 2 * j * n, hp.q[j], n, nullptr));
        SEAMD_HIP(launch_make_pairs(d_tmp + np * n + j * n, d_pk1 + 2 * j * n, hp.q[j], n, nullptr));
    }
    SEAMD_HIP(hipDeviceSynchronize());
    dt.pk0  = d_pk0;
    dt.pk1  = d_pk1;
    have_pk = true;
    return 0;
}

// gen_pk (ckks_asym.c:159-171, driven as device/test/ckks_tests_asym.c:174-208 does): the public
// key is a symmetric encryption of zero with a small error:  pk1_j = a_j (shareable PRNG re-seeded
// with pk_seed at counter 0 for EVERY prime), pk0_j = -(a_j . NTT(s)) + NTT(ep mod q_j), ep = n CBD
// samples from PRNG(ep_seed).  Built from the path's own kernels.
int Context::gen_public_key(const uint8_t *sk_packed, const uint8_t *pk_seed, const uint8_t *ep_seed,
                            uint32_t *pk0_out, uint32_t *pk1_out)
{
    // one critical section: the key the public key is derived from is the key that stays installed
    std::lock_guard<std::mutex> lk(mu);
    int rc = set_secret_key_impl(sk_packed);
    if (rc) return rc;
    rc = ensure_scratch(1);
    if (rc) return rc;
    const uint32_t n = (uint32_t)hp.n, np = (uint32_t)hp.nprimes;
    DevTemp seeds, slab;  // slab [2][np][n]: residues/pk0 then a/pk1
    SEAMD_HIP(seeds.alloc(128, true));
    SEAMD_HIP(slab.alloc((size_t)2 * np * n * sizeof(uint32_t)));
    uint8_t *d_seeds = seeds.as<uint8_t>();
    uint32_t *d_c    = slab.as<uint32_t>();
    SEAMD_HIP(hipMemcpy(d_seeds, ep_seed, 64, hipMemcpyHostToDevice));
    SEAMD_HIP(hipMemcpy(d_seeds + 64, pk_seed, 64, hipMemcpyHostToDevice));
    uint32_t *d_p0 = d_c, *d_p1 = d_c + (size_t)np * n;
    CbdArgs ca{d_seeds, nullptr, d_err, n / 16, 1};
    SEAMD_HIP(launch_sample_cbd(ca, nullptr));
    SEAMD_HIP(launch_reduce_small(dp, d_err, d_p0, 1, nullptr));
    EncArgs ea{nullptr, nullptr, nullptr, d_p0, d_p1, nullptr, nullptr, nullptr};
    for (uint32_t j = 0; j < np; j++)
    {
        UniformArgs ua{d_seeds + 64, nullptr, nullptr, d_p1, d_rej, rej_cap, 1, j, j + 1, np, d_spec, spec_cap, 0, 0};
        SEAMD_HIP(launch_sample_uniform(dp, ua, nullptr));
        SEAMD_HIP(launch_ntt_fuse(dp, dt, ea, kModeSym, (int)j, 1, nullptr));
    }
    SEAMD_HIP(hipDeviceSynchronize());
    SEAMD_HIP(hipMemcpy(pk0_out, d_p0, (size_t)np * n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SEAMD_HIP(hipMemcpy(pk1_out, d_p1, (size_t)np * n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SEAMD_HIP(hipMemset(d_err, 0, n));  // the key-generation error is secret too
    return 0;
}

// Batched key generation: K independent key pairs in one launch chain.
//   secret key k : ckks_setup_s's sample branch (ckks_sym.c:162-179) = sample_small_poly_ternary_prng_96
//                  from PRNG(sk_seeds[k]) at counter 0 (sample.c:218-242), 2-bit packed; or sk_in[k]
//   public key k : gen_pk per prime as device/test/ckks_tests_asym.c:174-208 drives it: ep = n CBD samples
//                  from PRNG(ep_seeds[k]); per prime the shareable PRNG restarts from pk_seeds[k] at
//                  counter 0: pk1_j = a_j, pk0_j = -(a_j . NTT(s)) + NTT(ep mod q_j)
// Launches: 1 ternary sampler + 1 pack + 1 CBD + per prime {uniform sampler, sym-prime kernel} for all K.
int Context::gen_keys_batch(size_t K, const uint8_t *sk_in, const uint8_t *sk_seeds, const uint8_t *pk_seeds,
                            const uint8_t *ep_seeds, uint8_t *sk_out, uint32_t *pk0_out, uint32_t *pk1_out)
{
    if (K == 0) return 0;
    if ((!sk_in && !sk_seeds) || !pk_seeds || !ep_seeds || !pk0_out || !pk1_out)
    {
        set_last_error("gen_keys_batch: sk_in or sk_seeds, pk_seeds, ep_seeds and both output slabs are required");
        return kErrInvalid;
    }
    std::lock_guard<std::mutex> lk(mu);
    SEAMD_HIP(hipSetDevice(device));
    int rc = ensure_scratch(K);
    if (rc) return rc;
    const uint32_t n = (uint32_t)hp.n, np = (uint32_t)hp.nprimes;
    const size_t slab = (size_t)K * np * n;
    DevTemp seeds, keys, codes, ep, pk0, pk1, tmp;
    SEAMD_HIP(seeds.alloc(K * 192, true));
    SEAMD_HIP(keys.alloc(K * (n / 4), true));
    SEAMD_HIP(codes.alloc((size_t)K * n, true));
    SEAMD_HIP(ep.alloc((size_t)K * n, true));
    SEAMD_HIP(pk0.alloc(slab * sizeof(uint32_t)));
    SEAMD_HIP(pk1.alloc(slab * sizeof(uint32_t)));
    SEAMD_HIP(tmp.alloc((size_t)K * n * sizeof(uint32_t), true));   // NTT(ep): secret as well
    uint8_t *d_sk_seeds = seeds.as<uint8_t>(), *d_pk_seeds = d_sk_seeds + K * 64, *d_ep_seeds = d_sk_seeds + K * 128;
    if (sk_seeds) SEAMD_HIP(hipMemcpy(d_sk_seeds, sk_seeds, K * 64, hipMemcpyHostToDevice));
    SEAMD_HIP(hipMemcpy(d_pk_seeds, pk_seeds, K * 64, hipMemcpyHostToDevice));
    SEAMD_HIP(hipMemcpy(d_ep_seeds, ep_seeds, K * 64, hipMemcpyHostToDevice));
    if (sk_in)
        SEAMD_HIP(hipMemcpy(keys.p, sk_in, K * (n / 4), hipMemcpyHostToDevice));
    else
    {
        TernaryArgs ta{d_sk_seeds, codes.as<int8_t>(), nullptr, n, (uint32_t)K, nullptr, (uint32_t)num_cus};
        SEAMD_HIP(launch_sample_ternary(ta, nullptr));
        SEAMD_HIP(launch_pack_ternary(codes.as<int8_t>(), keys.as<uint8_t>(), K * (n / 4), nullptr));
    }
    CbdArgs ca{d_ep_seeds, nullptr, ep.as<int8_t>(), n / 16, (uint32_t)K};
    SEAMD_HIP(launch_sample_cbd(ca, nullptr));
    for (uint32_t j = 0; j < np; j++)
    {
        // a_j for every key, counter 0 (gen_pk re-seeds per prime, ckks_asym.c:163), into pk1[:, j]
        UniformArgs ua{d_pk_seeds, nullptr, nullptr, pk1.as<uint32_t>(), d_rej, rej_cap, (uint32_t)K, j, j + 1, np,
                       d_spec,     spec_cap, 0,      debug_flags};
        SEAMD_HIP(launch_sample_uniform(dp, ua, nullptr));
        LowerSymArgs sa{keys.as<uint8_t>(), nullptr, ep.as<int8_t>(), pk1.as<uint32_t>() + (size_t)j * n,
                        pk0.as<uint32_t>() + (size_t)j * n, tmp.as<uint32_t>(), nullptr, (int)j, n / 4, np * n, np * n};
        SEAMD_HIP(launch_lower_sym_prime(dp, dt, sa, K, nullptr));
    }
    SEAMD_HIP(hipDeviceSynchronize());
    if (sk_out) SEAMD_HIP(hipMemcpy(sk_out, keys.p, K * (n / 4), hipMemcpyDeviceToHost));
    SEAMD_HIP(hipMemcpy(pk0_out, pk0.p, slab * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SEAMD_HIP(hipMemcpy(pk1_out, pk1.p, slab * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

void Context::stage_begin(int stage, hipStream_t st)
{
    if (!profiling) return;
    StageEvent ev;
    ev.stage = stage;
    (void)hipEventCreate(&ev.start);
    (void)hipEventCreate(&ev.stop);
    (void)hipEventRecord(ev.start, st);
    events.push_back(ev);
}

void Context::stage_end(hipStream_t st)
{
    if (!profiling || events.empty()) return;
    (void)hipEventRecord(events.back().stop, st);
}

void Context::collect_events()
{
    for (auto &ev : events)
    {
        float ms = 0;
        if (hipEventSynchronize(ev.stop) == hipSuccess &&
            hipEventElapsedTime(&ms, ev.start, ev.stop) == hipSuccess)
        {
            stage_ms[ev.stage] += ms;
            stage_launches[ev.stage]++;
        }
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    events.clear();
}

int Context::encrypt_sym(const float *d_values, size_t B, const uint8_t *d_share_seeds,
                         const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1,
                         uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(mu);
    int rc = begin_call(st);
    if (rc) return rc;
    rc = encrypt_sym_impl(d_values, B, d_share_seeds, d_seeds, d_c0, d_c1, d_ntt_pte, d_pte, d_status, st);
    return end_call(st, rc);
}

// Seed-compressed form: `a` goes to context scratch instead of a caller slab.  The scratch is grown, and
// its pointer handed to the kernels, inside ONE critical section -- a concurrent call with a larger batch
// cannot free it under a call that is still being launched (and the launched work is ordered behind
// ev_done like every other use of the context's scratch).
int Context::encrypt_sym_seeded(const float *d_values, size_t B, const uint8_t *d_share_seeds,
                                const uint8_t *d_seeds, uint32_t *d_c0, uint8_t *d_status, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(mu);
    SEAMD_HIP(hipSetDevice(device));
    if (B > a_cap)
    {
        SEAMD_HIP(hipDeviceSynchronize());   // earlier calls may still read the old slab
        wipe_device(d_a, a_cap * hp.nprimes * hp.n * sizeof(uint32_t));
        if (d_a) (void)hipFree(d_a);
        d_a   = nullptr;
        a_cap = 0;
        SEAMD_HIP(hipMalloc((void **)&d_a, B * hp.nprimes * hp.n * sizeof(uint32_t)));
        a_cap = B;
    }
    int rc = begin_call(st);
    if (rc) return rc;
    rc = encrypt_sym_impl(d_values, B, d_share_seeds, d_seeds, d_c0, d_a, nullptr, nullptr, d_status, st);
    return end_call(st, rc);
}

int Context::encrypt_asym(const float *d_values, size_t B, const uint8_t *d_seeds, uint32_t *d_c0,
                          uint32_t *d_c1, uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status,
                          hipStream_t st)
{
    std::lock_guard<std::mutex> lk(mu);
    int rc = begin_call(st);
    if (rc) return rc;
    rc = encrypt_asym_impl(d_values, B, d_seeds, d_c0, d_c1, d_ntt_pte, d_pte, d_status, st);
    return end_call(st, rc);
}

int Context::sample_uniform(const uint8_t *d_seeds, const uint64_t *d_ctr_in, size_t B, uint32_t *d_out,
                            uint64_t *d_ctr_out, hipStream_t st)
{
    if (B == 0) return 0;
    std::lock_guard<std::mutex> lk(mu);
    int rc = begin_call(st);
    if (rc) return rc;
    rc = ensure_scratch(B);
    if (rc == 0)
    {
        const uint32_t np = (uint32_t)hp.nprimes;
        UniformArgs ua{d_seeds, d_ctr_in, d_ctr_out, d_out, d_rej, rej_cap, (uint32_t)B, 0, np, np,
                       d_spec,  spec_cap, 0,         debug_flags};
        hipError_t e = launch_sample_uniform(dp, ua, st);
        if (e != hipSuccess) rc = hip_fail(e, "launch_sample_uniform");
    }
    return end_call(st, rc);
}

int Context::encrypt_sym_impl(const float *d_values, size_t B, const uint8_t *d_share_seeds,
                              const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1,
                              uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status, hipStream_t st)
{
    if (!have_sk)
    {
        set_last_error("symmetric encryption needs a secret key (se_amd_set_secret_key)");
        return kErrNoKey;
    }
    if (B == 0) return 0;
    if (!d_values || !d_share_seeds || !d_seeds || !d_c0 || !d_c1) return kErrInvalid;
    SEAMD_HIP(hipSetDevice(device));
    {
        // a handful of ciphertexts: latency path (all primes' samplers at once, prime speculation)
        SpecPlan plan;
        if (split_mode == 2 && small_batch_plan(B, plan) && speculation_pays(B, plan))
            return encrypt_sym_small(plan, d_values, d_share_seeds, d_seeds, d_c0, d_c1, d_ntt_pte, d_pte,
                                     d_status, st);
    }
    int rc = ensure_scratch(B);
    if (rc) return rc;
    const uint32_t n = (uint32_t)hp.n, np = (uint32_t)hp.nprimes;

    CbdArgs ca{d_seeds, nullptr, d_err, n / 16, (uint32_t)B};
    EncArgs ea{d_values, d_err, nullptr, d_c0, d_c1, d_ntt_pte, d_pte, d_status, d_general, d_compact};

    const size_t chain_waves_per_cu = ((B + 63) / 64 + (size_t)num_cus - 1) / (size_t)num_cus;
    const bool split = split_mode == 1 || (split_mode == 2 && (hp.n >= 8192 || chain_waves_per_cu < 4));
    if (!split)
    {
        // Simple chain: [cbd on the aux stream || uniform] -> fused encode+encrypt.
        hipStream_t cbd_stream = overlap ? aux_stream : st;
        if (overlap)
        {
            SEAMD_HIP(hipEventRecord(ev_fork, st));
            SEAMD_HIP(hipStreamWaitEvent(aux_stream, ev_fork, 0));
        }
        stage_begin(0, cbd_stream);
        SEAMD_HIP(launch_sample_cbd(ca, cbd_stream));
        stage_end(cbd_stream);
        if (overlap) SEAMD_HIP(hipEventRecord(ev_join, aux_stream));
        UniformArgs ua{d_share_seeds, nullptr, nullptr, d_c1, d_rej, rej_cap, (uint32_t)B, 0, np, np,
                       d_spec, spec_cap, 0, debug_flags};
        stage_begin(1, st);
        SEAMD_HIP(launch_sample_uniform(dp, ua, st));
        stage_end(st);
        if (overlap) SEAMD_HIP(hipStreamWaitEvent(st, ev_join, 0));
        stage_begin(3, st);
        SEAMD_HIP(launch_encode_encrypt(dp, dt, ea, kModeSym, B, st));
        stage_end(st);
        return 0;
    }

    // Software pipeline over the primes.  The uniform sampler is one long sequential chain per
    // ciphertext that keeps ONE wave per SIMD busy; everything that does not need a_j runs beside
    // it on the auxiliary stream:
    //   S : U_0 ──► U_1 ──► ... ──► U_{np-1} ─────────────► N_{np-1}
    //   A : cbd ► encode_rns ► (wait U_0) N_0 ► (wait U_1) N_1 ► ... ┘(join)
    // U_j = k_sample_uniform for prime j (counter carried in d_ctr), N_j = k_ntt_fuse for prime j.
    hipStream_t ax = overlap ? aux_stream : st;
    if (overlap)
    {
        SEAMD_HIP(hipEventRecord(ev_fork, st));
        SEAMD_HIP(hipStreamWaitEvent(ax, ev_fork, 0));
    }
    // n = 16384 (`late_encode`): the encode workgroup (136 KiB of LDS, 128 VGPRs x 16 waves) cannot
    // share a CU with a chain workgroup, so on the auxiliary stream it would sit behind every U_j
    // and drag all N_j with it (measured: no overlap at all).  There it runs on the main stream
    // between U_0 and U_1, the chain workgroups carry 2 helper waves instead of 6 (one wave per SIMD,
    // 128 VGPRs) and k_ntt_fuse<14> is held to 80 VGPRs, so an N_j workgroup fits beside U_{j+1}:
    //   S : U_0 ► (wait cbd) encode_rns ► U_1 ──► U_2 ──► ... ► U_{np-1} ──────► N_{np-1}
    //   A : cbd ─────────────────────────► N_0 ► (wait U_1) N_1 ► ...        ┘(join)
    const bool late_encode   = overlap && hp.n >= 16384 && !(debug_flags & 128);
    const uint32_t fill      = late_encode ? 4 : 0;  // leave VGPRs for the co-resident N_j workgroup
    stage_begin(0, ax);
    SEAMD_HIP(launch_sample_cbd(ca, ax));  // e, counters 0.. (ckks_sym.c:196)
    stage_end(ax);
    if (late_encode)
        SEAMD_HIP(hipEventRecord(ev_cbd, ax));
    else
    {
        stage_begin(4, ax);
        SEAMD_HIP(launch_encode_rns(dp, dt, ea, true, B, ax));
        stage_end(ax);
    }
    // Staged sampler (kernels/samplers.hip: k_bulk_pair / k_candidates / k_resolve_wave) for batches between
    // the wave form's limit and one PAIR wave per SIMD (B <= 128 per CU): the chain launches are the critical
    // path of this pipeline, and a ciphertext per lane pair shortens each by ~1.5x; the candidates, which the
    // helper waves of k_sample_uniform compute inside the chain workgroups, become a throughput kernel on a
    // stream of its own.  debug_flags 512 forces it, 1024 forbids it.
    const bool staged = overlap && !(debug_flags & (32 | 1024)) && staged_mode != 0 &&
                        ((B > uniform_wave_limit((unsigned)num_cus) && B <= (size_t)128 * (size_t)num_cus) ||
                         (debug_flags & 512) || staged_mode == 1);
    if (staged && !cand_stream)
    {
        SEAMD_HIP(hipStreamCreateWithFlags(&cand_stream, hipStreamNonBlocking));
        for (auto &e : ev_cand) SEAMD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    for (uint32_t j = 0; j < np; j++)
    {
        // a_j from the shareable seed, written straight into c1 (ckks_sym.c:220)
        UniformArgs ua{d_share_seeds, j ? d_ctr : nullptr, d_ctr, d_c1, d_rej, rej_cap, (uint32_t)B,
                       j,             j + 1,               np,    d_spec,      spec_cap,
                       0,             debug_flags,         nullptr, 0, fill, nullptr, d_nrej,
                       staged ? d_flagged : nullptr};
        if (staged)
        {
            // Candidates per ciphertext: here they are pure throughput work beside the chains, and a
            // ciphertext that needs more gets them from the resolving wave itself -- so mean + 1.5 sigma of the
            // draws instead of the helper waves' mean + 4 sigma (every unused candidate is a wasted permutation:
            // 384 -> 336 at n = 16384 is 4 % of the phase's instructions).
            {
                const double p_rej = (double)(0u - dp.bound[j]) / 4294967296.0;
                const double mean  = (double)hp.n * p_rej / (1.0 - p_rej);
                const uint32_t cap = ((uint32_t)(mean + 1.5 * sqrt((double)hp.n * p_rej) + 15.0)) & ~15u;
                if (cap < ua.spec_cap) ua.spec_cap = cap ? cap : 16u;
            }
            //   C : (start counters of prime j known) k_candidates_j ───────────┐
            //   S : k_bulk_pair_j ─────────────────────────────────── (wait C) k_resolve_wave_j
            SEAMD_HIP(hipStreamWaitEvent(cand_stream, j ? ev_prime[j - 1] : ev_fork, 0));
            SEAMD_HIP(launch_uniform_candidates(ua, cand_stream));
            SEAMD_HIP(hipEventRecord(ev_cand[j], cand_stream));
            stage_begin(1, st);
            SEAMD_HIP(launch_uniform_bulk_pair(dp, ua, st));
            SEAMD_HIP(hipStreamWaitEvent(st, ev_cand[j], 0));
            SEAMD_HIP(launch_uniform_resolve(dp, ua, st));
            stage_end(st);
        }
        else
        {
            stage_begin(1, st);
            SEAMD_HIP(launch_sample_uniform(dp, ua, st));
            stage_end(st);
        }
        if (late_encode && j == 0)
        {
            SEAMD_HIP(hipStreamWaitEvent(st, ev_cbd, 0));
            stage_begin(4, st);
            SEAMD_HIP(launch_encode_rns(dp, dt, ea, true, B, st));
            stage_end(st);
        }
        if (j + 1 < np)
        {
            if (overlap)
            {
                SEAMD_HIP(hipEventRecord(ev_prime[j], st));
                SEAMD_HIP(hipStreamWaitEvent(ax, ev_prime[j], 0));
            }
            stage_begin(5, ax);
            SEAMD_HIP(launch_ntt_fuse(dp, dt, ea, kModeSym, (int)j, B, ax));
            stage_end(ax);
        }
    }
    if (overlap)
    {
        SEAMD_HIP(hipEventRecord(ev_join, ax));
        SEAMD_HIP(hipStreamWaitEvent(st, ev_join, 0));
    }
    stage_begin(5, st);
    SEAMD_HIP(launch_ntt_fuse(dp, dt, ea, kModeSym, (int)np - 1, B, st));
    stage_end(st);
    return 0;
}

// ---- small-batch prime speculation -----------------------------------------------------------
bool Context::small_batch_plan(size_t B, SpecPlan &plan) const
{
    const uint32_t np = (uint32_t)hp.nprimes;
    if (!overlap || !have_sk || np < 2 || B == 0 || B > 1024) return false;
    plan         = SpecPlan{};
    plan.nprimes = np;
    plan.B       = (uint32_t)B;
    const double n     = (double)hp.n;
    const double width = (debug_flags & 256) ? 0.0 : 5.5;  // 256: windows of one guess (forces the fallback)
    double mu = 0.0, var = 0.0;
    uint64_t off = 0;
    for (uint32_t j = 1; j < np; j++)
    {
        // prime j-1 consumed 1 block + (draws) counters; draws ~ Binomial(n, p)/(1 - p)
        const double p = (double)(0u - dp.bound[j - 1]) / 4294967296.0;
        mu += 1.0 + n * p / (1.0 - p);
        var += n * p * 1.06;
        const uint64_t h = (debug_flags & 256) ? 0 : (uint64_t)(width * sqrt(var)) + 2;
        const uint64_t c = (uint64_t)(mu + 0.5);
        plan.base[j]     = c > h ? c - h : 0;
        plan.count[j]    = (uint32_t)(2 * h + 1);
        plan.offset[j]   = (uint32_t)off;
        off += (uint64_t)B * plan.count[j];
        if (off > (uint64_t)small_limit) return false;  // beyond this it stops paying
        if (off * hp.n * sizeof(uint32_t) > small_bytes) return false;  // scratch rows bounded in bytes
    }
    plan.total = (uint32_t)off;
    // All guesses are ONE launch (UniformArgs::prime_of).  Beyond the wave form's limit that launch takes the
    // lane form, whose per-lane-prime instantiation exists for workgroups of up to 8 waves = 512 ciphertexts per
    // CU: a fan-out wider than that (small devices / partitions: fewer than 128 CUs at the default small_limit)
    // is not planned at all and the call takes the ordinary per-prime chain.
    if ((uint64_t)B + off > (uint64_t)512 * (uint64_t)num_cus) return false;
    return true;
}

// Which form serves a small batch faster?  The uniform sampler is a chain of `steps` sequential permutations
// per polynomial; what it costs depends on the kernel form the launch takes (kernels/samplers.hip,
// launch_sample_uniform; per-permutation times measured with tools/ubench5 and tools/chain_step_probe.py):
//   wave per ciphertext : 3.3 us up to 256 waves, + 1.2 us per further 1 024 waves (4.2 us at one wave per
//                         SIMD, 7.8 us at four)
//   lane per ciphertext : 10.7 us whatever the batch (up to one wave per SIMD)
// Speculation runs the guesses of every prime at once: one chain deep, but B + plan.total ciphertexts wide;
// the plain form runs np chains in sequence, B wide.
bool Context::speculation_pays(size_t B, const SpecPlan &plan) const
{
    if (spec_mode >= 0) return spec_mode != 0;          // SE_AMD_SPECULATION
    const double steps = (double)((hp.n * 4 + 135) / 136);
    // measured on 256 CUs: flat up to one wave per CU, then + 1.2 us per further wave per SIMD (4 per CU)
    const double cus = (double)num_cus;
    auto perm_us = [&](size_t cts) {
        if (cts > uniform_wave_limit((unsigned)num_cus) || (debug_flags & 32)) return 10.7;
        return 3.3 + 1.2 * ((double)cts > cus ? ((double)cts - cus) / (4.0 * cus) : 0.0);
    };
    const double spec  = steps * perm_us(B + plan.total);
    const double plain = (double)hp.nprimes * steps * perm_us(B);
    return spec < 0.9 * plain;
}

int Context::encrypt_sym_small(const SpecPlan &plan, const float *d_values, const uint8_t *d_share_seeds,
                               const uint8_t *d_seeds, uint32_t *d_c0, uint32_t *d_c1,
                               uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status, hipStream_t st)
{
    if (!d_values || !d_share_seeds || !d_seeds || !d_c0 || !d_c1) return kErrInvalid;
    SEAMD_HIP(hipSetDevice(device));
    const size_t B     = plan.B;
    const uint32_t n   = (uint32_t)hp.n, np = (uint32_t)hp.nprimes;
    const size_t total = plan.total;
    int rc             = ensure_scratch(B, B + total);  // reject lists / candidates of the virtual ciphertexts too
    if (rc) return rc;
    if (total > sp_cap)
    {
        SEAMD_HIP(hipDeviceSynchronize());
        wipe_device(d_sp_seeds, sp_cap * 64);
        void *old[] = {d_sp_seeds, d_sp_ctr, d_sp_ctrout, d_sp_rows, d_sp_prime};
        for (void *p : old)
            if (p) (void)hipFree(p);
        d_sp_seeds = nullptr, d_sp_ctr = nullptr, d_sp_ctrout = nullptr, d_sp_rows = nullptr, d_sp_prime = nullptr;
        sp_cap = 0;
        SEAMD_HIP(hipMalloc((void **)&d_sp_seeds, total * 64));
        SEAMD_HIP(hipMalloc((void **)&d_sp_ctr, total * sizeof(uint64_t)));
        SEAMD_HIP(hipMalloc((void **)&d_sp_ctrout, total * sizeof(uint64_t)));
        SEAMD_HIP(hipMalloc((void **)&d_sp_rows, total * (size_t)n * sizeof(uint32_t)));
        SEAMD_HIP(hipMalloc((void **)&d_sp_prime, total));
        sp_cap = total;
    }
    if (B > sp_fail_cap)
    {
        SEAMD_HIP(hipDeviceSynchronize());
        if (d_sp_fail) (void)hipFree(d_sp_fail);
        d_sp_fail = nullptr, sp_fail_cap = 0;
        SEAMD_HIP(hipMalloc((void **)&d_sp_fail, 1024 * sizeof(uint32_t)));
        sp_fail_cap = 1024;
    }
    // Three streams whatever the length of the prime chain (the runtime's default of 4 hardware queues is
    // enough): the guesses of ALL primes are ONE launch (UniformArgs::prime_of).
    if (!spec_stream) SEAMD_HIP(hipStreamCreateWithFlags(&spec_stream, hipStreamNonBlocking));

    CbdArgs ca{d_seeds, nullptr, d_err, n / 16, (uint32_t)B};
    EncArgs ea{d_values, d_err, nullptr, d_c0, d_c1, d_ntt_pte, d_pte, d_status, d_general, d_compact};

    //   S : U_0 (real ciphertexts) ─────────────────────┐ (wait P) select ► redo (masked) ► (wait A) N_0 .. N_{np-1}
    //   A : cbd ► k_encode_rns ─────────────────────────┤
    //   P : setup ► U_{1..np-1} (all guesses, one launch)┘
    SEAMD_HIP(hipEventRecord(ev_fork, st));
    SEAMD_HIP(hipStreamWaitEvent(aux_stream, ev_fork, 0));
    SEAMD_HIP(hipStreamWaitEvent(spec_stream, ev_fork, 0));

    SEAMD_HIP(launch_spec_setup(plan, d_share_seeds, d_sp_seeds, d_sp_ctr, d_sp_prime, spec_stream));
    {
        // one output row per virtual ciphertext; its prime comes from d_sp_prime
        UniformArgs ug{d_sp_seeds, d_sp_ctr, d_sp_ctrout, d_sp_rows, d_rej + B * rej_cap, rej_cap, (uint32_t)total,
                       0,          0,        1,           d_spec + B * spec_cap, spec_cap, 0, debug_flags,
                       nullptr,    0,        0,           d_sp_prime};
        SEAMD_HIP(launch_sample_uniform(dp, ug, spec_stream));
        SEAMD_HIP(hipEventRecord(ev_enc, spec_stream));
    }

    stage_begin(0, aux_stream);
    SEAMD_HIP(launch_sample_cbd(ca, aux_stream));
    stage_end(aux_stream);
    stage_begin(4, aux_stream);
    SEAMD_HIP(launch_encode_rns(dp, dt, ea, true, B, aux_stream));
    stage_end(aux_stream);
    SEAMD_HIP(hipEventRecord(ev_join, aux_stream));

    UniformArgs u0{d_share_seeds, nullptr, d_ctr, d_c1, d_rej, rej_cap, (uint32_t)B, 0, 1, np,
                   d_spec,        spec_cap, 0,     debug_flags};
    stage_begin(1, st);
    SEAMD_HIP(launch_sample_uniform(dp, u0, st));
    stage_end(st);
    SEAMD_HIP(hipStreamWaitEvent(st, ev_enc, 0));
    SEAMD_HIP(launch_spec_select(plan, n, d_ctr, d_sp_ctrout, d_sp_rows, d_c1, d_sp_fail, st));
    // misses (~1e-7 per prime): the ordinary per-prime chain, masked to the ciphertexts that missed;
    // without a miss every workgroup of these launches returns at once
    for (uint32_t j = 1; j < np; j++)
    {
        UniformArgs ur{d_share_seeds, d_ctr,    d_ctr, d_c1,        d_rej,     rej_cap, (uint32_t)B, j, j + 1, np,
                       d_spec,        spec_cap, 0,     debug_flags, d_sp_fail, 0,       0};
        SEAMD_HIP(launch_sample_uniform(dp, ur, st));
    }
    SEAMD_HIP(hipStreamWaitEvent(st, ev_join, 0));
    for (uint32_t j = 0; j < np; j++)
    {
        stage_begin(5, st);
        SEAMD_HIP(launch_ntt_fuse(dp, dt, ea, kModeSym, (int)j, B, st));
        stage_end(st);
    }
    return 0;
}

int Context::encrypt_asym_impl(const float *d_values, size_t B, const uint8_t *d_seeds, uint32_t *d_c0,
                               uint32_t *d_c1, uint32_t *d_ntt_pte, int64_t *d_pte, uint8_t *d_status,
                               hipStream_t st)
{
    if (!have_pk)
    {
        set_last_error("asymmetric encryption needs a public key (se_amd_set_public_key)");
        return kErrNoKey;
    }
    if (B == 0) return 0;
    if (!d_values || !d_seeds || !d_c0 || !d_c1) return kErrInvalid;
    SEAMD_HIP(hipSetDevice(device));
    int rc = ensure_scratch(B);
    if (rc) return rc;
    const uint32_t n = (uint32_t)hp.n;

    // u first for the whole batch (its redraws decide where each ciphertext's CBD counters start,
    // ckks_asym.c:188-201): the ternary sampler is a per-ciphertext chain, so its duration does not
    // shrink with a chunk and it is launched once.  Then the batch is cut into chunks: the CBD sampler
    // of chunk k+1 (e0 | e1 contiguous per ciphertext; VALU-throughput-bound) runs on the auxiliary
    // stream beside the fused kernel of chunk k (latency-bound at 2 waves per SIMD), which leaves the
    // CBD sampler's time mostly hidden.
    //   S : ternary(all) ──fork──────► (wait C_0) E_0 ► (wait C_1) E_1 ► ...
    //   A :                 └► C_0 ► C_1 ► C_2 ► ...
    const size_t np = hp.nprimes;
    size_t nchunks  = overlap ? asym_chunks : 1;
    if (nchunks > (size_t)kMaxPrimes) nchunks = kMaxPrimes;   // one join event per chunk
    if (nchunks < 1 || B < 4096 * nchunks) nchunks = 1;        // small batches: one chunk
    TernaryArgs ta{d_seeds, d_ucodes, d_ctr, n, (uint32_t)B, nullptr, (uint32_t)num_cus, debug_flags};
    stage_begin(2, st);
    SEAMD_HIP(launch_sample_ternary(ta, st));
    stage_end(st);
    hipStream_t ax = nchunks > 1 ? aux_stream : st;
    if (nchunks > 1)
    {
        SEAMD_HIP(hipEventRecord(ev_fork, st));
        SEAMD_HIP(hipStreamWaitEvent(ax, ev_fork, 0));
    }
    for (size_t c = 0; c < nchunks; c++)
    {
        const size_t lo = B * c / nchunks, hi = B * (c + 1) / nchunks, cb = hi - lo;
        if (cb == 0) continue;
        CbdArgs ca{d_seeds + lo * 64, d_ctr + lo, d_err + lo * 2 * n, 2 * (n / 16), (uint32_t)cb};
        stage_begin(0, ax);
        SEAMD_HIP(launch_sample_cbd(ca, ax));
        stage_end(ax);
        if (nchunks > 1) SEAMD_HIP(hipEventRecord(ev_prime[c], ax));
    }
    for (size_t c = 0; c < nchunks; c++)
    {
        const size_t lo = B * c / nchunks, hi = B * (c + 1) / nchunks, cb = hi - lo;
        if (cb == 0) continue;
        if (nchunks > 1) SEAMD_HIP(hipStreamWaitEvent(st, ev_prime[c], 0));
        EncArgs ea{d_values + lo * (n / 2),
                   d_err + lo * 2 * n,
                   d_ucodes + lo * n,
                   d_c0 + lo * np * n,
                   d_c1 + lo * np * n,
                   d_ntt_pte ? d_ntt_pte + lo * np * n : nullptr,
                   d_pte ? d_pte + lo * n : nullptr,
                   d_status ? d_status + lo : nullptr,
                   d_general};
        stage_begin(3, st);
        SEAMD_HIP(launch_encode_encrypt(dp, dt, ea, kModeAsym, cb, st));
        stage_end(st);
    }
    return 0;
}

// d_out NULL: plain ckks_encode_base (only the int64 plaintexts are written)
int Context::encode_ntt(const float *d_values, size_t B, uint32_t *d_out, int64_t *d_pte,
                        uint8_t *d_status, hipStream_t st)
{
    if (B == 0) return 0;
    if (!d_values || (!d_out && !d_pte)) return kErrInvalid;
    std::lock_guard<std::mutex> lk(mu);   // the list of declined plaintexts is context scratch
    int rc = begin_call(st);
    if (rc) return rc;
    rc = ensure_general(B);
    if (rc == 0)
    {
        EncArgs ea{d_values, nullptr, nullptr, d_out, nullptr, nullptr, d_pte, d_status, d_general};
        stage_begin(3, st);
        hipError_t e = launch_encode_encrypt(dp, dt, ea, kModeEncodeOnly, B, st);
        if (e != hipSuccess) rc = hip_fail(e, "launch_encode_encrypt");
        stage_end(st);
    }
    return end_call(st, rc);
}

}  // namespace seamd
