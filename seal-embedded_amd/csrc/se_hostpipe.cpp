// se_hostpipe.cpp -- see se_hostpipe.h.
#include "se_hostpipe.h"

#include <sched.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>

#include "se_context.h"

namespace seamd {

// ---- memcpy pool ----------------------------------------------------------------------------
CopyPool::CopyPool(int nthreads) : nthreads_(std::max(1, nthreads))
{
    for (int i = 1; i < nthreads_; i++) workers_.emplace_back(&CopyPool::worker, this, i);
}

CopyPool::~CopyPool()
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
        generation_++;
    }
    cv_job_.notify_all();
    for (auto &t : workers_) t.join();
}

static void copy_part(char *dst, const char *src, size_t bytes, int id, int nparts)
{
    // 4 KiB-aligned split so no two threads share a page of the destination
    size_t per = ((bytes + nparts - 1) / nparts + 4095) & ~size_t(4095);
    size_t lo  = std::min(bytes, per * id), hi = std::min(bytes, per * (id + 1));
    if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
}

void CopyPool::worker(int id)
{
    uint64_t seen = 0;
    for (;;)
    {
        char *dst;
        const char *src;
        size_t bytes;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_job_.wait(lk, [&] { return generation_ != seen; });
            seen = generation_;
            if (stop_) return;
            dst = dst_, src = src_, bytes = bytes_;
        }
        copy_part(dst, src, bytes, id, nthreads_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
}

void CopyPool::copy(void *dst, const void *src, size_t bytes)
{
    if (nthreads_ == 1 || bytes < (size_t(1) << 20))
    {
        memcpy(dst, src, bytes);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        dst_ = (char *)dst, src_ = (const char *)src, bytes_ = bytes;
        pending_ = nthreads_ - 1;
        generation_++;
    }
    cv_job_.notify_all();
    copy_part((char *)dst, (const char *)src, bytes, 0, nthreads_);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
}

static int default_copy_threads()
{
    if (const char *e = getenv("SE_AMD_HOST_THREADS"))
    {
        int v = atoi(e);
        if (v >= 1) return std::min(v, 64);
    }
    cpu_set_t set;
    int ncpu = 1;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) ncpu = CPU_COUNT(&set);
    return std::max(1, std::min(8, ncpu / 2));
}

// ---- pipeline -------------------------------------------------------------------------------
static bool is_pinned(const void *p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess)
    {
        (void)hipGetLastError();  // unregistered host memory on older runtimes: not an error here
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// the device slots hold PRNG seeds, plaintext values and m + e between calls: zeroed before they go back to the
// allocator (callers have drained the streams that used them)
static int regrow(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return 0;
    if (*p)
    {
        (void)hipMemset(*p, 0, *cap);
        (void)hipFree(*p);
    }
    *p   = nullptr;
    *cap = 0;
    SEAMD_HIP(hipMalloc(p, bytes));
    *cap = bytes;
    return 0;
}

// zero the secret-bearing buffers of a slot (s.cap ciphertexts: seeds of e / u, the shareable seed, plaintext values)
void HostPipe::wipe_slot(Slot &s)
{
    if (s.seeds) (void)hipMemset(s.seeds, 0, s.cap * 64);
    if (s.share_seeds) (void)hipMemset(s.share_seeds, 0, s.cap * 64);
    if (s.values) (void)hipMemset(s.values, 0, s.cap * values_bytes_per_ct);
}

HostPipe::~HostPipe()
{
    (void)hipSetDevice(device);
    if (compute) (void)hipStreamSynchronize(compute);
    if (copy) (void)hipStreamSynchronize(copy);
    for (auto &s : slot)
    {
        // secret-bearing slots (seeds of e / u, the shareable seed, values, m + e) are wiped first
        wipe_slot(s);
        if (s.pte) (void)hipMemset(s.pte, 0, s.cap_pte);
        if (s.ntt_pte) (void)hipMemset(s.ntt_pte, 0, s.cap_ntt);   // NTT(m + e) is as sensitive as m + e
        (void)hipDeviceSynchronize();
        void *ptrs[] = {s.values, s.seeds, s.share_seeds, s.c0, s.c1, s.ntt_pte, s.pte};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        if (s.computed) (void)hipEventDestroy(s.computed);
        if (s.copied) (void)hipEventDestroy(s.copied);
    }
    for (int r = 0; r < kRing; r++)
    {
        if (ring[r])
        {
            explicit_bzero(ring[r], ring_bytes);   // the pinned staging ring carried m + e / ciphertext pieces
            (void)hipHostFree(ring[r]);
        }
        if (ring_ev[r]) (void)hipEventDestroy(ring_ev[r]);
    }
    if (d_status) (void)hipFree(d_status);
    if (h_status) (void)hipHostFree(h_status);
    if (compute) (void)hipStreamDestroy(compute);
    if (copy) (void)hipStreamDestroy(copy);
    delete pool;
}

int HostPipe::init(int dev)
{
    device = dev;
    SEAMD_HIP(hipSetDevice(device));
    SEAMD_HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
    SEAMD_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    for (auto &s : slot)
    {
        SEAMD_HIP(hipEventCreateWithFlags(&s.computed, hipEventDisableTiming));
        SEAMD_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    }
    for (int r = 0; r < kRing; r++)
        SEAMD_HIP(hipEventCreateWithFlags(&ring_ev[r], hipEventDisableTiming));
    pool = new CopyPool(default_copy_threads());
    return 0;
}

int HostPipe::ensure(Context &c, size_t chunk, size_t B, bool want_ntt, bool want_pte, bool staged)
{
    const size_t n = c.hp.n, np = c.hp.nprimes;
    const size_t ct_bytes = np * n * 4;
    for (auto &s : slot)
    {
        if (chunk > s.cap)
        {
            // wipe_slot's memsets run on the null stream, which does not order against the NON-BLOCKING compute / copy
            // streams: drain them first instead of relying on the callers having done so
            if (compute) SEAMD_HIP(hipStreamSynchronize(compute));
            if (copy) SEAMD_HIP(hipStreamSynchronize(copy));
            wipe_slot(s);
            values_bytes_per_ct = (n / 2) * sizeof(float);
            void **ptrs[] = {&s.values, &s.seeds, &s.share_seeds, &s.c0, &s.c1};
            for (void **p : ptrs)
                if (*p)
                {
                    (void)hipFree(*p);
                    *p = nullptr;
                }
            s.cap = 0;
            SEAMD_HIP(hipMalloc(&s.values, chunk * (n / 2) * sizeof(float)));
            SEAMD_HIP(hipMalloc(&s.seeds, chunk * 64));
            SEAMD_HIP(hipMalloc(&s.share_seeds, chunk * 64));
            SEAMD_HIP(hipMalloc(&s.c0, chunk * ct_bytes));
            SEAMD_HIP(hipMalloc(&s.c1, chunk * ct_bytes));
            s.cap = chunk;
        }
        int rc;
        if (want_ntt && (rc = regrow(&s.ntt_pte, &s.cap_ntt, chunk * ct_bytes))) return rc;
        if (want_pte && (rc = regrow(&s.pte, &s.cap_pte, chunk * n * 8))) return rc;
    }
    int rc;
    if ((rc = regrow(&d_status, &status_cap, B))) return rc;
    if (B > h_status_cap)
    {
        if (h_status) (void)hipHostFree(h_status);
        h_status = nullptr, h_status_cap = 0;
        const size_t want = (std::max<size_t>(B, 4096) + 4095) & ~size_t(4095);
        SEAMD_HIP(hipHostMalloc((void **)&h_status, want, hipHostMallocDefault));
        h_status_cap = want;
    }
    if (staged)
    {
        // ring entries sized to the work: a single small ciphertext must not pin 256 MiB
        size_t want = std::min(kPieceMax, std::max(chunk * ct_bytes, chunk * n * 8));
        want        = (want + 4095) & ~size_t(4095);
        if (want > ring_bytes)
        {
            for (int r = 0; r < kRing; r++)
            {
                if (ring[r])
                {
                    explicit_bzero(ring[r], ring_bytes);
                    (void)hipHostFree(ring[r]);
                }
                ring[r] = nullptr;
            }
            ring_bytes = 0;
            for (int r = 0; r < kRing; r++) SEAMD_HIP(hipHostMalloc(&ring[r], want, hipHostMallocDefault));
            ring_bytes = want;
        }
    }
    return 0;
}

namespace {
struct Piece
{
    const char *src;
    char *dst;
    size_t bytes;
    size_t chunk;
    bool first, last, direct;
};
struct InFlight
{
    int r;
    char *dst;
    size_t bytes;
};
}  // namespace

int HostPipe::run(Context &c, bool asym, const float *values, size_t B, const uint8_t *share_seeds,
                  const uint8_t *seeds, uint32_t *c0, uint32_t *c1, uint32_t *ntt_pte, int64_t *pte,
                  uint8_t *status)
{
    const size_t n = c.hp.n, np = c.hp.nprimes;
    const size_t ct_bytes = np * n * 4;
    SEAMD_HIP(hipSetDevice(device));

    // chunk: enough ciphertexts to fill the chip's chain kernels (16384 = 256 CUs x 64 lanes) but no
    // more than 4 GiB per output slab; chunks equalised so there is no tiny tail
    size_t chunk_max = chunk_override
                           ? chunk_override
                           : std::min<size_t>(16384, std::max<size_t>(256, (size_t(4) << 30) / (2 * ct_bytes)));
    const size_t nchunks = (B + chunk_max - 1) / chunk_max;
    size_t chunk         = (B + nchunks - 1) / nchunks;
    if (!chunk_override) chunk = std::min(B, (chunk + 63) & ~size_t(63));
    const size_t nch = (B + chunk - 1) / chunk;

    // c1 == nullptr (symmetric only): seed-compressed form, `a` stays in the device slot
    const bool pin0 = is_pinned(c0), pin1 = !c1 || is_pinned(c1);
    const bool pinn = ntt_pte && is_pinned(ntt_pte), pinp = pte && is_pinned(pte);
    const bool staged = !pin0 || !pin1 || (ntt_pte && !pinn) || (pte && !pinp);
    int rc = ensure(c, chunk, B, ntt_pte != nullptr, pte != nullptr, staged);
    if (rc) return rc;

    auto chunk_count = [&](size_t k) { return std::min(chunk, B - k * chunk); };

    auto launch_chunk = [&](size_t k) -> int {
        Slot &s          = slot[k % kSlots];
        const size_t lo  = k * chunk, cnt = chunk_count(k);
        // inputs go through the runtime's pageable H2D path (staging them through own pinned buffers
        // with the memcpy pool measured 9 % slower end to end).  They are issued first: they only
        // depend on the previous use of this slot's input buffers (same
        // stream); the wait for the slot's outputs to be copied out comes after, so the (host
        // blocking) pageable H2D never waits on ring pieces this thread has yet to drain
        SEAMD_HIP(hipMemcpyAsync(s.values, values + lo * (n / 2), cnt * (n / 2) * sizeof(float),
                                 hipMemcpyHostToDevice, compute));
        SEAMD_HIP(hipMemcpyAsync(s.seeds, seeds + lo * 64, cnt * 64, hipMemcpyHostToDevice, compute));
        if (!asym)
            SEAMD_HIP(hipMemcpyAsync(s.share_seeds, share_seeds + lo * 64, cnt * 64,
                                     hipMemcpyHostToDevice, compute));
        if (k >= kSlots) SEAMD_HIP(hipStreamWaitEvent(compute, s.copied, 0));
        uint8_t *st = (uint8_t *)d_status + lo;
        int r;
        if (asym)
            r = c.encrypt_asym((const float *)s.values, cnt, (const uint8_t *)s.seeds,
                               (uint32_t *)s.c0, (uint32_t *)s.c1, ntt_pte ? (uint32_t *)s.ntt_pte : nullptr,
                               pte ? (int64_t *)s.pte : nullptr, st, compute);
        else
            r = c.encrypt_sym((const float *)s.values, cnt, (const uint8_t *)s.share_seeds,
                              (const uint8_t *)s.seeds, (uint32_t *)s.c0, (uint32_t *)s.c1,
                              ntt_pte ? (uint32_t *)s.ntt_pte : nullptr, pte ? (int64_t *)s.pte : nullptr,
                              st, compute);
        if (r) return r;
        SEAMD_HIP(hipEventRecord(s.computed, compute));
        return 0;
    };

    // the D2H schedule: per chunk c0, c1, (ntt_pte), (pte); staged outputs in ring-sized pieces
    std::vector<Piece> pieces;
    for (size_t k = 0; k < nch; k++)
    {
        Slot &s         = slot[k % kSlots];
        const size_t lo = k * chunk, cnt = chunk_count(k);
        struct Out
        {
            const void *src;
            void *dst;
            size_t bytes;
            bool direct;
        } outs[4] = {
            {s.c0, (char *)c0 + lo * ct_bytes, cnt * ct_bytes, pin0},
            {s.c1, c1 ? (char *)c1 + lo * ct_bytes : nullptr, c1 ? cnt * ct_bytes : 0, pin1},
            {s.ntt_pte, ntt_pte ? (char *)ntt_pte + lo * ct_bytes : nullptr, ntt_pte ? cnt * ct_bytes : 0, pinn},
            {s.pte, pte ? (char *)pte + lo * n * 8 : nullptr, pte ? cnt * n * 8 : 0, pinp},
        };
        const size_t first_idx = pieces.size();
        for (auto &o : outs)
        {
            if (!o.bytes) continue;
            const size_t step = o.direct ? o.bytes : ring_bytes;
            for (size_t off = 0; off < o.bytes; off += step)
                pieces.push_back({(const char *)o.src + off, (char *)o.dst + off,
                                  std::min(step, o.bytes - off), k, false, false, o.direct});
        }
        pieces[first_idx].first = true;
        pieces.back().last      = true;
    }

    std::deque<InFlight> inflight;
    int failure    = 0;
    auto drain_one = [&]() -> int {
        InFlight f = inflight.front();
        inflight.pop_front();
        SEAMD_HIP(hipEventSynchronize(ring_ev[f.r]));
        pool->copy(f.dst, ring[f.r], f.bytes);
        return 0;
    };

    for (size_t k = 0; k < std::min<size_t>(kSlots, nch); k++)
        if ((rc = launch_chunk(k))) return rc;

    size_t ring_next = 0;
    for (const Piece &p : pieces)
    {
        Slot &s = slot[p.chunk % kSlots];
        if (p.first) SEAMD_HIP(hipStreamWaitEvent(copy, s.computed, 0));
        if (p.direct)
            SEAMD_HIP(hipMemcpyAsync(p.dst, p.src, p.bytes, hipMemcpyDeviceToHost, copy));
        else
        {
            while (inflight.size() == (size_t)kRing)
                if ((rc = drain_one())) { failure = rc; break; }
            if (failure) break;
            const int r = int(ring_next++ % kRing);
            SEAMD_HIP(hipMemcpyAsync(ring[r], p.src, p.bytes, hipMemcpyDeviceToHost, copy));
            SEAMD_HIP(hipEventRecord(ring_ev[r], copy));
            inflight.push_back({r, p.dst, p.bytes});
        }
        if (p.last)
        {
            SEAMD_HIP(hipEventRecord(s.copied, copy));
            if (p.chunk + kSlots < nch && (rc = launch_chunk(p.chunk + kSlots))) { failure = rc; break; }
        }
    }
    // the status bytes ride the copy stream behind the last piece (every chunk has been launched by now and the
    // compute stream is in order, so the last chunk's `computed` event covers them all): no synchronous copy of
    // their own after the drain -- 15-20 us of a single call
    if (!failure)
    {
        hipError_t es = hipStreamWaitEvent(copy, slot[(nch - 1) % kSlots].computed, 0);
        if (es == hipSuccess) es = hipMemcpyAsync(h_status, d_status, B, hipMemcpyDeviceToHost, copy);
        if (es != hipSuccess) failure = hip_fail(es, "status download");
    }
    while (!failure && !inflight.empty()) failure = drain_one();
    // always quiesce both streams before returning (also on the error path: the slots are reused)
    hipError_t e1 = hipStreamSynchronize(copy), e2 = hipStreamSynchronize(compute);
    if (failure) return failure;
    if (e1 != hipSuccess) return hip_fail(e1, "hipStreamSynchronize(copy)");
    if (e2 != hipSuccess) return hip_fail(e2, "hipStreamSynchronize(compute)");

    int failed = 0;
    for (size_t i = 0; i < B; i++) failed += h_status[i] ? 0 : 1;
    if (status) memcpy(status, h_status, B);
    return failed;
}

}  // namespace seamd
