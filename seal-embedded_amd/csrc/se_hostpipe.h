// se_hostpipe.h -- internal: the host-pointer entry points (se_encrypt_batch, se_amd_encrypt_*_host,
// the single-item se_encrypt*) as a chunked pipeline over PCIe.
//
// The batched boundary of SURVEY 8(b) hands over pageable host buffers; per ciphertext the library
// writes 8*n*np bytes and reads 2n+128, so the call is bound by the device-to-host link, not by
// the kernels (C2: 196 KiB per ciphertext; the GPU produces 6.4 M/s = 1.2 TB/s of them).  The
// pipeline therefore keeps the link busy and everything else out of its way:
//
//   compute stream : H2D inputs of chunk k, encrypt chunk k into device slot k%2
//   copy stream    : D2H of slot k%2 in 64 MiB pieces into a ring of pinned staging buffers
//                    (directly into the caller's memory when that is already pinned/registered)
//   host           : drains pieces from the ring into the caller's buffers with a small memcpy
//                    thread pool while the next pieces are in flight
//
// Device slots, pinned ring, streams and threads persist in the context (grow-only).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

namespace seamd {

struct Context;

class CopyPool
{
public:
    explicit CopyPool(int nthreads);
    ~CopyPool();
    void copy(void *dst, const void *src, size_t bytes);  // blocking, split over the pool
    int threads() const { return nthreads_; }

private:
    void worker(int id);
    int nthreads_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_job_, cv_done_;
    uint64_t generation_ = 0;
    int pending_         = 0;
    bool stop_           = false;
    char *dst_           = nullptr;
    const char *src_     = nullptr;
    size_t bytes_        = 0;
};

struct HostPipe
{
    static constexpr int kSlots       = 2;
    static constexpr int kRing        = 4;
    static constexpr size_t kPieceMax = size_t(64) << 20;

    struct Slot
    {
        void *values = nullptr, *seeds = nullptr, *share_seeds = nullptr;
        void *c0 = nullptr, *c1 = nullptr, *ntt_pte = nullptr, *pte = nullptr;
        size_t cap = 0, cap_ntt = 0, cap_pte = 0;
        hipEvent_t computed = nullptr, copied = nullptr;
    };

    int device = 0;
    Slot slot[kSlots];
    void *ring[kRing]           = {};
    hipEvent_t ring_ev[kRing]   = {};
    size_t ring_bytes           = 0;
    void *d_status              = nullptr;
    size_t status_cap           = 0;
    uint8_t *h_status           = nullptr;   // pinned: the status bytes come back on the copy stream, behind the last piece
    size_t h_status_cap         = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    CopyPool *pool      = nullptr;
    size_t chunk_override = 0;  // test hook: ciphertexts per chunk (0 = automatic)
    size_t values_bytes_per_ct = 0;   // (n / 2) floats: size of one ciphertext's slice of Slot::values
    void wipe_slot(Slot &s);          // zero the seeds / values a slot holds (before it is freed or regrown)

    ~HostPipe();
    int init(int device);
    int run(Context &c, bool asym, const float *values, size_t B, const uint8_t *share_seeds,
            const uint8_t *seeds, uint32_t *c0, uint32_t *c1, uint32_t *ntt_pte, int64_t *pte,
            uint8_t *status);

private:
    int ensure(Context &c, size_t chunk, size_t B, bool want_ntt, bool want_pte, bool staged);
};

}  // namespace seamd
