// se_multi.cpp -- device-resident multi-GPU entry of the C ABI.
//
// The path shards embarrassingly: every plaintext -> ciphertext is independent (own seeds; keys and
// tables are < 1 MiB and replicated per device).  A *group* holds one context per device; a batch of B
// units is cut into contiguous blocks, block i lives on device i -- inputs AND outputs are device
// pointers on that device -- and every device runs the ordinary batched call on its block, driven by a
// host thread of its own.  There is no collective on the data path.
//
// The one exchange the path has is the final gather of ciphertext records: with gather_root >= 0 every
// device writes its finished block straight into its slice of the root's [B][np][n] slab with
// hipMemcpyPeerAsync on ITS OWN stream -- 7 concurrent writers over 7 different xGMI links into the
// root on an 8-GPU node (xGMI is point to point: a ring would be bound by one link) -- ordered behind
// that device's kernels, so a block starts travelling while slower devices still compute.
//
// The boundary this extends is the reference's (device/lib/seal_embedded.h:91-130: one ciphertext per
// call, host callback); the reference has no batched or multi-device form.  Python callers get the same
// partition through torch.distributed (seal-embedded_amd/sharding.py); this file is the C equivalent.
#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/seal_embedded_amd.h"
#include "se_context.h"

namespace seamd {
const std::string &last_error();
}

namespace {

// A host thread that lives as long as the group and runs one member's share of every call (creating a
// std::thread per device and call cost tens of microseconds on small-batch group calls).
class Worker
{
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void submit(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = std::move(job), busy_ = true;
        }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return !busy_; });
    }

private:
    void loop()
    {
        std::unique_lock<std::mutex> l(m_);
        for (;;)
        {
            cv_.wait(l, [this] { return quit_ || (busy_ && job_); });
            if (quit_) return;
            std::function<void()> job = std::move(job_);
            job_                      = nullptr;
            l.unlock();
            // a long-lived library thread must not take the process down: the job reports its own errors through
            // its captured result slot; anything it throws (bad_alloc while building a message ...) is swallowed here
            // and leaves that slot at the failure value it was initialised with
            try
            {
                job();
            }
            catch (...)
            {
            }
            l.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::thread th_;   // last: the thread starts when every other member is constructed
};

}  // namespace

struct se_amd_group
{
    std::vector<se_amd_ctx *> ctx;
    std::vector<int> device;
    std::vector<hipStream_t> stream;   // one per member, created on that member's device
    std::vector<std::unique_ptr<Worker>> worker;   // members 1 .. ndev-1 (member 0 runs on the calling thread)
    std::mutex call;                               // one multi-device call at a time per group
};

namespace {

enum Mode { kSym, kAsym, kEncode };

struct Block
{
    size_t first, count;
};

Block block_of(size_t B, size_t ndev, size_t i)
{
    // contiguous blocks; the first B % ndev members take one more unit
    const size_t base = B / ndev, extra = B % ndev;
    return Block{i * base + (i < extra ? i : extra), base + (i < extra ? 1 : 0)};
}

int hip_rc(hipError_t e, const char *what, std::string &err)
{
    if (e == hipSuccess) return SE_SUCCESS;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return SE_ERR_HIP;
}

// One member's share of a multi-device call; runs on a host thread of its own.
int run_member(se_amd_group *g, size_t i, Mode mode, size_t B, const float *const *d_values,
               const uint8_t *const *d_share_seeds, const uint8_t *const *d_seeds, uint32_t *const *d_c0,
               uint32_t *const *d_c1, uint8_t *const *d_status, int gather_root, uint32_t *d_c0_all,
               uint32_t *d_c1_all, std::string &err)
{
    const size_t ndev = g->ctx.size();
    const Block blk   = block_of(B, ndev, i);
    se_amd_ctx *ctx   = g->ctx[i];
    const size_t rec  = ctx->c.hp.nprimes * ctx->c.hp.n;   // words per record
    hipStream_t st    = g->stream[i];
    int rc            = hip_rc(hipSetDevice(g->device[i]), "hipSetDevice", err);
    if (rc) return rc;
    // Whatever happens below, this member's stream is drained before the call returns: the caller may free or
    // reuse its buffers (and the root's slab) the moment it has the return code.  The first error is kept.
    auto finish = [&](int first_rc) {
        std::string sync_err;
        const int src = hip_rc(hipStreamSynchronize(st), "hipStreamSynchronize", sync_err);
        if (first_rc == SE_SUCCESS && src != SE_SUCCESS) err = sync_err;
        return first_rc != SE_SUCCESS ? first_rc : src;
    };
    if (blk.count)
    {
        uint8_t *status = d_status ? d_status[i] : nullptr;
        uint32_t *c1    = d_c1 ? d_c1[i] : nullptr;
        switch (mode)
        {
            case kSym:
                rc = c1 ? se_amd_encrypt_sym_device(ctx, d_values[i], blk.count, d_share_seeds[i], d_seeds[i],
                                                    d_c0[i], c1, nullptr, nullptr, status, st)
                        : se_amd_encrypt_sym_seeded_device(ctx, d_values[i], blk.count, d_share_seeds[i],
                                                           d_seeds[i], d_c0[i], status, st);
                break;
            case kAsym:
                rc = se_amd_encrypt_asym_device(ctx, d_values[i], blk.count, d_seeds[i], d_c0[i], c1, nullptr,
                                                nullptr, status, st);
                break;
            default:
                rc = se_amd_encode_ntt_device(ctx, d_values[i], blk.count, d_c0[i], nullptr, status, st);
        }
        if (rc != SE_SUCCESS)
        {
            err = se_amd_last_error();   // thread-local: carried to the calling thread
            return finish(rc);
        }
        if (gather_root >= 0)
        {
            // this member's block -> its slice of the root's slab, on this member's stream (behind its
            // kernels).  A block that was produced in place inside the slab needs no copy.
            const int root       = g->device[(size_t)gather_root];
            const size_t bytes   = blk.count * rec * sizeof(uint32_t);
            uint32_t *dst0       = d_c0_all + blk.first * rec;
            if (dst0 != d_c0[i])
            {
                rc = hip_rc(hipMemcpyPeerAsync(dst0, root, d_c0[i], g->device[i], bytes, st), "hipMemcpyPeerAsync(c0)",
                            err);
                if (rc) return finish(rc);
            }
            if (d_c1_all && c1)
            {
                uint32_t *dst1 = d_c1_all + blk.first * rec;
                if (dst1 != c1)
                {
                    rc = hip_rc(hipMemcpyPeerAsync(dst1, root, c1, g->device[i], bytes, st), "hipMemcpyPeerAsync(c1)",
                                err);
                    if (rc) return finish(rc);
                }
            }
        }
    }
    return finish(SE_SUCCESS);
}

int run_group(se_amd_group *g, Mode mode, size_t B, const float *const *d_values,
              const uint8_t *const *d_share_seeds, const uint8_t *const *d_seeds, uint32_t *const *d_c0,
              uint32_t *const *d_c1, uint8_t *const *d_status, int gather_root, uint32_t *d_c0_all,
              uint32_t *d_c1_all)
{
    if (!g || g->ctx.empty() || !d_values || !d_c0 || (mode != kEncode && !d_seeds) || (mode == kSym && !d_share_seeds) ||
        (mode == kAsym && !d_c1))
    {
        seamd::set_last_error("multi-device call: a required pointer array is NULL");
        return SE_ERR_INVALD_ARGUMENT;
    }
    const size_t ndev = g->ctx.size();
    if (gather_root >= (int)ndev || (gather_root >= 0 && !d_c0_all))
    {
        seamd::set_last_error("multi-device call: gather_root out of range, or no slab to gather into");
        return SE_ERR_INVALD_ARGUMENT;
    }
    for (size_t i = 0; i < ndev; i++)
    {
        const Block blk = block_of(B, ndev, i);
        if (!blk.count) continue;
        const bool need_c1 = mode == kAsym || (gather_root >= 0 && d_c1_all != nullptr);
        if (!d_values[i] || !d_c0[i] || (mode != kEncode && !d_seeds[i]) || (mode == kSym && !d_share_seeds[i]) ||
            (need_c1 && mode != kEncode && !(d_c1 && d_c1[i])))
        {
            seamd::set_last_error("multi-device call: member " + std::to_string(i) + " has a NULL buffer");
            return SE_ERR_INVALD_ARGUMENT;
        }
    }
    std::lock_guard<std::mutex> one_call(g->call);
    // a member's slot starts at "failed": only a job that ran to its end overwrites it (Worker::loop swallows what a
    // job throws so that a library thread never ends the process)
    std::vector<int> rcs(ndev, SE_ERR_HIP);
    std::vector<std::string> errs(ndev, "the member's worker thread did not finish its job (exception)");
    for (size_t i = 1; i < ndev; i++)
        g->worker[i - 1]->submit([&, i] {
            rcs[i] = run_member(g, i, mode, B, d_values, d_share_seeds, d_seeds, d_c0, d_c1, d_status, gather_root,
                                d_c0_all, d_c1_all, errs[i]);
        });
    rcs[0] = run_member(g, 0, mode, B, d_values, d_share_seeds, d_seeds, d_c0, d_c1, d_status, gather_root, d_c0_all,
                        d_c1_all, errs[0]);
    for (auto &w : g->worker) w->wait();   // every member has drained its stream (run_member), error or not
    for (size_t i = 0; i < ndev; i++)
        if (rcs[i] != SE_SUCCESS)
        {
            seamd::set_last_error("member " + std::to_string(i) + " (device " + std::to_string(g->device[i]) + "): " +
                                  errs[i]);
            return rcs[i];
        }
    return SE_SUCCESS;
}

}  // namespace

extern "C" {

int se_amd_group_create(se_amd_group **out, size_t degree, size_t nprimes, const int *devices, size_t ndev)
{
    if (!out) return SE_ERR_INVALD_ARGUMENT;
    *out = nullptr;
    std::vector<int> list;
    if (devices && ndev)
        list.assign(devices, devices + ndev);
    else
    {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        {
            seamd::set_last_error("no HIP device visible");
            return SE_ERR_NO_DEVICE;
        }
        for (int d = 0; d < count; d++) list.push_back(d);
    }
    se_amd_group *g = new (std::nothrow) se_amd_group();
    if (!g) return SE_ERR_NO_MEMORY;
    for (int d : list)
    {
        se_amd_ctx *c = nullptr;
        int rc        = se_amd_create(&c, degree, nprimes, d);
        hipStream_t st = nullptr;
        if (rc == SE_SUCCESS && (hipSetDevice(d) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess))
        {
            seamd::set_last_error("cannot create a stream on device " + std::to_string(d));
            rc = SE_ERR_HIP;
        }
        if (rc != SE_SUCCESS)
        {
            if (c) se_amd_destroy(c);
            se_amd_group_destroy(g);
            return rc;
        }
        g->ctx.push_back(c), g->device.push_back(d), g->stream.push_back(st);
    }
    // peer access for the gather (every member writes into whichever member is the root); "already
    // enabled" and "same device" are not errors, a pair without a link falls back to staged copies
    for (size_t i = 0; i < list.size(); i++)
        for (size_t j = 0; j < list.size(); j++)
            if (list[i] != list[j] && hipSetDevice(list[i]) == hipSuccess)
            {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, list[i], list[j]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(list[j], 0);
                (void)hipGetLastError();
            }
    for (size_t i = 1; i < list.size(); i++) g->worker.emplace_back(new Worker());
    *out = g;
    return SE_SUCCESS;
}

void se_amd_group_destroy(se_amd_group *g)
{
    if (!g) return;
    g->worker.clear();   // joins the member threads
    for (size_t i = 0; i < g->ctx.size(); i++)
    {
        if (g->stream[i] && hipSetDevice(g->device[i]) == hipSuccess) (void)hipStreamDestroy(g->stream[i]);
        se_amd_destroy(g->ctx[i]);
    }
    delete g;
}

size_t se_amd_group_size(const se_amd_group *g) { return g ? g->ctx.size() : 0; }

se_amd_ctx *se_amd_group_ctx(se_amd_group *g, size_t i) { return g && i < g->ctx.size() ? g->ctx[i] : nullptr; }

int se_amd_group_device(const se_amd_group *g, size_t i) { return g && i < g->device.size() ? g->device[i] : -1; }

int se_amd_group_partition(const se_amd_group *g, size_t B, size_t *first, size_t *count)
{
    if (!g || g->ctx.empty()) return SE_ERR_INVALD_ARGUMENT;
    for (size_t i = 0; i < g->ctx.size(); i++)
    {
        const Block b = block_of(B, g->ctx.size(), i);
        if (first) first[i] = b.first;
        if (count) count[i] = b.count;
    }
    return SE_SUCCESS;
}

int se_amd_group_set_secret_key(se_amd_group *g, const uint8_t *sk_packed)
{
    if (!g || !sk_packed) return SE_ERR_INVALD_ARGUMENT;
    for (se_amd_ctx *c : g->ctx)
        if (int rc = se_amd_set_secret_key(c, sk_packed)) return rc;
    return SE_SUCCESS;
}

int se_amd_group_set_public_key(se_amd_group *g, const uint32_t *pk0, const uint32_t *pk1)
{
    if (!g || !pk0 || !pk1) return SE_ERR_INVALD_ARGUMENT;
    for (se_amd_ctx *c : g->ctx)
        if (int rc = se_amd_set_public_key(c, pk0, pk1)) return rc;
    return SE_SUCCESS;
}

int se_amd_group_reserve(se_amd_group *g, size_t B)
{
    if (!g || g->ctx.empty()) return SE_ERR_INVALD_ARGUMENT;
    for (size_t i = 0; i < g->ctx.size(); i++)
        if (int rc = se_amd_reserve(g->ctx[i], block_of(B, g->ctx.size(), i).count)) return rc;
    return SE_SUCCESS;
}

int se_amd_encrypt_sym_multi_device(se_amd_group *g, size_t B, const float *const *d_values,
                                    const uint8_t *const *d_share_seeds, const uint8_t *const *d_seeds,
                                    uint32_t *const *d_c0, uint32_t *const *d_c1, uint8_t *const *d_status,
                                    int gather_root, uint32_t *d_c0_all, uint32_t *d_c1_all)
{
    return run_group(g, kSym, B, d_values, d_share_seeds, d_seeds, d_c0, d_c1, d_status, gather_root, d_c0_all,
                     d_c1_all);
}

int se_amd_encrypt_asym_multi_device(se_amd_group *g, size_t B, const float *const *d_values,
                                     const uint8_t *const *d_seeds, uint32_t *const *d_c0, uint32_t *const *d_c1,
                                     uint8_t *const *d_status, int gather_root, uint32_t *d_c0_all,
                                     uint32_t *d_c1_all)
{
    return run_group(g, kAsym, B, d_values, nullptr, d_seeds, d_c0, d_c1, d_status, gather_root, d_c0_all, d_c1_all);
}

int se_amd_encode_ntt_multi_device(se_amd_group *g, size_t B, const float *const *d_values, uint32_t *const *d_out,
                                   uint8_t *const *d_status, int gather_root, uint32_t *d_out_all)
{
    return run_group(g, kEncode, B, d_values, nullptr, nullptr, d_out, nullptr, d_status, gather_root, d_out_all,
                     nullptr);
}

}  // extern "C"
