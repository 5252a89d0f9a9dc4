// ssp_internal.hpp -- shared declarations of libssp_hip.so (gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <set>
#include <string>
#include <vector>

#include "../../include/ssp.h"
#include "../../include/ssp_math.h"

#define SSP_API extern "C" __attribute__((visibility("default")))

namespace ssp {

// ---- errors --------------------------------------------------------------------------------------
int set_error(int code, const char *fmt, ...);
#define SSP_FAIL(code, ...) return ::ssp::set_error((code), __VA_ARGS__)
#define SSP_HIP(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return ::ssp::set_error(e_ == hipErrorOutOfMemory ? SSP_ERR_MEMORY : SSP_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, \
                                    hipGetErrorString(e_), __FILE__, __LINE__);                                    \
    } while (0)
#define SSP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        int rc_ = (expr);                                                                                          \
        if (rc_) return rc_;                                                                                       \
    } while (0)
#define SSP_REQUIRE(cond, ...)                                                                                     \
    do {                                                                                                           \
        if (!(cond)) return ::ssp::set_error(SSP_ERR_ARG, __VA_ARGS__);                                            \
    } while (0)

// ---- runtime -------------------------------------------------------------------------------------
int ensure_init();
hipStream_t stream();
int pool_alloc(size_t bytes, void **out);  // stream-ordered reuse on the single library stream
void pool_free(void *p);
// the block was read on streams other than the one it lives on: `after` are events recorded behind those reads (ownership passes to
// the pool); whoever takes the block next makes its stream wait for them
void pool_free_after(void *p, std::vector<hipEvent_t> &after);
bool pool_stream_of(const void *p, hipStream_t *home);   // false: not a live pool block; else *home = the stream it was allocated under (nullptr = the null stream)
hipEvent_t pool_event_get();
void pool_event_put(hipEvent_t e);

struct ProfileScope {  // hipEvent pair around one kernel family launch (only when profiling is on)
    ProfileScope(const char *name, double algo_bytes);
    ~ProfileScope();
    int slot;
    hipEvent_t e0, e1;
};
bool profiling();

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Ring of pinned staging buffers + device copies for kernel descriptor arrays: acquire() hands out a host/device pair,
// commit() enqueues the upload, release() marks the point after the last kernel that reads it.  A slot is reused only
// after its release event has completed, so consecutive steps never wait for each other's uploads.
struct DescRing {
    static const int N = 4;
    void *h[N] = {nullptr}, *d[N] = {nullptr};
    hipEvent_t ev[N] = {nullptr};
    size_t cap[N] = {0};
    int next = 0;
    int acquire(size_t bytes, void **host, void **dev, int *slot);
    int commit(int slot, size_t bytes);
    int release(int slot);
    void destroy();
};
// per-composer knowledge about the tiles the LDS-staged warp cannot stage (ssp_warp.hip: warp_batch_launch)
struct WarpRestPlan {
    int state = 0;               // 0 unknown, 1 count on its way to the host, 2 known
    int count = 0, misfit = 0;
    bool has_far = true;         // any far tile in the geometry (k_warp_records_far); until known: assume so
    int capacity = 0;            // entries the list of the learning launch could hold (its tile count)
    long long sig = 0;           // launch shape the knowledge belongs to
    int *h_count = nullptr;      // pinned: {count, misfit, far flag}
    hipEvent_t ev = nullptr;
    int *d_list = nullptr;       // device: the rest list of the first panorama (count at [0]), reused while the geometry stands
    std::vector<char> prep_key;  // the descriptors (per-panorama fields blanked) the prep launch last ran for
};
// beyond this many pixels of every set mask pixel nothing of a fed image reaches the blended panorama (4 * 2^bands: k_warp_records_far,
// live_parts); 0 = unknown band count: nothing is skipped or split
static inline int live_reach(int num_bands) { return (num_bands >= 0 && num_bands <= 12) ? 4 << num_bands : 0; }
static inline int depth_size(int depth) { return depth == SSP_U8 ? 1 : depth == SSP_S16 ? 2 : depth == SSP_F32 ? 4 : 0; }

}  // namespace ssp

// ---- device image ---------------------------------------------------------------------------------
struct ssp_image {
    void *data = nullptr;
    size_t pitch = 0;  // bytes per row (multiple of 16 for pool-owned images)
    int w = 0, h = 0, cn = 0, depth = 0;
    int refs = 1;
    bool owned = true;
    // in-place writers (compensator.apply, fill, the seam finders) bump `version`.  An int16 image made by ssp_image_convert from an owned
    // 8-bit image remembers it (`origin`, retained) with both versions: while neither has been written since, the int16 image IS the 8-bit one
    // sample for sample, and MultiBandBlender.feed takes the 8-bit pyramid path for it (the reference feeds astype(np.int16) of its 8-bit
    // warps, sde.py:1755 -> :1886)
    int version = 0;
    ssp_image *origin = nullptr;
    int origin_ver = 0, self_ver = 0;
    // kernels of ANOTHER stream than the one the image lives on read it (a composer with its own stream warping frames that were
    // uploaded on the home stream): one event per such stream, recorded behind the last read (ssp::image_note_read)
    std::vector<std::pair<hipStream_t, hipEvent_t>> readers;
};

namespace ssp {
int image_new(int w, int h, int cn, int depth, ssp_image **out);
void image_unref(ssp_image *img);
int image_aligned_source(const ssp_image *src, const ssp_image **use, ssp_image **tmp);   // wrapped frames with a tight pitch / odd base: repacked copy in *tmp (unref after the launch)
void image_note_read(ssp_image *img);   // call after enqueuing kernels that read `img` on the current stream
template <typename T>
static inline T *row_ptr(const ssp_image *im, int y) { return (T *)((char *)im->data + (size_t)y * im->pitch); }
}  // namespace ssp

namespace ssp {
// Host worker pool: fn(i) for i in [0, n) on up to `threads` host threads (DpSeamFinder's pairs; first touch of download buffers).  The workers are
// started once per process and parked on a condition variable between calls: a round is a millisecond, starting threads for it (a 256-core host takes 50-100 us
// per thread) cost as much as the work.
class WorkerPool {
public:
    static WorkerPool &get() { static WorkerPool p; return p; }
    template <typename F>
    void run(int n, int threads, F fn)
    {
        threads = std::max(1, std::min(threads, n));
        if (threads == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        // one job at a time: a second host thread that comes while the pool is busy does its work itself
        std::unique_lock<std::mutex> use(use_, std::try_to_lock);
        if (!use.owns_lock()) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::function<void(int)> f = fn;
        {
            std::unique_lock<std::mutex> lk(m_);
            grow(threads - 1);
            job_ = &f; n_ = n; next_ = 0; busy_ = 0; ++epoch_; helpers_ = threads - 1;
        }
        cv_.notify_all();
        work(f, n);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&]() { return busy_ == 0 && next_ >= n_; });
        job_ = nullptr;
    }
private:
    void work(const std::function<void(int)> &f, int n) { for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) f(i); }
    void grow(int want)
    {
        while ((int)threads_.size() < want) {
            threads_.emplace_back([this]() {
                unsigned long long seen = 0;
                for (;;) {
                    const std::function<void(int)> *job; int n;
                    {
                        std::unique_lock<std::mutex> lk(m_);
                        cv_.wait(lk, [&]() { return stop_ || (epoch_ != seen && job_ && helpers_ > 0); });
                        if (stop_) return;
                        seen = epoch_; --helpers_; ++busy_;
                        job = job_; n = n_;
                    }
                    work(*job, n);
                    { std::unique_lock<std::mutex> lk(m_); --busy_; }
                    done_.notify_all();
                }
            });
        }
    }
    ~WorkerPool()
    {
        { std::unique_lock<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    std::mutex m_, use_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)> *job_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, busy_ = 0, helpers_ = 0;
    unsigned long long epoch_ = 0;
    bool stop_ = false;
};
}  // namespace ssp
