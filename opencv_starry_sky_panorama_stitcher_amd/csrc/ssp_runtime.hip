// ssp_runtime.hip -- device selection, stream, HBM pool allocator, image handles, timers, per-kernel profile.
#include "ssp_internal.hpp"
#include <chrono>
#include <sys/mman.h>
#include <unistd.h>

namespace ssp {

static thread_local char g_err[1024];
int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

static std::mutex g_mu;
static bool g_inited = false;
static int g_device = -1;
static hipStream_t g_stream = nullptr;
static bool g_stream_owned = false;
static hipStream_t g_home_stream = nullptr;  // what ssp_use_stream(NULL) returns to: the own stream, or the one given to ssp_set_stream

// ---- pool: size-bucketed free lists PER STREAM.  Work on one stream is ordered, so a block freed on the host can be handed out
// again immediately to the same stream: kernels that used it were enqueued before the next user's kernels.  A block freed under
// another stream than it was allocated under waits for the stream it was born on.  With several streams (two panoramas in flight)
// an image may also have READERS on a third stream -- a frame uploaded on the home stream and warped by a composer on its own
// stream, then released while yet another stream is current.  Those reads are registered where they happen
// (image_note_read: an event behind the reading kernels, kept with the image); image_unref parks the events with the block and
// whoever takes the block next makes its stream wait for them (hipStreamWaitEvent: device side, the host never blocks).
typedef std::pair<hipStream_t, size_t> FreeKey;
struct FreeBlock { void *p; std::vector<hipEvent_t> after; };
static std::multimap<FreeKey, FreeBlock> g_free;
struct LiveBlock { size_t bytes; hipStream_t stream; };
static std::map<void *, LiveBlock> g_live;
static size_t g_in_use = 0, g_cached = 0;
static std::set<hipStream_t> g_streams;          // every stream the library has run on and that is still alive
static std::vector<hipEvent_t> g_event_pool;
hipEvent_t event_get_locked();

hipEvent_t pool_event_get()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return event_get_locked();
}
void pool_event_put(hipEvent_t e)
{
    if (!e) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_event_pool.push_back(e);
}
bool pool_stream_of(const void *p, hipStream_t *home)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find((void *)p);
    if (it == g_live.end()) return false;
    *home = it->second.stream;                       // may be nullptr: HIP's null stream is a legitimate home (ssp_set_stream(NULL), torch's default stream)
    return true;
}
hipEvent_t event_get_locked()
{
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}
static void release_cached(FreeBlock &fb)          // caller holds g_mu; hipFree waits for the device itself
{
    for (hipEvent_t e : fb.after) g_event_pool.push_back(e);
    fb.after.clear();
    (void)hipFree(fb.p);
}

static size_t bucket(size_t bytes)
{
    if (bytes < 4096) return 4096;
    // round up to 1/8 of the power of two below: <= 12.5 % internal waste, few distinct sizes
    size_t p = 4096;
    while (p * 2 <= bytes) p *= 2;
    size_t step = p / 8;
    return (bytes + step - 1) / step * step;
}

int pool_alloc(size_t bytes, void **out)
{
    SSP_TRY(ensure_init());
    size_t b = bucket(bytes + 256);  // 256 B of slack: vector loads may touch a few bytes past the last row
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_free.find(FreeKey(g_stream, b));
    if (it != g_free.end()) {
        *out = it->second.p;
        for (hipEvent_t e : it->second.after) {      // readers on other streams at the time of the free
            (void)hipStreamWaitEvent(g_stream, e, 0);
            g_event_pool.push_back(e);               // the wait has captured the event's state: it may be recorded again
        }
        g_free.erase(it);
        g_cached -= b;
    } else {
        hipError_t e = hipMalloc(out, b);
        if (e != hipSuccess) {
            // give cached blocks back and retry once
            for (auto &kv : g_free) release_cached(kv.second);
            g_free.clear();
            g_cached = 0;
            e = hipMalloc(out, b);
            if (e != hipSuccess) return set_error(SSP_ERR_MEMORY, "hipMalloc(%zu) failed: %s", b, hipGetErrorString(e));
        }
    }
    g_live[*out] = LiveBlock{b, g_stream};
    g_in_use += b;
    // SSP_POOL_POISON=1 (tests): every block handed out starts as 0xFF bytes (NaN as float32, -1 as int16, 255 as a mask) -- anything the
    // kernels leave unwritten on purpose (image bytes under far tiles, slack behind rows) must not be able to reach a result
    static const bool poison = getenv("SSP_POOL_POISON") != nullptr;
    if (poison) (void)hipMemsetAsync(*out, 0xFF, b, g_stream);
    return 0;
}

static void pool_free_impl(void *p, std::vector<hipEvent_t> *after)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;
    const size_t b = it->second.bytes;
    const hipStream_t born = it->second.stream;
    g_live.erase(it);
    g_in_use -= b;
    // freed under another stream than it was allocated under: its users ran on the stream it was born on
    if (born != g_stream && g_streams.count(born)) (void)hipStreamSynchronize(born);
    FreeBlock fb{p, {}};
    if (after) fb.after.swap(*after);
    g_free.insert({FreeKey(g_stream, b), std::move(fb)});
    g_cached += b;
}
void pool_free(void *p) { pool_free_impl(p, nullptr); }
void pool_free_after(void *p, std::vector<hipEvent_t> &after) { pool_free_impl(p, &after); }

int ensure_init()
{
    if (g_inited) return 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return set_error(SSP_ERR_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                         e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    int dev = g_device >= 0 ? g_device : 0;
    SSP_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    SSP_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(SSP_ERR_DEVICE, "device %d is %s; this library contains gfx950 (MI355X) code objects only", dev, prop.gcnArchName);
    if (!g_stream) {
        SSP_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        g_stream_owned = true;
    }
    g_home_stream = g_stream;
    g_streams.insert(g_stream);
    g_device = dev;
    g_inited = true;
    return 0;
}

hipStream_t stream() { return g_stream; }

// ---- per-kernel profile -------------------------------------------------------------------------------
struct ProfEntry {
    std::string name;
    int launches = 0;
    double algo_bytes = 0;
    float ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
static bool g_prof = false;
static std::vector<ProfEntry> g_prof_entries;
bool profiling() { return g_prof; }

ProfileScope::ProfileScope(const char *name, double algo_bytes) : slot(-1), e0(nullptr), e1(nullptr)
{
    if (!g_prof) return;
    for (size_t i = 0; i < g_prof_entries.size(); ++i)
        if (g_prof_entries[i].name == name) slot = (int)i;
    if (slot < 0) {
        g_prof_entries.push_back(ProfEntry());
        slot = (int)g_prof_entries.size() - 1;
        g_prof_entries[slot].name = name;
    }
    g_prof_entries[slot].launches++;
    g_prof_entries[slot].algo_bytes += algo_bytes;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, g_stream);
}
ProfileScope::~ProfileScope()
{
    if (slot < 0) return;
    (void)hipEventRecord(e1, g_stream);
    g_prof_entries[slot].pending.push_back({e0, e1});
}
static void profile_drain()
{
    for (auto &pe : g_prof_entries) {
        for (auto &ev : pe.pending) {
            float ms = 0;
            (void)hipEventSynchronize(ev.second);
            (void)hipEventElapsedTime(&ms, ev.first, ev.second);
            pe.ms += ms;
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        pe.pending.clear();
    }
}

// ---- descriptor ring ------------------------------------------------------------------------------------
int DescRing::acquire(size_t bytes, void **host, void **dev, int *slot)
{
    SSP_TRY(ensure_init());
    const int s = next;
    next = (next + 1) % N;
    if (ev[s]) SSP_HIP(hipEventSynchronize(ev[s]));  // blocks only when N uploads are still in flight
    else SSP_HIP(hipEventCreateWithFlags(&ev[s], hipEventDisableTiming));
    if (cap[s] < bytes) {
        if (h[s]) (void)hipHostFree(h[s]);
        pool_free(d[s]);
        h[s] = nullptr; d[s] = nullptr; cap[s] = 0;
        size_t want = align_up(bytes * 2, 4096);
        SSP_HIP(hipHostMalloc(&h[s], want, hipHostMallocDefault));
        SSP_TRY(pool_alloc(want, &d[s]));
        cap[s] = want;
    }
    *host = h[s]; *dev = d[s]; *slot = s;
    return 0;
}
int DescRing::commit(int slot, size_t bytes)
{
    SSP_HIP(hipMemcpyAsync(d[slot], h[slot], bytes, hipMemcpyHostToDevice, stream()));
    return 0;
}
int DescRing::release(int slot)
{
    SSP_HIP(hipEventRecord(ev[slot], stream()));
    return 0;
}
void DescRing::destroy()
{
    for (int s = 0; s < N; ++s) {
        if (ev[s]) { (void)hipEventSynchronize(ev[s]); (void)hipEventDestroy(ev[s]); ev[s] = nullptr; }
        if (h[s]) { (void)hipHostFree(h[s]); h[s] = nullptr; }
        pool_free(d[s]); d[s] = nullptr; cap[s] = 0;
    }
}

// ---- images -------------------------------------------------------------------------------------------
int image_new(int w, int h, int cn, int depth, ssp_image **out)
{
    SSP_REQUIRE(w > 0 && h > 0 && cn >= 1 && cn <= 4 && depth_size(depth) > 0, "image: bad geometry %dx%dx%d depth %d", w, h, cn, depth);
    ssp_image *im = new ssp_image();
    im->w = w; im->h = h; im->cn = cn; im->depth = depth;
    im->pitch = align_up((size_t)w * cn * depth_size(depth), 16);
    int rc = pool_alloc(im->pitch * (size_t)h, &im->data);
    if (rc) { delete im; return rc; }
    *out = im;
    return 0;
}
void image_unref(ssp_image *im)
{
    if (!im) return;
    if (--im->refs > 0) return;
    std::vector<hipEvent_t> after;
    for (auto &r : im->readers) after.push_back(r.second);
    if (im->owned) pool_free_after(im->data, after);
    for (hipEvent_t e : after) pool_event_put(e);      // not a pool block: nothing to guard
    ssp_image *origin = im->origin;
    delete im;
    image_unref(origin);
}
// The LDS-staged warp and the aligned-window gathers read 16-byte chunks at (row * pitch + aligned column offset): pool images have a 256-byte
// aligned base, a pitch that is a multiple of 16 and slack behind the last row.  A wrapped caller buffer (ssp_image_wrap: tight pitch = 3 w,
// any base) has none of that -- with 3 w % 4 != 0 a chunk of the last row is partly out of the buffer range and comes back as zeros.  Such a
// frame is repacked into a pool image for the duration of the call (one device-to-device pass); aligned wrapped frames are used in place.
int image_aligned_source(const ssp_image *src, const ssp_image **use, ssp_image **tmp)
{
    *use = src; *tmp = nullptr;
    if (!src || src->owned || (src->pitch % 16 == 0 && (uintptr_t)src->data % 16 == 0)) return 0;
    SSP_TRY(image_new(src->w, src->h, src->cn, src->depth, tmp));
    const size_t row = (size_t)src->w * src->cn * depth_size(src->depth);
    hipError_t e = hipMemcpy2DAsync((*tmp)->data, (*tmp)->pitch, src->data, src->pitch, row, (size_t)src->h, hipMemcpyDeviceToDevice, g_stream);
    if (e != hipSuccess) { image_unref(*tmp); *tmp = nullptr; return set_error(SSP_ERR_DEVICE, "repacking a wrapped frame failed: %s", hipGetErrorString(e)); }
    *use = *tmp;
    return 0;
}
void image_note_read(ssp_image *im)
{
    if (!im || !im->owned) return;
    hipStream_t home = nullptr;
    const hipStream_t cur = stream();
    if (!pool_stream_of(im->data, &home) || home == cur) return;      // not a pool block, or the same stream: ordered by the stream itself
    for (auto &r : im->readers)
        if (r.first == cur) { (void)hipEventRecord(r.second, cur); return; }
    hipEvent_t e = pool_event_get();
    if (!e || hipEventRecord(e, cur) != hipSuccess) { pool_event_put(e); (void)hipStreamSynchronize(cur); return; }
    im->readers.push_back({cur, e});
}

}  // namespace ssp

using namespace ssp;

SSP_API const char *ssp_last_error(void) { return ssp::g_err; }
SSP_API int ssp_version(void) { return 100; }

SSP_API int ssp_init(int device)
{
    if (g_inited && device == g_device) return 0;
    if (g_inited) SSP_FAIL(SSP_ERR_STATE, "already initialised on device %d", g_device);
    g_device = device;
    return ensure_init();
}
SSP_API int ssp_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = e == hipSuccess ? n : 0;
    return 0;
}
SSP_API int ssp_device_name(char *buf, int len)
{
    SSP_TRY(ensure_init());
    hipDeviceProp_t prop;
    SSP_HIP(hipGetDeviceProperties(&prop, g_device));
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}
SSP_API int ssp_sync(void)
{
    SSP_TRY(ensure_init());
    SSP_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
// ---- extra streams: two panoramas in flight (bench.py --pipeline) ----------------------------------------------------------------
SSP_API int ssp_stream_create(void **out)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(out, "stream_create: null output");
    hipStream_t st = nullptr;
    SSP_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_streams.insert(st);
    }
    *out = st;
    return 0;
}
SSP_API int ssp_stream_destroy(void *s)
{
    if (!s) return 0;
    SSP_REQUIRE((hipStream_t)s != g_stream, "stream_destroy: the stream is in use (ssp_use_stream another one first)");
    (void)hipStreamSynchronize((hipStream_t)s);
    {
        std::lock_guard<std::mutex> lk(g_mu);   // its cached blocks can never be reused
        g_streams.erase((hipStream_t)s);
        for (auto it = g_free.begin(); it != g_free.end();) {
            if (it->first.first == (hipStream_t)s) { release_cached(it->second); g_cached -= it->first.second; it = g_free.erase(it); }
            else ++it;
        }
        // live blocks that were allocated under it: everything queued there has finished, so they now count as the home stream's
        for (auto &kv : g_live)
            if (kv.second.stream == (hipStream_t)s) kv.second.stream = g_home_stream;
    }
    (void)hipStreamDestroy((hipStream_t)s);
    return 0;
}
SSP_API int ssp_stream_sync(void *s)
{
    SSP_TRY(ensure_init());
    SSP_HIP(hipStreamSynchronize(s ? (hipStream_t)s : g_home_stream));
    return 0;
}
// switch the stream every following call runs on, WITHOUT synchronising (NULL: the library's own stream)
SSP_API int ssp_use_stream(void *s)
{
    SSP_TRY(ensure_init());
    g_stream = s ? (hipStream_t)s : g_home_stream;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_streams.insert(g_stream);     // a caller-owned stream handed in here is tracked like the library's own
    }
    return 0;
}

SSP_API int ssp_current_stream(void **out)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(out, "current_stream: null output");
    *out = (void *)g_stream;
    return 0;
}

// Make `s` (e.g. torch's current stream) the library's home stream.  Everything queued so far is waited for; the blocks the old home
// stream had cached or allocated move to the new one; the old home stream is destroyed only if the library created it.
SSP_API int ssp_set_stream(void *s)
{
    SSP_TRY(ensure_init());
    const hipStream_t old_home = g_home_stream, nu = (hipStream_t)s;
    SSP_HIP(hipStreamSynchronize(g_stream));
    if (old_home != g_stream) SSP_HIP(hipStreamSynchronize(old_home));
    if (nu == old_home) { g_stream = nu; return 0; }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        std::vector<std::pair<FreeKey, FreeBlock>> moved;
        for (auto it = g_free.begin(); it != g_free.end();) {
            if (it->first.first == old_home) { moved.push_back({FreeKey(nu, it->first.second), std::move(it->second)}); it = g_free.erase(it); }
            else ++it;
        }
        for (auto &m : moved) g_free.insert(std::move(m));
        for (auto &kv : g_live)
            if (kv.second.stream == old_home) kv.second.stream = nu;
        g_streams.erase(old_home);
        g_streams.insert(nu);
    }
    if (g_stream_owned) (void)hipStreamDestroy(old_home);      // the owned stream is always the home stream it was created as
    g_stream = nu;
    g_home_stream = nu;
    g_stream_owned = false;
    return 0;
}
// device-to-device copy on the library stream (tests emulate the multi-GPU transport with it)
SSP_API int ssp_device_copy(void *dst, const void *src, size_t bytes)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(dst && src, "device_copy: null pointer");
    if (bytes) SSP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_stream));
    return 0;
}
// the same copy as a kernel of this library: 16 bytes per lane, grid-stride, every work-group a contiguous 4 KB per pass -- the "float4 copy"
// MI355X_MICROARCH.md quotes the achievable HBM rate with (6.29 TB/s); bench.py's copy ceiling.  bytes % 16 == 0, 16-byte aligned pointers.
typedef uint32_t copy_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const copy_u32x4 *__restrict__ src, copy_u32x4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const copy_u32x4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}
SSP_API int ssp_device_copy_kernel(void *dst, const void *src, size_t bytes)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(dst && src && bytes % 16 == 0 && (uintptr_t)dst % 16 == 0 && (uintptr_t)src % 16 == 0, "device_copy_kernel: 16-byte aligned pointers and size");
    if (!bytes) return 0;
    const size_t n16 = bytes / 16;
    const int grid = (int)std::min<size_t>((n16 + 255) / 256, 256 * 32);      // 32 work-groups per CU: enough loads in flight, few passes each
    hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, g_stream, (const copy_u32x4 *)src, (copy_u32x4 *)dst, n16);
    SSP_HIP(hipGetLastError());
    return 0;
}
SSP_API int ssp_pool_stats(size_t *in_use, size_t *cached)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (in_use) *in_use = g_in_use;
    if (cached) *cached = g_cached;
    return 0;
}
SSP_API int ssp_pool_trim(void)
{
    SSP_TRY(ensure_init());
    SSP_HIP(hipStreamSynchronize(g_stream));
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_free) release_cached(kv.second);
    g_free.clear();
    g_cached = 0;
    return 0;
}

struct ssp_timer {
    hipEvent_t e0, e1;
};
SSP_API int ssp_timer_create(ssp_timer **t)
{
    SSP_TRY(ensure_init());
    ssp_timer *x = new ssp_timer();
    SSP_HIP(hipEventCreate(&x->e0));
    SSP_HIP(hipEventCreate(&x->e1));
    *t = x;
    return 0;
}
SSP_API int ssp_timer_start(ssp_timer *t) { SSP_HIP(hipEventRecord(t->e0, g_stream)); return 0; }
SSP_API int ssp_timer_stop(ssp_timer *t) { SSP_HIP(hipEventRecord(t->e1, g_stream)); return 0; }
SSP_API int ssp_timer_elapsed_ms(ssp_timer *t, float *ms)
{
    SSP_HIP(hipEventSynchronize(t->e1));
    SSP_HIP(hipEventElapsedTime(ms, t->e0, t->e1));
    return 0;
}
SSP_API int ssp_timer_destroy(ssp_timer *t)
{
    if (!t) return 0;
    (void)hipEventDestroy(t->e0);
    (void)hipEventDestroy(t->e1);
    delete t;
    return 0;
}

SSP_API int ssp_profile_enable(int on) { g_prof = on != 0; return 0; }
SSP_API int ssp_profile_reset(void)
{
    profile_drain();
    g_prof_entries.clear();
    return 0;
}
SSP_API int ssp_profile_count(int *n)
{
    profile_drain();
    *n = (int)g_prof_entries.size();
    return 0;
}
SSP_API int ssp_profile_get(int idx, char *name, int name_len, int *launches, float *total_ms, double *algo_bytes)
{
    profile_drain();
    SSP_REQUIRE(idx >= 0 && idx < (int)g_prof_entries.size(), "profile index out of range");
    const ProfEntry &pe = g_prof_entries[idx];
    snprintf(name, name_len, "%s", pe.name.c_str());
    *launches = pe.launches;
    *total_ms = pe.ms;
    *algo_bytes = pe.algo_bytes;
    return 0;
}

// ---- image API ---------------------------------------------------------------------------------------
SSP_API int ssp_image_create(int w, int h, int cn, int depth, ssp_image **out) { return image_new(w, h, cn, depth, out); }

SSP_API int ssp_image_upload(const void *host, int w, int h, int cn, int depth, ssp_image **out)
{
    SSP_REQUIRE(host != nullptr, "upload: null host pointer");
    ssp_image *im = nullptr;
    SSP_TRY(image_new(w, h, cn, depth, &im));
    size_t row = (size_t)w * cn * depth_size(depth);
    // Host memory always moves as ONE linear copy (55 GB/s from pageable memory on this platform); a pitched host<->device copy goes
    // row by row through the runtime's staging buffers and measured 8x slower.  Rows that need padding are spread on the device.
    hipError_t e;
    void *tmp = nullptr;
    if (im->pitch == row) {
        e = hipMemcpyAsync(im->data, host, row * (size_t)h, hipMemcpyHostToDevice, g_stream);
    } else {
        int rc = pool_alloc(row * (size_t)h, &tmp);
        if (rc) { image_unref(im); return rc; }
        e = hipMemcpyAsync(tmp, host, row * (size_t)h, hipMemcpyHostToDevice, g_stream);
        if (e == hipSuccess) e = hipMemcpy2DAsync(im->data, im->pitch, tmp, row, row, h, hipMemcpyDeviceToDevice, g_stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);  // the host buffer is only borrowed for this call
    pool_free(tmp);
    if (e != hipSuccess) {
        image_unref(im);
        SSP_FAIL(SSP_ERR_DEVICE, "upload failed: %s", hipGetErrorString(e));
    }
    *out = im;
    return 0;
}

SSP_API int ssp_image_wrap(void *dev_ptr, size_t pitch, int w, int h, int cn, int depth, ssp_image **out)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE(dev_ptr && w > 0 && h > 0 && cn >= 1 && cn <= 4 && depth_size(depth) > 0, "wrap: bad arguments");
    size_t row = (size_t)w * cn * depth_size(depth);
    if (pitch == 0) pitch = row;
    SSP_REQUIRE(pitch >= row, "wrap: pitch %zu smaller than a row (%zu)", pitch, row);
    ssp_image *im = new ssp_image();
    im->data = dev_ptr; im->pitch = pitch; im->w = w; im->h = h; im->cn = cn; im->depth = depth;
    im->owned = false;
    *out = im;
    return 0;
}

// A download into memory nobody has touched yet (numpy.empty: what UMat.get() hands over) faults its pages in one by one inside the copy: 131 MB of
// mosaic came down at 11.6 GB/s instead of 50 (tools/pcie_probe.py).  The destination is overwritten as a whole, so its pages can be touched first --
// by the worker pool, 4 MB per task -- when a look at three of them (mincore) says they are not there yet.
static void first_touch(void *host, size_t bytes)
{
    if (bytes < ((size_t)8 << 20)) return;
    const size_t ps = (size_t)sysconf(_SC_PAGESIZE);
    char *base = (char *)host;
    bool fresh = false;
    for (int k = 0; k < 3 && !fresh; ++k) {
        const uintptr_t page = ((uintptr_t)base + (bytes - 1) * (size_t)k / 2) & ~(uintptr_t)(ps - 1);
        unsigned char v = 1;
        if (mincore((void *)page, ps, &v) == 0 && !(v & 1)) fresh = true;
    }
    if (!fresh) return;
    const size_t chunk = (size_t)4 << 20;
    const int n = (int)((bytes + chunk - 1) / chunk);
    static const bool timing = getenv("SSP_TOUCH_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int hw = (int)std::thread::hardware_concurrency();
    WorkerPool::get().run(n, std::max(1, std::min(hw > 0 ? hw : 1, 16)), [&](int i) {
        char *p = base + (size_t)i * chunk, *e = base + std::min(bytes, (size_t)(i + 1) * chunk);
        for (; p < e; p += ps) *(volatile char *)p = 0;
        *(volatile char *)(e - 1) = 0;
    });
    if (timing) fprintf(stderr, "first_touch: %zu MB in %.2f ms\n", bytes >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

SSP_API int ssp_image_download(const ssp_image *im, void *host)
{
    SSP_REQUIRE(im && host, "download: null argument");
    size_t row = (size_t)im->w * im->cn * depth_size(im->depth);
    first_touch(host, row * (size_t)im->h);
    if (im->pitch == row) {
        SSP_HIP(hipMemcpyAsync(host, im->data, row * (size_t)im->h, hipMemcpyDeviceToHost, g_stream));
        SSP_HIP(hipStreamSynchronize(g_stream));
        return 0;
    }
    // padded rows: pack on the device, then one linear copy (see ssp_image_upload)
    void *tmp = nullptr;
    SSP_TRY(pool_alloc(row * (size_t)im->h, &tmp));
    hipError_t e = hipMemcpy2DAsync(tmp, row, im->data, im->pitch, row, im->h, hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host, tmp, row * (size_t)im->h, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    pool_free(tmp);
    if (e != hipSuccess) SSP_FAIL(SSP_ERR_DEVICE, "download failed: %s", hipGetErrorString(e));
    return 0;
}

SSP_API int ssp_image_info(const ssp_image *im, int *w, int *h, int *cn, int *depth, size_t *pitch, void **ptr)
{
    SSP_REQUIRE(im, "info: null image");
    if (w) *w = im->w;
    if (h) *h = im->h;
    if (cn) *cn = im->cn;
    if (depth) *depth = im->depth;
    if (pitch) *pitch = im->pitch;
    if (ptr) *ptr = im->data;
    return 0;
}
SSP_API int ssp_image_retain(ssp_image *im) { SSP_REQUIRE(im, "retain: null image"); im->refs++; return 0; }
SSP_API int ssp_image_release(ssp_image *im) { image_unref(im); return 0; }
