// Reusable device workspaces for the context-free entry points (SGM front
// end, bilateral upsample, cut_depth_maps).
//
// The reference allocates its cost volumes per SGMStereo object and frees them
// when the object dies (lib/sgm_stereo.cc:192-225); on the device an
// allocation is a driver call that synchronises the whole GPU and costs
// hundreds of microseconds to milliseconds (the 270 MB of volumes of one
// 960 x 540 x 128 run_sgm: ~20 ms to allocate and free, against 6 ms of
// kernels).  A workspace -- its own HIP stream, named device buffers that only
// grow, pinned staging memory -- is therefore checked out of a per-device free
// list for the duration of one call and handed back afterwards; concurrent
// callers (one host thread per reference view, host/view_queue.cc) get
// different workspaces.  Nothing is returned to the driver before
// smvs_release_workspaces() or process exit.
#include "common.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace smvs_hip {

namespace {
std::vector<int> const &
device_map(void)
{
    static std::vector<int> const map = [] {
        std::vector<int> m;
        const char *e = std::getenv("SMVS_DEVICE_MAP");
        if (e == nullptr || e[0] == 0)
            return m;
        int physical = 0;
        if (hipGetDeviceCount(&physical) != hipSuccess)
            physical = 0;
        for (const char *c = e; *c != 0;) {
            char *end = nullptr;
            long const v = std::strtol(c, &end, 10);
            if (end == c)
                break;
            if (v < 0 || v >= physical) {
                std::fprintf(stderr, "[smvs_hip] SMVS_DEVICE_MAP=%s names device %ld, "
                    "%d visible: the map is ignored\n", e, v, physical);
                m.clear();
                return m;
            }
            m.push_back((int)v);
            c = *end == ',' ? end + 1 : end;
            if (*end != ',' && *end != 0)
                break;
        }
        return m;
    }();
    return map;
}
}

int
logical_device_count(void)
{
    std::vector<int> const &m = device_map();
    if (!m.empty())
        return (int)m.size();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return -1;
    return n;
}

int
physical_device(int logical)
{
    std::vector<int> const &m = device_map();
    if (m.empty() || logical < 0 || logical >= (int)m.size())
        return logical;
    return m[(size_t)logical];
}

int
allow_dynamic_lds(int device, const void *kernel, size_t bytes)
{
    // what a launch gets without asking
    if (bytes <= (size_t)64 * 1024)
        return SMVS_OK;
    static std::mutex mutex;
    static std::map<std::pair<int, const void *>, size_t> granted;
    std::lock_guard<std::mutex> guard(mutex);
    size_t &have = granted[std::make_pair(physical_device(device), kernel)];
    if (bytes <= have)
        return SMVS_OK;
    // (the attribute belongs to the current device: the callers have set it)
    hipError_t const err = hipFuncSetAttribute(kernel,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        set_error("allow_dynamic_lds: %zu bytes of LDS refused on device %d: %s", bytes,
            device, hipGetErrorString(err));
        return SMVS_ERR_HIP;
    }
    have = bytes;
    return SMVS_OK;
}

namespace {
constexpr int MAX_DEVICES = 16;
std::mutex g_pool_mutex;
std::vector<Workspace *> g_free[MAX_DEVICES];
int g_created[MAX_DEVICES] = { 0 };
}

static int release_idle_workspaces(void);

int
device_malloc(void **ptr, size_t bytes)
{
    hipError_t err = hipMalloc(ptr, bytes);
    if (err == hipErrorOutOfMemory) {
        // the memory may be ours: parked contexts, idle workspaces
        (void)hipGetLastError();
        if (release_idle_device_memory() > 0)
            err = hipMalloc(ptr, bytes);
    }
    if (err != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        set_error("hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(err));
        return SMVS_ERR_NOMEM;
    }
    return SMVS_OK;
}

int
release_idle_device_memory(void)
{
    return release_idle_workspaces() + ctx_pool_release();
}

// ---- page-locked host buffers (smvs_pinned_alloc), pooled by exact size ----
// hipHostMalloc of the 6 MB of a 1920x1080 RGB image costs more than the
// staging copy it saves; a view's buffers come in a handful of sizes (its
// images, its maps), so freed buffers wait in per-size free lists for the next
// view.  The pool keeps at most PINNED_POOL_BYTES idle.
namespace {
std::mutex g_pinned_mutex;
std::vector<std::pair<size_t, void *>> g_pinned_free;    // (bytes, ptr), idle
std::vector<std::pair<void *, size_t>> g_pinned_live;    // handed out
size_t g_pinned_idle_bytes = 0;
constexpr size_t PINNED_POOL_BYTES = (size_t)4 << 30;
}

int
pinned_pool_release(void)
{
    std::vector<std::pair<size_t, void *>> all;
    {
        std::lock_guard<std::mutex> guard(g_pinned_mutex);
        all.swap(g_pinned_free);
        g_pinned_idle_bytes = 0;
    }
    for (auto &e : all)
        (void)hipHostFree(e.second);
    return (int)all.size();
}

int
Workspace::ensure(int slot, size_t bytes, void **out)
{
    Buf &b = dev[slot];
    if (b.p == nullptr || b.cap < bytes) {
        if (b.p != nullptr) {
            // (work that still uses the old buffer is ordered on this stream)
            SMVS_HIP_CHECK(hipStreamSynchronize(stream));
            (void)hipFree(b.p);
            b.p = nullptr;
            b.cap = 0;
        }
        size_t const want = bytes ? bytes : 1;
        int const rc = device_malloc(&b.p, want);
        if (rc != SMVS_OK) {
            b.p = nullptr;
            return rc;
        }
        b.cap = want;
    }
    *out = b.p;
    return SMVS_OK;
}

int
Workspace::ensure_pinned(size_t bytes)
{
    if (pinned != nullptr && pinned_cap >= bytes)
        return SMVS_OK;
    if (pinned != nullptr) {
        SMVS_HIP_CHECK(hipStreamSynchronize(stream));
        (void)hipHostFree(pinned);
        pinned = nullptr;
        pinned_cap = 0;
    }
    hipError_t const e = hipHostMalloc(&pinned, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
        pinned = nullptr;
        return SMVS_ERR_NOMEM;
    }
    pinned_cap = bytes;
    pinned_used = 0;
    return SMVS_OK;
}

// Host -> device through the pinned staging area: the caller's (pageable)
// buffer is free again when the call returns, the transfer itself is
// asynchronous on the workspace's stream.
int
Workspace::upload(void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0)
        return SMVS_OK;
    size_t const aligned = (bytes + 255) & ~(size_t)255;
    if (pinned == nullptr || pinned_cap < aligned) {
        // grow geometrically: a view's uploads are a handful of images
        size_t want = pinned_cap ? pinned_cap : (size_t)1 << 22;
        while (want < aligned)
            want *= 2;
        int const rc = ensure_pinned(want);
        if (rc != SMVS_OK)
            return rc;
    }
    if (pinned_used + aligned > pinned_cap) {
        // the staging area is full of transfers in flight: wait for them
        SMVS_HIP_CHECK(hipStreamSynchronize(stream));
        pinned_used = 0;
    }
    char *stage = static_cast<char *>(pinned) + pinned_used;
    memcpy(stage, src_host, bytes);
    pinned_used += aligned;
    SMVS_HIP_CHECK(hipMemcpyAsync(dst_dev, stage, bytes, hipMemcpyHostToDevice,
        stream));
    return SMVS_OK;
}

// Device -> host through the staging area; returns when dst_host is filled.
int
Workspace::download(void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0)
        return SMVS_OK;
    // (everything uploaded so far has to be out of the staging area)
    SMVS_HIP_CHECK(hipStreamSynchronize(stream));
    pinned_used = 0;
    if (pinned == nullptr || pinned_cap < bytes) {
        int const rc = ensure_pinned(bytes);
        if (rc != SMVS_OK)
            return rc;
    }
    SMVS_HIP_CHECK(hipMemcpyAsync(pinned, src_dev, bytes, hipMemcpyDeviceToHost,
        stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(stream));
    memcpy(dst_host, pinned, bytes);
    return SMVS_OK;
}

Workspace *
workspace_acquire(int device)
{
    int const count = logical_device_count();
    if (device < 0 || device >= count || device >= MAX_DEVICES) {
        set_error("workspace_acquire: no such HIP device (%d)", device);
        return nullptr;
    }
    if (set_device(device) != hipSuccess) {
        set_error("set_device(%d) failed", device);
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> guard(g_pool_mutex);
        if (!g_free[device].empty()) {
            Workspace *w = g_free[device].back();
            g_free[device].pop_back();
            w->pinned_used = 0;
            return w;
        }
        g_created[device] += 1;
    }
    Workspace *w = new Workspace();
    w->device = device;
    hipError_t const e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(e));
        delete w;
        return nullptr;
    }
    return w;
}

void
workspace_release(Workspace *w)
{
    if (w == nullptr)
        return;
    std::lock_guard<std::mutex> guard(g_pool_mutex);
    g_free[w->device].push_back(w);
}

static void
workspace_destroy(Workspace *w)
{
    (void)set_device(w->device);
    if (w->stream != nullptr)
        (void)hipStreamSynchronize(w->stream);
    for (auto &b : w->dev)
        if (b.p != nullptr)
            (void)hipFree(b.p);
    if (w->pinned != nullptr)
        (void)hipHostFree(w->pinned);
    if (w->stream != nullptr)
        (void)hipStreamDestroy(w->stream);
    delete w;
}

// The workspaces nobody has checked out (a checked-out one is in use by its
// caller and stays).
static int
release_idle_workspaces(void)
{
    int current = 0;
    bool const have_device = hipGetDevice(&current) == hipSuccess;
    std::vector<Workspace *> all;
    {
        std::lock_guard<std::mutex> guard(g_pool_mutex);
        for (auto &list : g_free) {
            all.insert(all.end(), list.begin(), list.end());
            list.clear();
        }
    }
    for (Workspace *w : all)
        workspace_destroy(w);
    if (have_device && !all.empty())
        (void)hipSetDevice(current);
    return (int)all.size();
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_release_workspaces(void)
{
    return release_idle_device_memory() + pinned_pool_release();
}

extern "C" int
smvs_pinned_alloc(size_t bytes, void **out)
{
    SMVS_REQUIRE(out != nullptr && bytes > 0, "bad argument");
    {
        std::lock_guard<std::mutex> guard(g_pinned_mutex);
        for (size_t i = 0; i < g_pinned_free.size(); ++i)
            if (g_pinned_free[i].first == bytes) {
                *out = g_pinned_free[i].second;
                g_pinned_free.erase(g_pinned_free.begin() + (long)i);
                g_pinned_idle_bytes -= bytes;
                g_pinned_live.push_back({ *out, bytes });
                return SMVS_OK;
            }
    }
    void *p = nullptr;
    // (portable: one process may drive several devices -- ViewQueue(N, ...) --
    // and a pooled buffer outlives the view that allocated it)
    hipError_t const e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
        *out = nullptr;
        return e == hipErrorOutOfMemory ? SMVS_ERR_NOMEM : SMVS_ERR_HIP;
    }
    std::lock_guard<std::mutex> guard(g_pinned_mutex);
    g_pinned_live.push_back({ p, bytes });
    *out = p;
    return SMVS_OK;
}

extern "C" int
smvs_pinned_free(void *ptr)
{
    if (ptr == nullptr)
        return SMVS_OK;
    size_t bytes = 0;
    bool keep = false;
    {
        std::lock_guard<std::mutex> guard(g_pinned_mutex);
        for (size_t i = 0; i < g_pinned_live.size(); ++i)
            if (g_pinned_live[i].first == ptr) {
                bytes = g_pinned_live[i].second;
                g_pinned_live.erase(g_pinned_live.begin() + (long)i);
                break;
            }
        SMVS_REQUIRE(bytes != 0, "pointer was not handed out by smvs_pinned_alloc");
        if (g_pinned_idle_bytes + bytes <= PINNED_POOL_BYTES) {
            g_pinned_free.push_back({ bytes, ptr });
            g_pinned_idle_bytes += bytes;
            keep = true;
        }
    }
    if (!keep)
        (void)hipHostFree(ptr);
    return SMVS_OK;
}
