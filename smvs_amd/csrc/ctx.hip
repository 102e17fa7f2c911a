// Context management, plane / surface uploads, profiling.
#include "common.h"

#include <cstdlib>

#include <cmath>

namespace smvs_hip {

PatchLayout
patch_layout(const smvs_ctx *ctx)
{
    static bool const aos = [] {
        const char *e = std::getenv("SMVS_HP_LAYOUT");
        return e != nullptr && e[0] == 'a';
    }();
    PatchLayout L;
    if (aos) {
        L.hq = 1; L.hp = PATCH_H_STRIDE / 4;
        L.gq = 1; L.gp = 4;
    } else {
        unsigned const plane = ((unsigned)ctx->num_patches + 3u) & ~3u;
        L.hq = plane; L.hp = 1;
        L.gq = plane; L.gp = 1;
    }
    return L;
}

static thread_local char g_error[512] = "";

void
set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

ScopedKernelTimer::ScopedKernelTimer(smvs_ctx *c, int k) : ctx(c), cls(k)
{
    if (!ctx->prof.enabled)
        return;
    auto take = [&]() {
        hipEvent_t e = nullptr;
        if (!ctx->prof.pool.empty()) {
            e = ctx->prof.pool.back();
            ctx->prof.pool.pop_back();
        } else {
            (void)hipEventCreate(&e);
        }
        return e;
    };
    a = take();
    b = take();
    (void)hipEventRecord(a, ctx->stream);
}

ScopedKernelTimer::~ScopedKernelTimer()
{
    if (a == nullptr)
        return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof.pending.push_back({cls, a, b});
    if (ctx->prof.pending.size() > 4096)
        (void)profile_collect(ctx);
}

int
profile_collect(smvs_ctx *ctx)
{
    if (ctx->prof.pending.empty())
        return SMVS_OK;
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->prof.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof.ms[p.cls] += ms;
            ctx->prof.launches[p.cls] += 1;
        }
        ctx->prof.pool.push_back(p.a);
        ctx->prof.pool.push_back(p.b);
    }
    ctx->prof.pending.clear();
    return SMVS_OK;
}

__global__ void
expand_hessian_kernel(const float *__restrict__ src, float4 *__restrict__ dst,
    size_t count)
{
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    dst[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.f);
}

static int
ensure_pinned(smvs_ctx *ctx, size_t bytes)
{
    if (ctx->pin != nullptr && ctx->pin_cap >= bytes)
        return SMVS_OK;
    if (ctx->pin != nullptr) {
        (void)hipHostFree(ctx->pin);
        ctx->pin = nullptr;
        ctx->pin_cap = 0;
    }
    size_t const want = bytes < ((size_t)1 << 20) ? (size_t)1 << 20 : bytes;
    hipError_t const e = hipHostMalloc(&ctx->pin, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
        ctx->pin = nullptr;
        return SMVS_ERR_NOMEM;
    }
    ctx->pin_cap = want;
    return SMVS_OK;
}

bool
host_pointer_is_pinned(const void *p)
{
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    hipError_t const e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // (a pageable pointer is not an error here)
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

int
ctx_upload(smvs_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0)
        return SMVS_OK;
    if (bytes >= ((size_t)1 << 20) && host_pointer_is_pinned(src_host)) {
        // page-locked source (smvs_pinned_alloc): one DMA, no staging copy
        SMVS_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice,
            ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return SMVS_OK;
    }
    int const rc = ensure_pinned(ctx, bytes);
    if (rc != SMVS_OK)
        return rc;
    memcpy(ctx->pin, src_host, bytes);
    SMVS_HIP_CHECK(hipMemcpyAsync(dst_dev, ctx->pin, bytes, hipMemcpyHostToDevice,
        ctx->stream));
    // (one staging buffer: the next transfer may overwrite it)
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

int
ctx_download(smvs_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0)
        return SMVS_OK;
    if (bytes >= ((size_t)1 << 20) && host_pointer_is_pinned(dst_host)) {
        SMVS_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost,
            ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return SMVS_OK;
    }
    int const rc = ensure_pinned(ctx, bytes);
    if (rc != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->pin, src_dev, bytes, hipMemcpyDeviceToHost,
        ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(dst_host, ctx->pin, bytes);
    return SMVS_OK;
}

static int
ensure_stage(smvs_ctx *ctx, size_t floats)
{
    if (ctx->stage_cap >= floats)
        return SMVS_OK;
    int rc = device_alloc(&ctx->stage, floats);
    if (rc != SMVS_OK)
        return rc;
    ctx->stage_cap = floats;
    return SMVS_OK;
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" const char *
smvs_last_error(void)
{
    return g_error;
}

extern "C" int
smvs_device_count(void)
{
    // (logical devices: SMVS_DEVICE_MAP, common.h)
    int const n = logical_device_count();
    return n < 0 ? 0 : n;
}

// A context is a few hundred MB of device buffers (the per-patch systems, H,
// the neighbours' planes) behind ~40 allocations.  The reference creates its
// per-view state inside the view's task and drops it afterwards
// (app/smvsrecon.cc:662-732); doing the same with device memory costs every
// view tens of milliseconds, and every hipFree waits for the whole device --
// for the views in flight on the other streams too.  smvs_ctx_destroy
// therefore parks the context, smvs_ctx_create takes a parked context of the
// same geometry and resets its state; smvs_release_workspaces() really frees.
namespace {
std::mutex g_ctx_pool_mutex;
std::vector<smvs_ctx *> g_ctx_pool;
constexpr size_t CTX_POOL_MAX = 32;

void
ctx_reset_for_reuse(smvs_ctx *ctx)
{
    ctx->image_ok = ctx->planes_ok = 0;
    ctx->image_pending = 0;
    ctx->image_direct = 0;
    ctx->upload_stage_busy = 0;   // (the caller has synchronised the stream)
    ctx->sgm_pin_busy = false;
    ctx->sgm_resident = false;
    ctx->surf_depth_ok = false;
    ctx->has_cameras = ctx->has_surface = ctx->has_system = false;
    ctx->has_shading = false;
    ctx->update_prepared = false;
    ctx->cg_use_active = false;
    ctx->last_cg_iterations = 0;
    ctx->last_loop_steps = 1 << 30;
    ctx->nodes_saved_count = 0;
    ctx->solver_mode = 0;
    ctx->resident_disabled = false;   // (a new view tries the resident solver again)
    ctx->num_nodes = ctx->num_patches = 0;
    (void)profile_collect(ctx);
    ctx->prof.enabled = false;
    for (int i = 0; i < SMVS_K_COUNT; ++i) {
        ctx->prof.ms[i] = 0.0;
        ctx->prof.launches[i] = 0;
    }
}
}

static int ctx_free(smvs_ctx *ctx);

extern "C" int
smvs_ctx_create(int device, int width, int height, int n_subs, smvs_ctx **out)
{
    if (out != nullptr && width > 4 && height > 4 && n_subs >= 1
        && n_subs <= SMVS_MAX_SUBS) {
        std::lock_guard<std::mutex> guard(g_ctx_pool_mutex);
        for (size_t i = 0; i < g_ctx_pool.size(); ++i) {
            smvs_ctx *c = g_ctx_pool[i];
            if (c->device == device && c->width == width && c->height == height
                && c->n_subs == n_subs) {
                g_ctx_pool.erase(g_ctx_pool.begin() + (long)i);
                *out = c;
                return SMVS_OK;
            }
        }
    }
    SMVS_REQUIRE(out != nullptr, "out must not be null");
    SMVS_REQUIRE(width > 4 && height > 4, "image too small");
    SMVS_REQUIRE(n_subs >= 1 && n_subs <= SMVS_MAX_SUBS,
        "n_subs must be in [1, SMVS_MAX_SUBS]");
    int const count = logical_device_count();
    SMVS_REQUIRE(device >= 0 && device < count, "no such HIP device");
    SMVS_HIP_CHECK(set_device(device));

    smvs_ctx *ctx = new smvs_ctx();
    ctx->device = device;
    ctx->width = width;
    ctx->height = height;
    ctx->n_subs = n_subs;
    hipError_t err = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (err != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(err));
        delete ctx;
        return SMVS_ERR_HIP;
    }
    size_t const npix = (size_t)width * height;
    int rc = SMVS_OK;
    if ((rc = device_alloc(&ctx->main_grad, npix)) != SMVS_OK
        || (rc = device_alloc(&ctx->cams, 1)) != SMVS_OK
        || (rc = device_alloc(&ctx->subs_dev, SMVS_MAX_SUBS)) != SMVS_OK
        || (rc = device_alloc(&ctx->scalars, S_NUM)) != SMVS_OK
        || (rc = device_alloc(&ctx->status, I_NUM)) != SMVS_OK
        || (rc = device_alloc(&ctx->lighting, 16)) != SMVS_OK
        || (rc = device_alloc(&ctx->lightAb, 272)) != SMVS_OK
        || (rc = device_alloc(&ctx->partials, 2 * 8 * 512)) != SMVS_OK
        || (rc = device_alloc(&ctx->step_counter, 2)) != SMVS_OK
        || (rc = device_alloc(&ctx->zero_block, 16)) != SMVS_OK
        || (rc = device_alloc(reinterpret_cast<char **>(&ctx->cg_state), 256)) != SMVS_OK) {
        ctx_free(ctx);
        return rc;
    }
    // coherent (fine-grained) pinned memory: the CG kernels publish their
    // progress with system-scope stores that the host polls while they run
    unsigned const host_flags = hipHostMallocCoherent | hipHostMallocMapped;
    if (hipHostMalloc((void **)&ctx->cg_progress, sizeof(int) * 8, host_flags) != hipSuccess
        || hipHostMalloc((void **)&ctx->step_words, sizeof(int) * STEP_SLOTS * STEP_SLOT_INTS, host_flags) != hipSuccess
        || hipHostMalloc((void **)&ctx->status_host, sizeof(int) * I_NUM, host_flags) != hipSuccess
        || hipHostMalloc((void **)&ctx->scalars_host, sizeof(double) * S_NUM, host_flags) != hipSuccess) {
        set_error("hipHostMalloc failed");
        ctx_free(ctx);
        return SMVS_ERR_NOMEM;
    }
    (void)hipMemsetAsync(ctx->scalars, 0, sizeof(double) * S_NUM, ctx->stream);
    (void)hipMemsetAsync(ctx->status, 0, sizeof(int) * I_NUM, ctx->stream);
    (void)hipMemsetAsync(ctx->step_counter, 0, 2 * sizeof(unsigned long long), ctx->stream);
    (void)hipMemsetAsync(ctx->zero_block, 0, 16 * sizeof(double), ctx->stream);
    for (int i = 0; i < 8; ++i)
        ctx->cg_progress[i] = 0;
    for (int i = 0; i < STEP_SLOTS * STEP_SLOT_INTS; ++i)
        ctx->step_words[i] = 0;
    *out = ctx;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_destroy(smvs_ctx *ctx)
{
    if (ctx == nullptr)
        return SMVS_OK;
    // park it for the next view of this geometry
    (void)set_device(ctx->device);
    // (a DMA of smvs_ctx_upload_image_async nobody waited for reads the caller's
    // buffer: it must have ended when this returns)
    if (ctx->copy_stream != nullptr)
        (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->stream != nullptr && hipStreamSynchronize(ctx->stream) == hipSuccess) {
        ctx_reset_for_reuse(ctx);
        // The pool is bounded by count.  What it holds goes back to the driver
        // when an allocation of this library runs out of memory (device_malloc
        // releases the parked contexts and the idle workspaces and tries again)
        // -- not by looking at the device's free memory: with several processes
        // on one GPU, or this process's own live contexts filling it, that rule
        // (round 4) drained the pool on every destroy and every view paid its
        // ~40 allocations again.
        std::vector<smvs_ctx *> evict;
        {
            std::lock_guard<std::mutex> guard(g_ctx_pool_mutex);
            g_ctx_pool.push_back(ctx);
            while (g_ctx_pool.size() > CTX_POOL_MAX) {
                evict.push_back(g_ctx_pool.front());
                g_ctx_pool.erase(g_ctx_pool.begin());
            }
        }
        for (smvs_ctx *c : evict)
            (void)ctx_free(c);
        return SMVS_OK;
    }
    return ctx_free(ctx);
}

int
smvs_hip::ctx_pool_release(void)
{
    int current = 0;
    bool const have_device = hipGetDevice(&current) == hipSuccess;
    std::vector<smvs_ctx *> all;
    {
        std::lock_guard<std::mutex> guard(g_ctx_pool_mutex);
        all.swap(g_ctx_pool);
    }
    for (smvs_ctx *c : all)
        (void)ctx_free(c);
    // (ctx_free selects the context's device; the caller may be in the middle
    // of an allocation on another one)
    if (have_device && !all.empty())
        (void)hipSetDevice(current);
    return (int)all.size();
}

static int
ctx_free(smvs_ctx *ctx)
{
    if (ctx == nullptr)
        return SMVS_OK;
    (void)set_device(ctx->device);
    if (ctx->copy_stream)
        (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i <= SMVS_MAX_SUBS; ++i) {
        if (ctx->upload_stage[i])
            (void)hipFree(ctx->upload_stage[i]);
        if (ctx->image_ready[i])
            (void)hipEventDestroy(ctx->image_ready[i]);
    }
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->sgm_pin)
        (void)hipHostFree(ctx->sgm_pin);
    void *bufs[] = { ctx->main_grad, ctx->main_shading, ctx->main_shading_grad,
        ctx->cams, ctx->subs_dev, ctx->nodes, ctx->node_valid, ctx->patch_valid,
        ctx->patch_vis, ctx->active, ctx->active_next, ctx->cg_mask, ctx->hermite_all,
        ctx->Hp, ctx->gp, ctx->H9, ctx->Pinv, ctx->g, ctx->lighting, ctx->x,
        ctx->r, ctx->z, ctx->Ad, ctx->d, ctx->d2, ctx->b, ctx->partials,
        ctx->cg_state, ctx->scalars,
        ctx->status, ctx->lightAb, ctx->stage, ctx->map_scratch,
        ctx->light_partial, ctx->res_work, ctx->res_zx, ctx->live_list,
        ctx->step_counter, ctx->nodes_saved, ctx->zero_block, ctx->byte_stage,
        ctx->surf_depth, ctx->surf_tmp, ctx->surf_tmp_bytes, ctx->surf_bits };
    for (void *p : bufs)
        if (p)
            (void)hipFree(p);
    for (int i = 0; i <= SMVS_MAX_SUBS; ++i)
        if (ctx->images[i].data)
            (void)hipFree(ctx->images[i].data);
    for (int i = 0; i < SMVS_MAX_SUBS; ++i)
        if (ctx->topo_zbuf[i])
            (void)hipFree(ctx->topo_zbuf[i]);
    if (ctx->topo_sgm)
        (void)hipFree(ctx->topo_sgm);
    if (ctx->sgm_lowres)
        (void)hipFree(ctx->sgm_lowres);
    if (ctx->bil_lut)
        (void)hipFree(ctx->bil_lut);
    if (ctx->bil_tri)
        (void)hipFree(ctx->bil_tri);
    if (ctx->topo_ncc)
        (void)hipFree(ctx->topo_ncc);
    if (ctx->topo_mse)
        (void)hipFree(ctx->topo_mse);
    if (ctx->topo_border)
        (void)hipFree(ctx->topo_border);
    if (ctx->topo_mse_list)
        (void)hipFree(ctx->topo_mse_list);
    if (ctx->topo_mse_parts)
        (void)hipFree(ctx->topo_mse_parts);
    if (ctx->topo_mse_arrived)
        (void)hipFree(ctx->topo_mse_arrived);
    if (ctx->topo_pix)
        (void)hipFree(ctx->topo_pix);
    if (ctx->topo_pair_alive)
        (void)hipFree(ctx->topo_pair_alive);
    if (ctx->blur_tmp[0])
        (void)hipFree(ctx->blur_tmp[0]);
    if (ctx->blur_tmp[1])
        (void)hipFree(ctx->blur_tmp[1]);
    for (int i = 0; i < SMVS_MAX_SUBS; ++i) {
        if (ctx->subs[i].grad)
            (void)hipFree(ctx->subs[i].grad);
        if (ctx->subs[i].hess)
            (void)hipFree(ctx->subs[i].hess);
    }
    if (ctx->pin)
        (void)hipHostFree(ctx->pin);
    if (ctx->status_host)
        (void)hipHostFree(ctx->status_host);
    if (ctx->cg_progress)
        (void)hipHostFree(ctx->cg_progress);
    if (ctx->step_words)
        (void)hipHostFree(ctx->step_words);
    if (ctx->scalars_host)
        (void)hipHostFree(ctx->scalars_host);
    for (auto &p : ctx->prof.pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : ctx->prof.pool)
        (void)hipEventDestroy(e);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_synchronize(smvs_ctx *ctx)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    SMVS_HIP_CHECK(set_device(ctx->device));
    if (ctx->copy_stream != nullptr)
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->copy_stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->upload_stage_busy = 0;
    ctx->sgm_pin_busy = false;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_set_solver(smvs_ctx *ctx, int mode)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    SMVS_REQUIRE(mode == SMVS_SOLVER_AUTO || mode == SMVS_SOLVER_STREAMING
        || mode == SMVS_SOLVER_RESIDENT_REF, "unknown solver mode");
    ctx->solver_mode = mode;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_set_cameras(smvs_ctx *ctx, const double *Mi, const double *ti,
    float flen, float inv_flen)
{
    SMVS_REQUIRE(ctx && Mi && ti, "null argument");
    SMVS_HIP_CHECK(set_device(ctx->device));
    DeviceCameras cams;
    memset(&cams, 0, sizeof(cams));
    for (int s = 0; s < ctx->n_subs; ++s) {
        memcpy(cams.M[s], Mi + 9 * s, sizeof(double) * 9);
        memcpy(cams.t[s], ti + 3 * s, sizeof(double) * 3);
        const double *M = cams.M[s];
        const double *t = cams.t[s];
        for (int k = 0; k < 3; ++k) {
            cams.shift[s][k] = M[k] * t[2] - M[6 + k] * t[0];
            cams.shift[s][3 + k] = M[3 + k] * t[2] - M[6 + k] * t[1];
        }
    }
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->cams, &cams, sizeof(cams),
        hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->flen = flen;
    ctx->inv_flen = inv_flen;
    ctx->has_cameras = true;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_upload_main(smvs_ctx *ctx, const float *grad2, const float *shading1,
    const float *shading_grad2)
{
    SMVS_REQUIRE(ctx && grad2, "null argument");
    SMVS_REQUIRE((shading1 == nullptr) == (shading_grad2 == nullptr),
        "shading image and shading gradients come together");
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const npix = (size_t)ctx->width * ctx->height;
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->main_grad, grad2, npix * sizeof(float2),
        hipMemcpyHostToDevice, ctx->stream));
    if (shading1 != nullptr) {
        int rc;
        if (ctx->main_shading == nullptr) {
            if ((rc = device_alloc(&ctx->main_shading, npix)) != SMVS_OK)
                return rc;
            if ((rc = device_alloc(&ctx->main_shading_grad, npix)) != SMVS_OK)
                return rc;
        }
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->main_shading, shading1,
            npix * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->main_shading_grad, shading_grad2,
            npix * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
        ctx->has_shading = true;
    }
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_ctx_upload_sub(smvs_ctx *ctx, int sub, int width, int height,
    const float *grad2, const float *hess3)
{
    SMVS_REQUIRE(ctx && grad2 && hess3, "null argument");
    SMVS_REQUIRE(sub >= 0 && sub < ctx->n_subs, "sub view index out of range");
    SMVS_REQUIRE(width > 1 && height > 1, "bad sub view size");
    SMVS_HIP_CHECK(set_device(ctx->device));
    SubPlanes &sp = ctx->subs[sub];
    size_t const npix = (size_t)width * height;
    int rc;
    if (sp.width != width || sp.height != height || sp.grad == nullptr
        || sp.hess == nullptr) {
        // a failed allocation leaves the planes marked absent, so that the
        // next call allocates both again
        sp.width = sp.height = 0;
        if ((rc = device_alloc(&sp.grad, npix)) != SMVS_OK
            || (rc = device_alloc(&sp.hess, npix)) != SMVS_OK) {
            (void)device_alloc(&sp.grad, 0);
            (void)device_alloc(&sp.hess, 0);
            return rc;
        }
        sp.width = width;
        sp.height = height;
    }
    if ((rc = ensure_stage(ctx, npix * 3)) != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemcpyAsync(sp.grad, grad2, npix * sizeof(float2),
        hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->stage, hess3, npix * 3 * sizeof(float),
        hipMemcpyHostToDevice, ctx->stream));
    unsigned const blocks = (unsigned)((npix + 255) / 256);
    hipLaunchKernelGGL(expand_hessian_kernel, dim3(blocks), dim3(256), 0,
        ctx->stream, ctx->stage, sp.hess, npix);
    SMVS_HIP_CHECK(hipGetLastError());
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->subs_dev, ctx->subs,
        sizeof(SubPlanes) * SMVS_MAX_SUBS, hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->planes_ok |= 1u << sub;
    return SMVS_OK;
}

// 1-D cubic Hermite basis (value at node 0, value at node 1, slope at node 0,
// slope at node 1) with derivatives 0..2 already divided by patchsize^k:
// the bicubic patch basis (reference: bicubic_patch.cc:20-38, 258-316 and
// surface.cc:929-955) is the tensor product of two rows of this table.
static void
build_hermite_table(int ps, std::vector<double> *tab)
{
    tab->assign((size_t)ps * 12, 0.0);
    double const s1 = 1.0 / ps;
    double const s2 = s1 / ps;
    for (int c = 0; c < ps; ++c) {
        double const t = ((double)c + 0.5) / ps;
        double const t2 = t * t, t3 = t2 * t;
        double *row = tab->data() + (size_t)c * 12;
        // h00 = 1 - 3t^2 + 2t^3
        row[0] = 1.0 - 3.0 * t2 + 2.0 * t3;
        row[1] = (-6.0 * t + 6.0 * t2) * s1;
        row[2] = (-6.0 + 12.0 * t) * s2;
        // h01 = 3t^2 - 2t^3
        row[3] = 3.0 * t2 - 2.0 * t3;
        row[4] = (6.0 * t - 6.0 * t2) * s1;
        row[5] = (6.0 - 12.0 * t) * s2;
        // h10 = t - 2t^2 + t^3
        row[6] = t - 2.0 * t2 + t3;
        row[7] = (1.0 - 4.0 * t + 3.0 * t2) * s1;
        row[8] = (-4.0 + 6.0 * t) * s2;
        // h11 = -t^2 + t^3
        row[9] = -t2 + t3;
        row[10] = (-2.0 * t + 3.0 * t2) * s1;
        row[11] = (-2.0 + 6.0 * t) * s2;
    }
}

__global__ void
init_active_kernel(const uint8_t *__restrict__ node_valid,
    uint8_t *__restrict__ active, int n)
{
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        active[i] = node_valid[i] ? 1 : 0;
}

int
smvs_hip::ctx_ensure_grid(smvs_ctx *ctx, int scale, int npx, int npy, int start_x,
    int start_y)
{
    SMVS_REQUIRE(scale >= 0 && scale <= 10, "scale out of range");
    SMVS_REQUIRE(npx >= 1 && npy >= 1, "empty patch grid");
    int const ps = 1 << scale;
    SMVS_REQUIRE(start_x >= 0 && start_y >= 0
        && start_x + npx * ps <= ctx->width && start_y + npy * ps <= ctx->height,
        "patch grid does not fit the image");
    SMVS_HIP_CHECK(set_device(ctx->device));

    size_t const N = (size_t)(npx + 1) * (npy + 1);
    size_t const P = (size_t)npx * npy;
    int rc = SMVS_OK;
    if (N > ctx->cap_nodes) {
        size_t const cap = N + N / 8;
        // (a failure below leaves the context without a surface and with a
        // zero capacity: the next call allocates everything again)
        ctx->cap_nodes = 0;
        ctx->has_surface = false;
        ctx->has_system = false;
        if ((rc = device_alloc(&ctx->nodes, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->node_valid, cap)) != SMVS_OK
            || (rc = device_alloc(&ctx->active, cap)) != SMVS_OK
            || (rc = device_alloc(&ctx->active_next, cap)) != SMVS_OK
            || (rc = device_alloc(&ctx->cg_mask, cap)) != SMVS_OK
            || (rc = device_alloc(&ctx->H9, cap * 5 * 16)) != SMVS_OK
            || (rc = device_alloc(&ctx->Pinv, cap * 16)) != SMVS_OK
            || (rc = device_alloc(&ctx->g, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->x, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->r, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->z, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->Ad, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->d, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->d2, cap * 4)) != SMVS_OK
            || (rc = device_alloc(&ctx->b, cap * 4)) != SMVS_OK)
            return rc;
        ctx->cap_nodes = cap;
    }
    if (P > ctx->cap_patches) {
        size_t const cap = P + P / 8;
        ctx->cap_patches = 0;
        ctx->has_surface = false;
        ctx->has_system = false;
        if ((rc = device_alloc(&ctx->patch_valid, cap)) != SMVS_OK
            || (rc = device_alloc(&ctx->live_list, cap)) != SMVS_OK
            || (rc = device_alloc(&ctx->patch_vis, cap)) != SMVS_OK
            // (36 and 4 quads of 4 doubles per patch, in planes of the patch
            // count rounded up to a multiple of four: PatchLayout, common.h)
            || (rc = device_alloc(&ctx->Hp, (cap + 4) * PATCH_H_STRIDE)) != SMVS_OK
            || (rc = device_alloc(&ctx->gp, (cap + 4) * 16)) != SMVS_OK)
            return rc;
        ctx->cap_patches = cap;
    }
    ctx->update_prepared = false;
    ctx->scale = scale;
    ctx->patchsize = ps;
    ctx->npx = npx;
    ctx->npy = npy;
    ctx->start_x = start_x;
    ctx->start_y = start_y;
    ctx->num_nodes = (int)N;
    ctx->num_patches = (int)P;
    ctx->node_stride = npx + 1;

    if (ctx->hermite_all == nullptr) {
        // the tables of all patch sizes, once per context: [ps - 1 + row][12]
        // for ps = 1, 2, 4, .. 1024 (a table per scale change was a device
        // allocation, i.e. a device-wide synchronisation, per scale)
        std::vector<double> all, tab;
        for (int s = 0; s <= 10; ++s) {
            build_hermite_table(1 << s, &tab);
            all.insert(all.end(), tab.begin(), tab.end());
        }
        if ((rc = device_alloc(&ctx->hermite_all, all.size())) != SMVS_OK)
            return rc;
        SMVS_HIP_CHECK(hipMemcpy(ctx->hermite_all, all.data(),
            all.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    ctx->hermite_tab = ctx->hermite_all + (size_t)(ps - 1) * 12;
    ctx->hermite_tab_ps = ps;
    ctx->has_system = false;
    ctx->cg_use_active = false;
    // saved nodes belong to the surface they were saved from (same grid size
    // does not mean same surface)
    ctx->nodes_saved_count = 0;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_set_surface(smvs_ctx *ctx, int scale, int npx, int npy, int start_x,
    int start_y, const double *nodes, const uint8_t *node_valid,
    const uint8_t *patch_valid, const uint32_t *patch_vis)
{
    SMVS_REQUIRE(ctx && nodes && node_valid && patch_valid && patch_vis,
        "null argument");
    int rc = ctx_ensure_grid(ctx, scale, npx, npy, start_x, start_y);
    if (rc != SMVS_OK)
        return rc;
    size_t const N = (size_t)ctx->num_nodes;
    size_t const P = (size_t)ctx->num_patches;
    if ((rc = ctx_upload(ctx, ctx->nodes, nodes, N * 4 * sizeof(double))) != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->node_valid, node_valid, N,
        hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->patch_valid, patch_valid, P,
        hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->patch_vis, patch_vis,
        P * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(init_active_kernel, dim3((unsigned)((N + 255) / 256)),
        dim3(256), 0, ctx->stream, ctx->node_valid, ctx->active, (int)N);
    SMVS_HIP_CHECK(hipGetLastError());
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->has_surface = true;
    ctx->has_system = false;
    ctx->cg_use_active = false;
    // saved nodes belong to the surface they were saved from (same grid size
    // does not mean same surface)
    ctx->nodes_saved_count = 0;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_set_active(smvs_ctx *ctx, const uint8_t *active)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_surface) {
        set_error("smvs_ctx_set_active: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    // a system built for another active set no longer matches the flags
    ctx->cg_use_active = false;
    if (active == nullptr) {
        hipLaunchKernelGGL(init_active_kernel,
            dim3((unsigned)((ctx->num_nodes + 255) / 256)), dim3(256), 0,
            ctx->stream, ctx->node_valid, ctx->active, ctx->num_nodes);
        SMVS_HIP_CHECK(hipGetLastError());
    } else {
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->active, active, ctx->num_nodes,
            hipMemcpyHostToDevice, ctx->stream));
    }
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_get_active(smvs_ctx *ctx, uint8_t *active, int *num_active)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_surface) {
        set_error("smvs_get_active: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    std::vector<uint8_t> tmp(ctx->num_nodes);
    SMVS_HIP_CHECK(hipMemcpyAsync(tmp.data(), ctx->active, ctx->num_nodes,
        hipMemcpyDeviceToHost, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (active != nullptr)
        memcpy(active, tmp.data(), tmp.size());
    if (num_active != nullptr) {
        int n = 0;
        for (uint8_t a : tmp)
            n += (a == 1);
        *num_active = n;
    }
    return SMVS_OK;
}

extern "C" int
smvs_get_nodes(smvs_ctx *ctx, double *nodes)
{
    SMVS_REQUIRE(ctx && nodes, "null argument");
    if (!ctx->has_surface) {
        set_error("smvs_get_nodes: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    return ctx_download(ctx, nodes, ctx->nodes,
        (size_t)ctx->num_nodes * 4 * sizeof(double));
}

extern "C" int
smvs_set_nodes(smvs_ctx *ctx, const double *nodes)
{
    SMVS_REQUIRE(ctx && nodes, "null argument");
    if (!ctx->has_surface) {
        set_error("smvs_set_nodes: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->nodes, nodes,
        (size_t)ctx->num_nodes * 4 * sizeof(double), hipMemcpyHostToDevice,
        ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_ctx_save_nodes(smvs_ctx *ctx)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_surface) {
        set_error("smvs_ctx_save_nodes: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const n = (size_t)ctx->num_nodes * 4;
    if (ctx->nodes_saved_cap < n) {
        if (ctx->nodes_saved != nullptr)
            (void)hipFree(ctx->nodes_saved);
        ctx->nodes_saved = nullptr;
        ctx->nodes_saved_cap = 0;
        ctx->nodes_saved_count = 0;
        int const rc = device_alloc(&ctx->nodes_saved, n);
        if (rc != SMVS_OK)
            return rc;
        ctx->nodes_saved_cap = n;
    }
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->nodes_saved, ctx->nodes, n * sizeof(double),
        hipMemcpyDeviceToDevice, ctx->stream));
    ctx->nodes_saved_count = ctx->num_nodes;
    ctx->nodes_saved_stride = ctx->node_stride;
    return SMVS_OK;
}

extern "C" int
smvs_ctx_restore_nodes(smvs_ctx *ctx)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_surface || ctx->nodes_saved == nullptr
        || ctx->nodes_saved_count != ctx->num_nodes
        || ctx->nodes_saved_stride != ctx->node_stride) {
        set_error("smvs_ctx_restore_nodes: no saved nodes for this surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->nodes, ctx->nodes_saved,
        (size_t)ctx->num_nodes * 4 * sizeof(double), hipMemcpyDeviceToDevice,
        ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_ctx_clone_loop_state(smvs_ctx *src, smvs_ctx **out)
{
    SMVS_REQUIRE(src != nullptr && out != nullptr, "null argument");
    if (!src->has_surface || !src->has_cameras) {
        set_error("smvs_ctx_clone_loop_state: the context holds no surface / cameras");
        return SMVS_ERR_STATE;
    }
    smvs_ctx *dst = nullptr;
    int rc = smvs_ctx_create(src->device, src->width, src->height, src->n_subs, &dst);
    if (rc != SMVS_OK)
        return rc;
    auto fail = [&](int code) {
        (void)smvs_ctx_destroy(dst);
        return code;
    };
    if (set_device(src->device) != hipSuccess)
        return fail(SMVS_ERR_HIP);
    // everything below is ordered behind the source's pending work
    if (hipStreamSynchronize(src->stream) != hipSuccess)
        return fail(SMVS_ERR_HIP);
    hipStream_t const st = dst->stream;
    auto copy = [&](void *to, const void *from, size_t bytes) {
        return hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    };
    size_t const npix = (size_t)src->width * src->height;
    bool ok = copy(dst->cams, src->cams, sizeof(DeviceCameras))
        && copy(dst->main_grad, src->main_grad, npix * sizeof(float2))
        && copy(dst->lighting, src->lighting, 16 * sizeof(double));
    dst->flen = src->flen;
    dst->inv_flen = src->inv_flen;
    dst->has_cameras = true;
    dst->solver_mode = src->solver_mode;
    if (ok && src->has_shading) {
        if (dst->main_shading == nullptr
            && ((rc = device_alloc(&dst->main_shading, npix)) != SMVS_OK
                || (rc = device_alloc(&dst->main_shading_grad, npix)) != SMVS_OK))
            return fail(rc);
        ok = copy(dst->main_shading, src->main_shading, npix * sizeof(float))
            && copy(dst->main_shading_grad, src->main_shading_grad, npix * sizeof(float2));
        dst->has_shading = true;
    }
    for (int j = 0; ok && j < src->n_subs; ++j) {
        if (!((src->planes_ok >> j) & 1u))
            continue;
        SubPlanes const &from = src->subs[j];
        SubPlanes &to = dst->subs[j];
        size_t const n = (size_t)from.width * from.height;
        if (to.width != from.width || to.height != from.height || to.grad == nullptr
            || to.hess == nullptr) {
            to.width = to.height = 0;
            if ((rc = device_alloc(&to.grad, n)) != SMVS_OK
                || (rc = device_alloc(&to.hess, n)) != SMVS_OK) {
                (void)device_alloc(&to.grad, 0);
                (void)device_alloc(&to.hess, 0);
                return fail(rc);
            }
            to.width = from.width;
            to.height = from.height;
        }
        ok = copy(to.grad, from.grad, n * sizeof(*from.grad))
            && copy(to.hess, from.hess, n * sizeof(*from.hess));
        dst->planes_ok |= 1u << j;
    }
    if (ok)
        ok = hipMemcpyAsync(dst->subs_dev, dst->subs, sizeof(SubPlanes) * SMVS_MAX_SUBS,
            hipMemcpyHostToDevice, st) == hipSuccess;
    if (!ok)
        return fail(SMVS_ERR_HIP);
    if ((rc = ctx_ensure_grid(dst, src->scale, src->npx, src->npy, src->start_x,
             src->start_y)) != SMVS_OK)
        return fail(rc);
    size_t const N = (size_t)src->num_nodes, P = (size_t)src->num_patches;
    ok = copy(dst->nodes, src->nodes, N * 4 * sizeof(double))
        && copy(dst->node_valid, src->node_valid, N)
        && copy(dst->patch_valid, src->patch_valid, P)
        && copy(dst->patch_vis, src->patch_vis, P * sizeof(uint32_t));
    if (!ok)
        return fail(SMVS_ERR_HIP);
    hipLaunchKernelGGL(init_active_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,
        st, dst->node_valid, dst->active, (int)N);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(SMVS_ERR_HIP);
    dst->has_surface = true;
    if ((rc = smvs_ctx_save_nodes(dst)) != SMVS_OK)
        return fail(rc);
    if (hipStreamSynchronize(st) != hipSuccess)
        return fail(SMVS_ERR_HIP);
    *out = dst;
    return SMVS_OK;
}

extern "C" int
smvs_profile_enable(smvs_ctx *ctx, int on)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!on)
        (void)profile_collect(ctx);
    ctx->prof.enabled = on != 0;
    return SMVS_OK;
}

extern "C" int
smvs_profile_reset(smvs_ctx *ctx)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    (void)profile_collect(ctx);
    for (int i = 0; i < SMVS_K_COUNT; ++i) {
        ctx->prof.ms[i] = 0.0;
        ctx->prof.launches[i] = 0;
    }
    return SMVS_OK;
}

extern "C" int
smvs_profile_get(smvs_ctx *ctx, double *ms, long long *launches)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    int rc = profile_collect(ctx);
    if (rc != SMVS_OK)
        return rc;
    for (int i = 0; i < SMVS_K_COUNT; ++i) {
        if (ms)
            ms[i] = ctx->prof.ms[i];
        if (launches)
            launches[i] = ctx->prof.launches[i];
    }
    return SMVS_OK;
}
