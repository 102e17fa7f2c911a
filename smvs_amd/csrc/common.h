// Internal declarations shared by the HIP translation units of libsmvs_hip.so.
// gfx950 only: no portability shims.
#pragma once

#include "host/topo_math.h"
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/smvs_hip.h"

namespace smvs_hip {

void set_error(const char *fmt, ...);

#define SMVS_HIP_CHECK(expr)                                                  \
    do {                                                                      \
        hipError_t err__ = (expr);                                            \
        if (err__ != hipSuccess) {                                            \
            smvs_hip::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, \
                hipGetErrorString(err__));                                    \
            return SMVS_ERR_HIP;                                              \
        }                                                                     \
    } while (0)

#define SMVS_REQUIRE(cond, msg)                                               \
    do {                                                                      \
        if (!(cond)) {                                                        \
            smvs_hip::set_error("%s: %s", __func__, msg);                     \
            return SMVS_ERR_INVALID;                                          \
        }                                                                     \
    } while (0)

// Number of double scalars kept on the device for the CG / GN loop.
enum {
    S_RR = 0,        // r_dot_r (z.r with preconditioner)
    S_DAD,           // d . A d
    S_RR_NEW,        // r . r after the update
    S_Q0,
    S_Q1,
    S_ZR,            // z . r after the update
    S_BETA,
    S_ALPHA,
    S_TOL,           // error tolerance
    S_GNORM,         // ||g||
    S_SUMDIFF,       // full_optimization: sum of reprojection deltas
    S_COUNT_DIFF,    // number of terms in S_SUMDIFF
    S_NUM = 16
};

// Integer status words on the device.
enum {
    I_DONE = 0,      // CG finished
    I_INFO,          // smvs_cg_info
    I_ITER,          // CG iteration counter (starts at 1)
    I_NAN,           // delta[0] is NaN
    I_NUM_ACTIVE,    // active nodes after re-activation
    I_ACTIVE_PATCHES,
    I_TOPO_DELETED,  // patches deleted by one cut_boundaries pass
    I_TOPO_CANDIDATES,   // (behind I_TOPO_DELETED: one memset clears both) patches
                         // the mse kernel evaluates (topology.hip)
    I_LIVE_PATCHES,  // entries of the compacted live-patch list
    I_NUM_INITIAL,   // active nodes at the start of the Newton loop (finish_step_kernel, begin)
    I_STOP,          // pipelined Newton loop: the loop has ended, enqueued steps do nothing
    I_STEP_ABORT,    // pipelined Newton loop: this step was abandoned (ABORT_*); it and
                     // the steps enqueued behind it change nothing
    I_SURF_VALID,    // valid patches of the device surface (surface.hip)
    I_SURF_CHANGED,  // patches filled / deleted by the last grid operation
    I_TOPO_PASS0 = 16,   // cut_boundaries passes enqueued ahead: {deleted, candidates} of
                         // pass k at I_TOPO_PASS0 + 2 k (topology.hip), TOPO_AHEAD passes
    I_NUM = 24
};
constexpr int TOPO_AHEAD = 4;

struct SubPlanes {
    int width = 0, height = 0;
    float2 *grad = nullptr;   // [h][w]  (I_x, I_y)
    float4 *hess = nullptr;   // [h][w]  (I_xx, I_xy, I_yy, 0)
};

struct DeviceCameras {
    double M[SMVS_MAX_SUBS][9];
    double t[SMVS_MAX_SUBS][3];
    // p t2 - r t0 and q t2 - r t1 as affine functions of the pixel (u, v, 1),
    // with (p, q, r) = M (u, v, 1): the numerators of the reprojection shift
    // (update.hip, reactivate_kernel)
    double shift[SMVS_MAX_SUBS][6];
};

struct Profile {
    bool enabled = false;
    double ms[SMVS_K_COUNT] = {0};
    long long launches[SMVS_K_COUNT] = {0};
    struct Pending { int cls; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

} // namespace smvs_hip

struct smvs_ctx {
    int device = 0;
    int width = 0, height = 0, n_subs = 0;
    hipStream_t stream = nullptr;
    float flen = 0.f, inv_flen = 0.f;
    // which views hold an uploaded image / which neighbours hold planes: the
    // buffers themselves outlive a pooled context's previous use (ctx.hip)
    uint32_t image_ok = 0, planes_ok = 0;
    bool has_cameras = false, has_surface = false, has_system = false;
    bool has_shading = false;
    bool update_prepared = false;   // active_next / counters cleared for the next update

    // image planes
    float2 *main_grad = nullptr;
    float *main_shading = nullptr;
    float2 *main_shading_grad = nullptr;
    smvs_hip::SubPlanes subs[SMVS_MAX_SUBS];
    smvs_hip::DeviceCameras *cams = nullptr;  // device copy
    smvs_hip::SubPlanes *subs_dev = nullptr;  // device copy of subs[]

    // surface
    int scale = 0, patchsize = 0, npx = 0, npy = 0, start_x = 0, start_y = 0;
    int num_nodes = 0, num_patches = 0, node_stride = 0;
    size_t cap_nodes = 0, cap_patches = 0;
    double *nodes = nullptr;        // [N][4]
    uint8_t *node_valid = nullptr;
    uint8_t *patch_valid = nullptr;
    uint32_t *patch_vis = nullptr;
    uint8_t *active = nullptr, *active_next = nullptr;
    int *live_list = nullptr;       // [P] patches with an active node, compacted
    uint16_t *cg_mask = nullptr;    // per node: stencil slots present in the CG matrix
    double *hermite_tab = nullptr;  // [ps][12] 1-D Hermite basis table (inside hermite_all)
    double *hermite_all = nullptr;  // the tables of ps = 1 .. 1024, [ps - 1 + row][12]
    int hermite_tab_ps = 0;

    // Gauss-Newton system
    double *Hp = nullptr;           // [P][256] per-patch systems
    double *gp = nullptr;           // [P][16]
    double *H9 = nullptr;           // [5][N][16] upper half of the block stencil
    double *Pinv = nullptr;         // [N][16]
    double *g = nullptr;            // [N][4]
    double *lighting = nullptr;     // [16]

    // CG vectors, [N][4] each
    double *x = nullptr, *r = nullptr, *z = nullptr, *Ad = nullptr,
        *d = nullptr, *d2 = nullptr, *b = nullptr;
    double *partials = nullptr;     // [2][8][512] per-block reduction partials of cg.hip (high, low words)
    void *cg_state = nullptr;       // CgState[2] (cg.hip)
    int last_cg_iterations = 0;     // sizes the first chunk of the next solve
    bool cg_use_active = false;     // system built by gn_construct: skip inactive nodes
    double *scalars = nullptr;      // [S_NUM]
    int *status = nullptr;          // [I_NUM]
    int *status_host = nullptr;     // pinned
    int *cg_progress = nullptr;     // pinned, written by the CG kernels (cg.hip)
    int *step_words = nullptr;      // pinned ring of STEP_SLOTS result slots, written by
                                    // finish_step_kernel (update.hip)
    unsigned long long *step_counter = nullptr;   // device, its packed counters
    double *zero_block = nullptr;   // 16 doubles of +0.0 (cg_resident.hip)
    int last_loop_steps = 1 << 30;  // Newton steps of the previous smvs_gn_run_loop (launch-ahead heuristic)
    double *nodes_saved = nullptr;  // smvs_ctx_save_nodes
    size_t nodes_saved_cap = 0;
    int nodes_saved_count = 0, nodes_saved_stride = 0;
    int step_seq = 0;
    int cg_solve_id = 0;
    double *scalars_host = nullptr; // pinned
    double *lightAb = nullptr;      // [272] lighting normal equations
    float *stage = nullptr;         // upload staging (3-channel planes)
    size_t stage_cap = 0;
    void *pin = nullptr;            // pinned host staging of the large transfers
    size_t pin_cap = 0;
    uint8_t *byte_stage = nullptr;  // device staging of smvs_ctx_upload_image
    size_t byte_stage_cap = 0;
    // smvs_ctx_upload_image_async (scale.hip): the DMA of view v runs on a stream
    // of its own into a staging buffer per view; `image_ready[v]` is recorded
    // behind it, and the conversion to float is enqueued on the context's
    // stream (behind a wait for that event) where the image is first needed
    hipStream_t copy_stream = nullptr;
    // the low-resolution SGM map on its way to the device (smvs_ctx_sgm_init_depth):
    // page-locked, read by a kernel over the bus -- a DMA would queue behind the
    // nine image transfers that are under way at that moment
    float *sgm_pin = nullptr;
    size_t sgm_pin_cap = 0;
    bool sgm_pin_busy = false;
    hipEvent_t image_ready[SMVS_MAX_SUBS + 1] = { nullptr };
    uint8_t *upload_stage[SMVS_MAX_SUBS + 1] = { nullptr };
    size_t upload_stage_cap[SMVS_MAX_SUBS + 1] = { 0 };
    uint32_t image_pending = 0;     // bit v: on its way (staged or being converted), not yet waited for
    uint32_t image_direct = 0;      // bit v: converted by the upload itself (no staging buffer)
    uint32_t upload_stage_busy = 0; // bit v: a conversion from upload_stage[v] may be in flight
    // resident PCG (cg_resident.hip)
    double *res_work = nullptr;     // partial sums + barrier words
    double *res_zx = nullptr;       // [cap_nodes][4] z exchanged between workgroups
    size_t res_zx_cap = 0;
    // the resident solver's tiling of the current grid (cg_resident.hip,
    // resident_plan: ~1,000 candidate shapes), kept while its inputs stay
    struct ResidentPlanMemo {
        int stride = -1, nodes = -1, solver_mode = -1, cus = -1;
        bool ok = false;
        int tw = 0, th = 0, one = 0, blocks = 0;
    };
    mutable ResidentPlanMemo res_plan;
    int resident_cus = 0, resident_lds = 0;
    bool resident_disabled = false;
    int solver_mode = 0;            // smvs_solver_mode (smvs_ctx_set_solver)
    float *map_scratch = nullptr;   // depth / normal map output, W*H*3 (or *4) floats
    size_t map_scratch_floats = 0;
    double *light_partial = nullptr;  // per-block lighting sums (update.hip)

    // scale space on the device (scale.hip): float images of the views
    struct ViewImage { int w = 0, h = 0, c = 0; float *data = nullptr; };
    ViewImage images[SMVS_MAX_SUBS + 1];   // [0] main, [1 + j] neighbour j
    float *blur_tmp[2] = { nullptr, nullptr };
    size_t blur_cap = 0;
    float *main_hess_scratch = nullptr;    // set_scale writes no main Hessian

    // topology tests between Newton batches (topology.hip)
    float *topo_zbuf[SMVS_MAX_SUBS] = { nullptr };
    size_t topo_zbuf_cap[SMVS_MAX_SUBS] = { 0 };
    float *topo_sgm = nullptr;
    size_t topo_sgm_cap = 0;
    bool sgm_resident = false;   // topo_sgm holds smvs_ctx_sgm_init_depth's result
    float *sgm_lowres = nullptr; // its input, SGM resolution
    size_t sgm_lowres_cap = 0;
    float *bil_lut = nullptr;    // compressed colour-weight table of the bilateral filter (sgm.hip)
    float *bil_tri = nullptr;    // ... and the triangle of all byte pairs (round 6)
    int topo_slot = 0;           // cut_boundaries: the word pair of the next pass (topology.hip)
    bool topo_slots_clean = false;   // ... and whether both pairs are zero on the stream
    smvs_topo::NccSample *topo_ncc = nullptr;
    int topo_ncc_off[33] = { 0 };
    int topo_ncc_ps = 0;
    double *topo_mse = nullptr;
    size_t topo_mse_cap = 0;
    uint8_t *topo_border = nullptr;   // nodes with > 1 missing neighbour (cut_boundaries)
    size_t topo_border_cap = 0;
    int *topo_mse_list = nullptr;     // patches whose error is evaluated (topology.hip)
    size_t topo_mse_list_cap = 0;
    double *topo_mse_parts = nullptr;  // partial sums of the chunked patch MSE (topology.hip)
    size_t topo_mse_parts_cap = 0;
    int *topo_mse_arrived = nullptr;   // ... and its arrival counters (zero between launches)
    size_t topo_mse_arrived_cap = 0;
    double *topo_pix = nullptr;       // [H][W][3]: surface depth, d/dx, d/dy per pixel
    size_t topo_pix_cap = 0;
    uint8_t *topo_pair_alive = nullptr;   // [P][n_subs]: verdict of the visibility test's
    size_t topo_pair_cap = 0;             // geometric half (topo_visibility_kernel<1>)

    // grid surgery on the device (surface.hip)
    float *surf_depth = nullptr;       // [H][W] Surface::depth (surface.cc:46-50): the
    size_t surf_depth_cap = 0;         // initial depth the nodes are (re)filled from
    bool surf_depth_ok = false;
    double *surf_tmp = nullptr;        // scratch: the nodes before a subdivision,
    size_t surf_tmp_cap = 0;           // the proposals of expand (doubles)
    uint8_t *surf_tmp_bytes = nullptr; // scratch: validity before a subdivision, proposed flags
    size_t surf_tmp_bytes_cap = 0;
    unsigned *surf_bits = nullptr;     // global fallback of the isolated-patch bit columns
    size_t surf_bits_cap = 0;

    smvs_hip::Profile prof;
};

namespace smvs_hip {

// RAII helper that times one kernel class when profiling is enabled.
struct ScopedKernelTimer {
    smvs_ctx *ctx;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedKernelTimer(smvs_ctx *ctx, int cls);
    ~ScopedKernelTimer();
};
int profile_collect(smvs_ctx *ctx);
int ctx_pool_release(void);   // frees the parked contexts (ctx.hip) -> how many
// Large host <-> device transfers of a context through its pinned staging
// buffer (a pageable copy runs at a fraction of the link's rate); both return
// when the caller's buffer may be reused / is filled.
int ctx_upload(smvs_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int ctx_download(smvs_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
// Is `p` page-locked host memory the device can DMA from / to directly
// (smvs_pinned_alloc, hipHostMalloc, hipHostRegister)?
bool host_pointer_is_pinned(const void *p);
int pinned_pool_release(void);   // pool.hip -> buffers returned to the driver
// Geometry of the surface the context holds (Surface::create, surface.cc:28-37)
// and room for it in every per-node / per-patch buffer; the buffers only grow.
// Contents of the node / patch arrays are unspecified afterwards when the grid
// grew.  Leaves has_surface untouched.
int ctx_ensure_grid(smvs_ctx *ctx, int scale, int npx, int npy, int start_x, int start_y);
// Images handed over by smvs_ctx_upload_image_async that have not been
// converted yet: the wait for their DMA and the conversion are enqueued on the
// context's stream (views: bit v = view v - 1 as in image_ok; every consumer of
// ctx->images calls this for the views it reads).
int ctx_materialise_images(smvs_ctx *ctx, uint32_t views);

// hipMalloc that does not give up on memory the library itself is holding:
// when the driver reports out-of-memory, the parked contexts and the idle
// workspaces (pool.hip; hundreds of MB each, kept for the next view of the
// same geometry) go back to the driver and the allocation is tried once more.
int device_malloc(void **ptr, size_t bytes);
// Returns what the pools hold idle to the driver; -> objects freed.
int release_idle_device_memory(void);

// Logical -> physical devices.  Every device index of the C ABI (and
// smvs_ctx::device, Workspace::device, the per-device pools) is a LOGICAL
// device; SMVS_DEVICE_MAP="0,0,1" makes logical device i the HIP device
// map[i] (identity without it).  A node's GPUs can be renumbered or shared
// that way, and the multi-device code -- ViewQueue(num_devices > 1), its
// per-device context / workspace / pinned pools -- can run on a box with one
// GPU (tests/test_gpu_front.py).  What must be shared by logical devices on
// one GPU is keyed by the physical one: the tile budget of the barrier kernels.
int logical_device_count(void);          // -1: HIP error
int physical_device(int logical);        // (the caller has checked the range)
inline hipError_t
set_device(int logical)
{
    return hipSetDevice(physical_device(logical));
}

// Launches with more than 64 KB of dynamic LDS need
// hipFuncAttributeMaxDynamicSharedMemorySize raised for the kernel on the
// device: done once per (device, kernel) and size reached, under a mutex, for
// any device index.  Returns SMVS_ERR_HIP when the device cannot grant it.
int allow_dynamic_lds(int device, const void *kernel, size_t bytes);

template <typename T>
int device_alloc(T **ptr, size_t count)
{
    if (*ptr != nullptr) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
    }
    if (count == 0)
        return SMVS_OK;
    return device_malloc(reinterpret_cast<void **>(ptr), count * sizeof(T));
}

// A reusable device workspace of the context-free entry points (pool.hip): a
// stream, named device buffers that only grow, pinned staging memory.
struct Workspace {
    enum { SLOTS = 24 };
    struct Buf { void *p = nullptr; size_t cap = 0; };
    int device = 0;
    hipStream_t stream = nullptr;
    Buf dev[SLOTS];
    void *pinned = nullptr;
    size_t pinned_cap = 0, pinned_used = 0;
    int ensure(int slot, size_t bytes, void **out);
    template <typename T> int ensure(int slot, size_t count, T **out)
    {
        void *p = nullptr;
        int const rc = ensure(slot, count * sizeof(T), &p);
        *out = static_cast<T *>(p);
        return rc;
    }
    int ensure_pinned(size_t bytes);
    int upload(void *dst_dev, const void *src_host, size_t bytes);
    int download(void *dst_host, const void *src_dev, size_t bytes);
};
Workspace *workspace_acquire(int device);   // nullptr: error text set
void workspace_release(Workspace *w);
struct WorkspaceLease {
    Workspace *w;
    explicit WorkspaceLease(int device) : w(workspace_acquire(device)) {}
    ~WorkspaceLease() { workspace_release(w); }
    WorkspaceLease(WorkspaceLease const &) = delete;
    WorkspaceLease &operator=(WorkspaceLease const &) = delete;
};

// Maps the hardware block index to a logical block so that the blocks that
// land on one XCD (blockIdx % 8, MI355X_MICROARCH.md "Workgroup dispatch")
// cover one contiguous band of patches: neighbouring patches sample
// neighbouring texels, so each XCD's private L2 sees one image region.
__device__ __forceinline__ unsigned
xcd_band_block(unsigned bid, unsigned nblocks)
{
    unsigned const per = nblocks >> 3;
    unsigned const body = per << 3;
    if (bid >= body)
        return bid;
    return (bid & 7u) * per + (bid >> 3);
}

// Per-patch normal equations as stored between the two kernels (144 doubles):
// the four diagonal node blocks as their UPPER TRIANGLES, 10 entries in 12
// doubles {h00 h01 h02 h03} {h11 h12 h13 h22} {h23 h33 - -} (the assembly never
// read the lower triangles: gauss_newton_step.cc:103, 113-119), then the six
// blocks (bi < bj) of the upper block triangle, [4][4] doubles each.  (Round 3:
// all ten blocks in full, 160 doubles -- 10 % more to write and to read back.)
constexpr int PATCH_DIAG_STRIDE = 12;
constexpr int PATCH_H_STRIDE = 4 * PATCH_DIAG_STRIDE + 6 * 16;
// index of (i, j), i <= j, in the packed upper triangle of a 4 x 4 (also: of
// node block (bi, bj) among the ten blocks of the upper block triangle)
__host__ __device__ __forceinline__ constexpr int
upper_block(int bi, int bj)
{
    return bi * 4 - bi * (bi - 1) / 2 + (bj - bi);
}
__host__ __device__ __forceinline__ constexpr int
patch_diag_offset(int b)
{
    return b * PATCH_DIAG_STRIDE;
}
__host__ __device__ __forceinline__ constexpr int
patch_upper_offset(int bi, int bj)   // bi < bj
{
    return 4 * PATCH_DIAG_STRIDE + (upper_block(bi, bj) - (bi + 1)) * 16;
}

// Where the 36 quads (4 doubles = 32 bytes) of a patch's packed system and the
// 4 quads of its gradient live in HBM.  Two strides, in quads, per buffer:
// quad j of patch p sits at quad index j * quad_stride + p * patch_stride.
//   planar (default since round 5): quad_stride = P rounded up to 4,
//     patch_stride = 1 -- 36 (4) planes of one quad per patch.  The consumer is
//     the resident solver's prologue, whose thread for node n reads the same
//     quad of patches p, p + 1, ... in neighbouring lanes: one contiguous
//     32 bytes x 64 lanes per load instead of 64 lines 1,152 bytes apart (the
//     prologue was bound by the address rate of its texture unit: 2.7 TB/s);
//   patch-major (SMVS_HP_LAYOUT=aos, rounds 1-4): quad_stride = 1,
//     patch_stride = 36 (4).
// The values are the same either way; tests run both.
struct PatchLayout {
    unsigned hq, hp;    // Hp: quad stride, patch stride
    unsigned gq, gp;    // gp
};
__host__ __device__ __forceinline__ size_t
patch_h_at(PatchLayout const &L, size_t patch, int element)     // index in doubles
{
    return ((size_t)(element >> 2) * L.hq + patch * L.hp) * 4 + (size_t)(element & 3);
}
__host__ __device__ __forceinline__ size_t
patch_g_at(PatchLayout const &L, size_t patch, int element)
{
    return ((size_t)(element >> 2) * L.gq + patch * L.gp) * 4 + (size_t)(element & 3);
}
PatchLayout patch_layout(const smvs_ctx *ctx);


// lib/ldl_decomposition.h:43-92 for a 4x4 block, same operation order.
__device__ __forceinline__ void
ldl_inverse4(double A[16])
{
#pragma clang fp contract(off)
    double L[16], D[4];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        L[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        D[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        D[j] = A[j * 4 + j];
        L[j * 4 + j] = 1.0;
#pragma unroll
        for (int k = 0; k < j; ++k)
            D[j] -= (L[j * 4 + k] * L[j * 4 + k]) * D[k];
        if (D[j] == 0.0)
            return;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            L[i * 4 + j] = A[i * 4 + j];
#pragma unroll
            for (int k = 0; k < j; ++k)
                L[i * 4 + j] -= L[i * 4 + k] * D[k] * L[j * 4 + k];
            L[i * 4 + j] /= D[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            double sum = 0.0;
#pragma unroll
            for (int k = i; k < j; ++k)
                sum -= L[j * 4 + k] * L[k * 4 + i];
            L[j * 4 + i] = sum;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        D[i] = 1.0 / D[i];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        A[i] = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2)
                A[c1 * 4 + c2] += L[r * 4 + c2] * L[r * 4 + c1] * D[r];
}

// reasons in status[I_STEP_ABORT]; finish_step_kernel reports 1 + reason
constexpr int ABORT_SOLVER = 1;   // the resident solver gave up (workgroups not co-resident)
constexpr int ABORT_GRID = 2;     // a launch was sized for a shorter live list
constexpr int STEP_SLOTS = 4;        // steps whose result words can coexist
constexpr int STEP_SLOT_INTS = 16;   // one 64-byte line per slot

// Launch-ahead mode of the Newton loop: the launches of a step are sized and
// enqueued before the previous step's result is known; the kernels read the
// list length and the loop's stop word on the device.
struct StepPipeline {
    int seq;             // sequence tag finish_step_kernel publishes (never 0)
    int grid_live;       // list length the launches are sized for
    double full_opt_threshold;
};

// internal entry points used by the fused loop
// known_live: entries of the live-patch list the previous reactivate_launch
// of the same Newton loop built (its count read back by the host), or -1 to
// build the list here.
// skip_assembly: leave the per-patch systems unassembled (the resident solver
// gathers them itself, cg_resident_solve(..., fused = true)).
int gn_construct_launch(smvs_ctx *ctx, double reg, double light_reg,
    bool use_lighting, int known_live = -1, bool skip_assembly = false,
    bool check_stop = false);
int gn_assemble_launch(smvs_ctx *ctx);
int live_patch_list_launch(smvs_ctx *ctx);
int cg_solve_launch(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info);
// known_live >= 0: walk the live list (its length as read back by the host);
// publish_seq != 0: the Newton loop's fused end of step (finish_step_kernel
// publishes the result words tagged with this sequence number).
int reactivate_launch(smvs_ctx *ctx, double threshold, int full_optimization,
    bool build_live_list = false, int known_live = -1, int publish_seq = 0,
    const StepPipeline *pipe = nullptr);
// fused: assemble H, g, P from the per-patch systems inside the kernel (the
// caller has NOT run the assembly kernel); *ran = false means nothing was done.
int cg_resident_solve(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info, bool *ran,
    bool fused = false);
bool cg_resident_applies(smvs_ctx *ctx, int max_iterations);
// Barrier kernels (the resident PCG: every workgroup of a launch must be
// co-resident, one per CU) share a device by a TILE BUDGET: a Newton loop
// (update.hip holds its share for the whole loop, patch kernels included) or a
// single solve acquires as many tiles as its grid has workgroups and waits
// while the tiles in use plus its own exceed the device's CUs.  A full-size
// solve (256 tiles at 1920x1080, scale 2) therefore still runs alone, but the
// loops of the coarse scales -- 1, 4, 16, 64 tiles -- of several views in
// flight run side by side instead of taking turns (round 3's exclusive lock).
// Requests are served in arrival order, so a large request is not starved by
// a stream of small ones.  Across PROCESSES that share the GPU the budget
// cannot be shared; there an advisory lock on a file named after the device's
// PCI bus id still makes the processes take turns: two such kernels of two
// processes started together could each hold half of the CUs and wait for the
// other half for ever.  The file lock is taken without the mutex held, kept
// while loops of this process follow each other, and handed back after 100 ms at
// the latest so that a process waiting for it gets its turn (cg_resident.hip).
class DeviceTileBudget {
public:
    void acquire(int device, int tiles);
    void release(int tiles);
private:
    void bind(int device);      // capacity, lock file: once
    bool take_file_lock(void);
    void unlock_file(void);
    std::mutex mutex;
    std::condition_variable turn;
    int capacity = 0, used = 0, holders = 0;
    unsigned long long next_ticket = 0, serving = 0;
    int fd = -1;
    bool bound = false, file_locked = false;
    std::chrono::steady_clock::time_point file_since{}, no_file_until{};
};
DeviceTileBudget &cg_resident_budget(int device);
// workgroups (= tiles = CUs) the resident solver launches for this context's
// grid; 0 when it does not apply
int cg_resident_tiles(smvs_ctx *ctx);
struct ScopedTileBudget {
    DeviceTileBudget &budget;
    int tiles;
    ScopedTileBudget(int device, int tiles_) : budget(cg_resident_budget(device)), tiles(tiles_)
    {
        budget.acquire(physical_device(device), tiles);
    }
    ~ScopedTileBudget() { budget.release(tiles); }
    ScopedTileBudget(ScopedTileBudget const &) = delete;
    ScopedTileBudget &operator=(ScopedTileBudget const &) = delete;
};
int cg_resident_enqueue(smvs_ctx *ctx, int max_iterations, double q_tolerance,
    bool test_give_up = false);
int cg_resident_gave_up(smvs_ctx *ctx);   // a resident solve failed: what runs from now on


} // namespace smvs_hip
