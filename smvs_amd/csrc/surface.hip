// Grid surgery of smvs::Surface on the device (the part
// rounds 1-3 left on the host): Surface::create's node initialisation from a
// depth map (lib/surface.cc:19-53, 140-152, 667-760), subdivide_patches
// (:983-1107), expand (:482-628), fill_holes (:630-651),
// remove_nodes_without_patch (:762-869), remove_isolated_patches (:887-927).
//
// With these the surface never leaves the context between the Newton batches
// of DepthOptimizer::optimize: no smvs_ctx_set_surface / smvs_get_nodes per
// batch (a view's wall time was three times its GPU time).  The host keeps the
// geometry (a handful of integers) and reads one word per operation: the
// number of valid patches, which drives the optimiser's iteration rule
// (depth_optimizer.cc:339-356).
//
// Every kernel is a gather for one output element with the per-element
// arithmetic of csrc/host/surface_math.h -- the source the C++ host mirror
// compiles too (and the CPU tests compare with the oracle).  The reference's
// loops scatter; where a scatter's result depends on the visiting order
// (edge midpoints of a subdivision, the in-place deletions of
// remove_isolated_patches) the gather reproduces that order's outcome.
// Double arithmetic in source order, contraction off: nodes are bit-identical
// with the host mirror's.
#include "common.h"

#include "host/surface_math.h"

#include <algorithm>
#include <vector>

namespace smvs_hip {

using smvs_surf::Grid;

struct SurfArgs {
    double *nodes;
    uint8_t *node_valid;
    uint8_t *patch_valid;
    const float *depth;
    Grid g;
    int stride, num_nodes, num_patches;
    int *status;             // device status words (I_SURF_*)
    // subdivision: the surface before it
    const double *old_nodes;
    const uint8_t *old_node_valid;
    const uint8_t *old_patch_valid;
    int old_npx, old_npy, off_x, off_y;
    // expand
    double *proposal;
    uint8_t *proposed;
};

// ---- Surface::create with an initial depth map (surface.cc:74-79) ----
__global__ void __launch_bounds__(256)
surf_clamp_depth_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float const v = src[i];
        dst[i] = v > 0.0f ? v : 0.0f;
    }
}

// ---- initialize_depth_from_bundle (surface.cc:90-130): the projected
// features, already reduced by the caller to one entry per pixel ----
__global__ void __launch_bounds__(256)
surf_scatter_depth_kernel(const int *__restrict__ pixel, const float *__restrict__ value,
    int n, float *__restrict__ depth)
{
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        depth[pixel[i]] = value[i];
}

// ---- initialize_node_from_depth (surface.cc:667-760) for every null node ----
// One lane group per node (G = min(64, ps^2) lanes).  The window's ps^2 pixels
// are dealt to the lanes; quadrant minima and the count of positive depths by
// group reductions; the median -- std::nth_element's element of rank
// count / 2 -- by a radix selection on the bit patterns of the positive
// floats, most significant bit first: 31 counting passes over the window's
// keys, which a lane keeps in registers (CACHE = 4, 16 or 64 of them: patch
// sizes up to 64; beyond that the passes re-read the depth map, L1 hits).
template <int CACHE>
__global__ void __launch_bounds__(256)
surf_init_nodes_kernel(SurfArgs A, int G, int glog)
{
#pragma clang fp contract(off)
    Grid const g = A.g;
    int const T = g.ps * g.ps;                 // pixels of a window
    int const E = T / G;                       // per lane (<= CACHE when CACHE > 0)
    int const lane = threadIdx.x & 63;
    int const gl = lane & (G - 1);
    long long const gid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> glog;
    bool const in_range = gid < (long long)A.num_nodes;
    int const id = in_range ? (int)gid : 0;
    int const idx = id % A.stride, idy = id / A.stride;
    bool const todo = in_range && A.node_valid[id] == 0;
    // key of window element e (0: outside the image or no depth) and its quadrant
    auto load_key = [&](int e, int *q) -> unsigned {
        int xx = 0, yy = 0;
        *q = 0;
        if (todo && smvs_surf::window_pixel(g, idx, idy, e, q, &xx, &yy)) {
            float const d = A.depth[(size_t)yy * g.width + xx];
            if (d > 0.0f)
                return smvs_surf::depth_key(d);
        }
        return 0u;
    };
    unsigned keys[CACHE > 0 ? CACHE : 1] = { 0u };
    unsigned lowq[4] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu };
    int count = 0;
    auto account = [&](unsigned key, int q) {
        if (key != 0u) {
            count += 1;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q == k)
                    lowq[k] = key < lowq[k] ? key : lowq[k];
        }
    };
    if (CACHE > 0) {
#pragma unroll
        for (int i = 0; i < (CACHE > 0 ? CACHE : 1); ++i) {
            int q = 0;
            unsigned const key = i < E ? load_key(gl + G * i, &q) : 0u;
            keys[i] = key;
            account(key, q);
        }
    } else {
        for (int i = 0; i < E; ++i) {
            int q = 0;
            unsigned const key = load_key(gl + G * i, &q);
            account(key, q);
        }
    }
    // (whole groups take every branch together: the shuffles stay inside a group)
    for (int off = G >> 1; off > 0; off >>= 1) {
        count += __shfl_xor(count, off);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned const o = __shfl_xor(lowq[k], off);
            lowq[k] = o < lowq[k] ? o : lowq[k];
        }
    }
    // rank selection over the positive keys, most significant bit first
    int rank = count / 2;
    unsigned prefix = 0u, mask = 0u;
    if (__any(todo && count >= 2)) {
        for (int b = 30; b >= 0; --b) {
            unsigned const bit = 1u << b;
            int zeros = 0;
            if (CACHE > 0) {
#pragma unroll
                for (int i = 0; i < (CACHE > 0 ? CACHE : 1); ++i) {
                    unsigned const key = keys[i];
                    zeros += (key != 0u && (key & mask) == prefix && (key & bit) == 0u)
                        ? 1 : 0;
                }
            } else {
                for (int i = 0; i < E; ++i) {
                    int q = 0;
                    unsigned const key = load_key(gl + G * i, &q);
                    zeros += (key != 0u && (key & mask) == prefix && (key & bit) == 0u)
                        ? 1 : 0;
                }
            }
            for (int off = G >> 1; off > 0; off >>= 1)
                zeros += __shfl_xor(zeros, off);
            if (rank >= zeros) {
                rank -= zeros;
                prefix |= bit;
            }
            mask |= bit;
        }
    }
    if (todo && gl == 0) {
        int quadrants = 4;
        double lowest[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool const any = lowq[k] != 0xFFFFFFFFu;
            lowest[k] = any ? (double)__uint_as_float(lowq[k]) : 0.0;
            quadrants -= any ? 0 : 1;
        }
        double node[4];
        if (smvs_surf::node_from_window((double)__uint_as_float(prefix), lowest,
                quadrants, (size_t)count, node)) {
            double *dst = A.nodes + 4 * (size_t)id;
            dst[0] = node[0]; dst[1] = node[1]; dst[2] = node[2]; dst[3] = node[3];
            A.node_valid[id] = 1;
        }
    }
}

// ---- fill_holes (surface.cc:630-651): a null patch whose four nodes exist ----
__global__ void __launch_bounds__(256)
surf_fill_holes_kernel(SurfArgs A)
{
    int const p = blockIdx.x * blockDim.x + threadIdx.x;
    bool filled = false;
    if (p < A.num_patches && !A.patch_valid[p]) {
        int const ix = p % A.g.npx, iy = p / A.g.npx;
        size_t const n00 = (size_t)iy * A.stride + ix;
        if (A.node_valid[n00] && A.node_valid[n00 + 1] && A.node_valid[n00 + A.stride]
            && A.node_valid[n00 + A.stride + 1]) {
            A.patch_valid[p] = 1;
            filled = true;
        }
    }
    int const cnt = __syncthreads_count(filled);
    if (threadIdx.x == 0 && cnt != 0)
        atomicAdd(A.status + I_SURF_CHANGED, cnt);
}

// ---- remove_nodes_without_patch (surface.cc:762-869) ----
__global__ void __launch_bounds__(256)
surf_remove_nodes_kernel(SurfArgs A)
{
    int const n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.num_nodes || !A.node_valid[n])
        return;
    int const idx = n % A.stride, idy = n / A.stride;
    bool any = false;
    for (int dy = -1; dy <= 0; ++dy)
        for (int dx = -1; dx <= 0; ++dx) {
            int const qx = idx + dx, qy = idy + dy;
            if (qx >= 0 && qy >= 0 && qx < A.g.npx && qy < A.g.npy
                && A.patch_valid[(size_t)qy * A.g.npx + qx])
                any = true;
        }
    if (!any)
        A.node_valid[n] = 0;
}

// ---- number of valid patches -> status[I_SURF_VALID] ----
__global__ void __launch_bounds__(256)
surf_count_kernel(SurfArgs A)
{
    int cnt = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < A.num_patches;
         p += gridDim.x * blockDim.x)
        cnt += A.patch_valid[p] ? 1 : 0;
    __shared__ int total;
    if (threadIdx.x == 0)
        total = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1)
        cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt != 0)
        atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && total != 0)
        atomicAdd(A.status + I_SURF_VALID, total);
}

// ---- subdivide_patches (surface.cc:983-1107): one thread per new node ----
__global__ void __launch_bounds__(256)
surf_subdivide_kernel(SurfArgs A)
{
#pragma clang fp contract(off)
    int const n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.num_nodes)
        return;
    int const X = n % A.stride, Y = n / A.stride;
    double out[4] = { 0.0, 0.0, 0.0, 0.0 };
    bool const ok = smvs_surf::subdivide_node(A.old_npx, A.old_npy, A.off_x, A.off_y,
        A.old_nodes, A.old_node_valid, A.old_patch_valid, X, Y, out);
    double *dst = A.nodes + 4 * (size_t)n;
    dst[0] = ok ? out[0] : 0.0;
    dst[1] = ok ? out[1] : 0.0;
    dst[2] = ok ? out[2] : 0.0;
    dst[3] = ok ? out[3] : 0.0;
    A.node_valid[n] = ok ? 1 : 0;
}

// ---- expand (surface.cc:482-628): one round = propose, then apply ----
__global__ void __launch_bounds__(256)
surf_expand_propose_kernel(SurfArgs A)
{
#pragma clang fp contract(off)
    int const n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.num_nodes)
        return;
    if (A.node_valid[n] && !A.proposed[n])
        return;
    smvs_surf::expand_node(A.g.npx, A.g.npy, A.nodes, A.node_valid, n % A.stride,
        n / A.stride, A.proposal + n, A.proposed + n);
}

__global__ void __launch_bounds__(256)
surf_expand_apply_kernel(SurfArgs A)
{
    int const n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.num_nodes || !A.proposed[n])
        return;
    double *dst = A.nodes + 4 * (size_t)n;
    dst[0] = A.proposal[n];
    dst[1] = dst[2] = dst[3] = 0.0;
    A.node_valid[n] = 1;
}

// ---- remove_isolated_patches (surface.cc:887-927) ----
// The reference deletes in place while it walks the grid column by column, so
// the walk's order is part of the result: when patch (x, y) is examined, its
// neighbours (x-1, y-1), (x-1, y), (x-1, y+1), (x, y-1) have been through the
// walk already (their deletions count), the other four have not (they count
// as they were).  With del(p) = "the walk deletes p" that is a recurrence on a
// DAG,
//     del(p) = valid(p) and  #valid neighbours - #deleted earlier neighbours < 3,
// whose unique solution a relaxation reaches from any start once every patch
// has seen final values of its four predecessors.  ONE workgroup relaxes all
// patches at once, 32 rows of a column per thread and step as bit vectors
// (one-cell border of zeros, rows packed into 32-bit words, in LDS -- or in
// global memory when the grid is too large for it): the eight neighbour counts
// of 32 patches are a bit-sliced adder, ~40 logic operations.  The words are
// updated in place (a neighbour's word is read either before or after its
// update, both are states of the relaxation); a sweep in which no word changed
// read one consistent state and ends the kernel.  The number of sweeps is the
// length of the longest chain of deletions each caused by the one before --
// a handful on real surfaces (deletions eat into ragged borders, not along
// them), bounded by the 2 npx + npy steps of the walk itself.
// (Rounds 3-4 replayed the walk as a wavefront of 2 npx + npy steps of
// independent cells, one barrier each: 494 us at 478 x 268 however few
// patches go; profiles/r4_optimize_timeline_*.txt before / after.)
__global__ void __launch_bounds__(1024)
surf_isolated_kernel(SurfArgs A, unsigned *global_bits, int wpc)
{
    extern __shared__ unsigned lds_bits[];
    int const npx = A.g.npx, npy = A.g.npy;
    int const cols = npx + 2;
    int const words = cols * wpc;
    unsigned *orig = global_bits != nullptr ? global_bits : lds_bits;   // [cols][wpc]
    unsigned *del = orig + words;                                        // [cols][wpc]
    for (int i = threadIdx.x; i < 2 * words; i += blockDim.x)
        orig[i] = 0u;
    __syncthreads();
    // column c = x + 1, row r = y + 1; a thread packs 32 rows of one column
    for (int i = threadIdx.x; i < npx * wpc; i += blockDim.x) {
        int const x = i / wpc, w = i - x * wpc;
        unsigned word = 0u;
        for (int b = 0; b < 32; ++b) {
            int const y = w * 32 + b - 1;
            if (y >= 0 && y < npy && A.patch_valid[(size_t)y * npx + x])
                word |= 1u << b;
        }
        orig[(x + 1) * wpc + w] = word;
    }
    __syncthreads();
    // bit r of the result = bit r - 1 / r + 1 of the column
    auto from_above = [&](const volatile unsigned *col, int w) -> unsigned {
        return (col[w] << 1) | (w > 0 ? col[w - 1] >> 31 : 0u);
    };
    auto from_below = [&](const volatile unsigned *col, int w) -> unsigned {
        return (col[w] >> 1) | (w + 1 < wpc ? col[w + 1] << 31 : 0u);
    };
    int const max_sweeps = 2 * npx + npy + 2;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        bool changed = false;
        for (int i = threadIdx.x; i < npx * wpc; i += blockDim.x) {
            int const c = 1 + i / wpc, w = i % wpc;
            unsigned const mine = orig[c * wpc + w];
            if (mine == 0u)
                continue;
            const volatile unsigned *Lo = orig + (c - 1) * wpc, *Ld = del + (c - 1) * wpc;
            const volatile unsigned *Mo = orig + c * wpc, *Md = del + c * wpc;
            const volatile unsigned *Ro = orig + (c + 1) * wpc;
            // the left column and the cell above as the walk left them ...
            unsigned const l_above = from_above(Lo, w) & ~from_above(Ld, w);
            unsigned const l_same = Lo[w] & ~Ld[w];
            unsigned const l_below = from_below(Lo, w) & ~from_below(Ld, w);
            unsigned const m_above = from_above(Mo, w) & ~from_above(Md, w);
            // ... the cell below and the right column as they were
            unsigned const m_below = from_below(Mo, w);
            unsigned const r_above = from_above(Ro, w), r_same = Ro[w],
                           r_below = from_below(Ro, w);
            // 32 sums of eight bits: ones s0, twos s1, fours s2, eights s3
            unsigned const sa = l_above ^ l_same ^ l_below;
            unsigned const ca = (l_above & l_same) | (l_below & (l_above ^ l_same));
            unsigned const sb = m_above ^ m_below ^ r_above;
            unsigned const cb = (m_above & m_below) | (r_above & (m_above ^ m_below));
            unsigned const sc = r_same ^ r_below, cc = r_same & r_below;
            unsigned const s0 = sa ^ sb ^ sc;
            unsigned const cd = (sa & sb) | (sc & (sa ^ sb));
            unsigned const ts = ca ^ cb ^ cc;
            unsigned const tc = (ca & cb) | (cc & (ca ^ cb));
            unsigned const s1 = ts ^ cd, u = ts & cd;
            unsigned const s2 = tc ^ u, s3 = tc & u;
            unsigned const three_or_more = s3 | s2 | (s1 & s0);
            unsigned const now = mine & ~three_or_more;
            if (now != del[c * wpc + w]) {
                del[c * wpc + w] = now;
                changed = true;
            }
        }
        if (!__syncthreads_or(changed ? 1 : 0))
            break;
    }
    int removed = 0;
    for (int p = threadIdx.x; p < npx * npy; p += blockDim.x) {
        int const x = p % npx, y = p / npx;
        int const c = x + 1, r = y + 1;
        unsigned const gone = (del[c * wpc + (r >> 5)] >> (r & 31)) & 1u;
        if (gone != 0u) {
            // (del is a subset of the valid patches)
            A.patch_valid[p] = 0;
            removed += 1;
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        removed += __shfl_xor(removed, off);
    if ((threadIdx.x & 63) == 0 && removed != 0)
        atomicAdd(A.status + I_SURF_CHANGED, removed);
}

// ---- create_subview_surfaces' tail (depth_optimizer.cc:592-603): patches
// nobody sees go away ----
__global__ void __launch_bounds__(256)
surf_delete_unseen_kernel(SurfArgs A, const uint32_t *__restrict__ patch_vis)
{
    int const p = blockIdx.x * blockDim.x + threadIdx.x;
    bool removed = false;
    if (p < A.num_patches && A.patch_valid[p] && patch_vis[p] == 0u) {
        A.patch_valid[p] = 0;
        removed = true;
    }
    int const cnt = __syncthreads_count(removed);
    if (threadIdx.x == 0 && cnt != 0)
        atomicAdd(A.status + I_SURF_CHANGED, cnt);
}

// test hook of the scripted sequences: every k-th valid patch (in id order)
__global__ void __launch_bounds__(1024)
surf_delete_every_kernel(SurfArgs A, int every)
{
    // one workgroup, sequential ranks through a running prefix
    __shared__ int base;
    __shared__ int wave_cnt[16];
    if (threadIdx.x == 0)
        base = 0;
    __syncthreads();
    for (int start = 0; start < A.num_patches; start += blockDim.x) {
        int const p = start + threadIdx.x;
        bool const valid = p < A.num_patches && A.patch_valid[p] != 0;
        unsigned long long const ballot = __ballot(valid);
        int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0)
            wave_cnt[wave] = __popcll(ballot);
        __syncthreads();
        int before = base + __popcll(ballot & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w)
            before += wave_cnt[w];
        if (valid && ((before + 1) % every) == 0)
            A.patch_valid[p] = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
                total += wave_cnt[w];
            base += total;
        }
        __syncthreads();
    }
}

static void
fill_surf_args(smvs_ctx *ctx, SurfArgs *A)
{
    A->nodes = ctx->nodes;
    A->node_valid = ctx->node_valid;
    A->patch_valid = ctx->patch_valid;
    A->depth = ctx->surf_depth;
    A->g.width = ctx->width;
    A->g.height = ctx->height;
    A->g.scale = ctx->scale;
    A->g.ps = ctx->patchsize;
    A->g.npx = ctx->npx;
    A->g.npy = ctx->npy;
    A->g.start_x = ctx->start_x;
    A->g.start_y = ctx->start_y;
    A->stride = ctx->node_stride;
    A->num_nodes = ctx->num_nodes;
    A->num_patches = ctx->num_patches;
    A->status = ctx->status;
    A->old_nodes = nullptr;
    A->old_node_valid = nullptr;
    A->old_patch_valid = nullptr;
    A->old_npx = A->old_npy = A->off_x = A->off_y = 0;
    A->proposal = nullptr;
    A->proposed = nullptr;
}

static unsigned
blocks_for(size_t n, int threads = 256)
{
    return (unsigned)((n + (size_t)threads - 1) / (size_t)threads);
}

static int
ensure_tmp(smvs_ctx *ctx, size_t doubles, size_t bytes)
{
    int rc;
    if (ctx->surf_tmp_cap < doubles) {
        ctx->surf_tmp_cap = 0;
        if ((rc = device_alloc(&ctx->surf_tmp, doubles + doubles / 8)) != SMVS_OK)
            return rc;
        ctx->surf_tmp_cap = doubles + doubles / 8;
    }
    if (ctx->surf_tmp_bytes_cap < bytes) {
        ctx->surf_tmp_bytes_cap = 0;
        if ((rc = device_alloc(&ctx->surf_tmp_bytes, bytes + bytes / 8)) != SMVS_OK)
            return rc;
        ctx->surf_tmp_bytes_cap = bytes + bytes / 8;
    }
    return SMVS_OK;
}

// the grid changed: everything derived from the old surface is stale
static void
surface_changed(smvs_ctx *ctx)
{
    ctx->has_system = false;
    ctx->cg_use_active = false;
    ctx->update_prepared = false;
    ctx->nodes_saved_count = 0;
}

static void
launch_fill_holes(smvs_ctx *ctx, SurfArgs const &A)
{
    hipLaunchKernelGGL(surf_fill_holes_kernel, dim3(blocks_for((size_t)A.num_patches)),
        dim3(256), 0, ctx->stream, A);
}

static void
launch_remove_nodes(smvs_ctx *ctx, SurfArgs const &A)
{
    hipLaunchKernelGGL(surf_remove_nodes_kernel, dim3(blocks_for((size_t)A.num_nodes)),
        dim3(256), 0, ctx->stream, A);
}

// initialize_node_from_depth for all nodes + fill_holes +
// remove_nodes_without_patch (fill_patches_from_depth, surface.cc:140-152)
static int
launch_fill_from_depth(smvs_ctx *ctx, SurfArgs const &A)
{
    int const ps = A.g.ps;
    if (ps >= 2) {
        int const T = ps * ps;
        int const G = T < 64 ? T : 64;
        int glog = 0;
        while ((1 << glog) < G)
            glog += 1;
        size_t const threads = (size_t)A.num_nodes * (size_t)G;
        int const E = T / G;
        // (the window's keys of a lane in registers up to 64 of them -- patch
        // size 64, the coarsest scale of a 1920 x 1080 view: the 31 counting
        // passes of the rank selection re-read nothing)
        if (E <= 4)
            hipLaunchKernelGGL((surf_init_nodes_kernel<4>), dim3(blocks_for(threads)),
                dim3(256), 0, ctx->stream, A, G, glog);
        else if (E <= 16)
            hipLaunchKernelGGL((surf_init_nodes_kernel<16>), dim3(blocks_for(threads)),
                dim3(256), 0, ctx->stream, A, G, glog);
        else if (E <= 64)
            hipLaunchKernelGGL((surf_init_nodes_kernel<64>), dim3(blocks_for(threads)),
                dim3(256), 0, ctx->stream, A, G, glog);
        else
            hipLaunchKernelGGL((surf_init_nodes_kernel<0>), dim3(blocks_for(threads)),
                dim3(256), 0, ctx->stream, A, G, glog);
    }
    // (ps = 1: the window is empty, no node is initialised, surface.cc:700)
    launch_fill_holes(ctx, A);
    launch_remove_nodes(ctx, A);
    SMVS_HIP_CHECK(hipGetLastError());
    return SMVS_OK;
}

// valid patches (and the "changed" word) to the host: one synchronisation --
// or none when the caller asks for neither.  The operations themselves never
// need the host: DepthOptimizer enqueues the sequence between two Newton
// batches (cuts, expand, subview surfaces, isolated patches) and asks for the
// number of valid patches once, with the last of them
// (lib/depth_optimizer.cc:339-356 is all that reads it).
static int
read_counts(smvs_ctx *ctx, SurfArgs const &A, int *valid, int *changed)
{
    if (valid == nullptr && changed == nullptr)
        return SMVS_OK;
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->status + I_SURF_VALID, 0, sizeof(int),
        ctx->stream));
    int const blocks = (int)std::min<size_t>(blocks_for((size_t)A.num_patches), 512);
    hipLaunchKernelGGL(surf_count_kernel, dim3((unsigned)std::max(blocks, 1)), dim3(256), 0,
        ctx->stream, A);
    SMVS_HIP_CHECK(hipGetLastError());
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host + I_SURF_VALID,
        ctx->status + I_SURF_VALID, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (valid != nullptr)
        *valid = ctx->status_host[I_SURF_VALID];
    if (changed != nullptr)
        *changed = ctx->status_host[I_SURF_CHANGED];
    return SMVS_OK;
}

static int
require_surface(smvs_ctx *ctx, const char *who)
{
    if (ctx == nullptr) {
        set_error("%s: null context", who);
        return SMVS_ERR_INVALID;
    }
    if (!ctx->has_surface) {
        set_error("%s: the context holds no surface (smvs_surface_create / "
            "smvs_ctx_set_surface)", who);
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    return SMVS_OK;
}

} // namespace smvs_hip

using namespace smvs_hip;

static_assert(I_SURF_CHANGED == I_SURF_VALID + 1, "read_counts copies both words at once");

extern "C" int
smvs_surface_create(smvs_ctx *ctx, int scale, const float *depth,
    const int32_t *point_pixel, const float *point_depth, int n_points,
    int *num_valid_patches)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    SMVS_REQUIRE(n_points >= 0 && (n_points == 0 || (point_pixel && point_depth)),
        "bad point list");
    SMVS_REQUIRE(scale >= 0 && scale <= 10, "scale out of range");
    SMVS_HIP_CHECK(set_device(ctx->device));
    Grid const g = smvs_surf::grid_for_scale(ctx->width, ctx->height, scale);
    SMVS_REQUIRE(g.npx >= 1 && g.npy >= 1, "image too small for this scale");
    size_t const npix = (size_t)ctx->width * ctx->height;
    int rc;
    if (ctx->surf_depth_cap < npix) {
        ctx->surf_depth_cap = 0;
        if ((rc = device_alloc(&ctx->surf_depth, npix)) != SMVS_OK)
            return rc;
        ctx->surf_depth_cap = npix;
    }
    ctx->surf_depth_ok = false;
    if (depth != nullptr) {
        // through the context's staging buffer, then the clamp of :74-79
        if ((rc = ctx_upload(ctx, ctx->surf_depth, depth, npix * sizeof(float))) != SMVS_OK)
            return rc;
        hipLaunchKernelGGL(surf_clamp_depth_kernel, dim3(blocks_for(npix)), dim3(256), 0,
            ctx->stream, ctx->surf_depth, ctx->surf_depth, npix);
    } else if (n_points > 0) {
        for (int i = 0; i < n_points; ++i)
            SMVS_REQUIRE(point_pixel[i] >= 0 && (size_t)point_pixel[i] < npix,
                "point outside the image");
        SMVS_HIP_CHECK(hipMemsetAsync(ctx->surf_depth, 0, npix * sizeof(float), ctx->stream));
        // pixel indices and depths through the scratch buffers
        size_t const bytes = (size_t)n_points * (sizeof(int) + sizeof(float));
        if ((rc = ensure_tmp(ctx, 0, bytes)) != SMVS_OK)
            return rc;
        int *pix_dev = reinterpret_cast<int *>(ctx->surf_tmp_bytes);
        float *val_dev = reinterpret_cast<float *>(pix_dev + n_points);
        if ((rc = ctx_upload(ctx, pix_dev, point_pixel, (size_t)n_points * sizeof(int)))
                != SMVS_OK
            || (rc = ctx_upload(ctx, val_dev, point_depth, (size_t)n_points * sizeof(float)))
                != SMVS_OK)
            return rc;
        hipLaunchKernelGGL(surf_scatter_depth_kernel, dim3(blocks_for((size_t)n_points)),
            dim3(256), 0, ctx->stream, pix_dev, val_dev, n_points, ctx->surf_depth);
    } else {
        if (!ctx->sgm_resident || ctx->topo_sgm == nullptr) {
            set_error("smvs_surface_create: no depth map given and none resident "
                "(smvs_ctx_sgm_init_depth)");
            return SMVS_ERR_STATE;
        }
        hipLaunchKernelGGL(surf_clamp_depth_kernel, dim3(blocks_for(npix)), dim3(256), 0,
            ctx->stream, ctx->topo_sgm, ctx->surf_depth, npix);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    ctx->surf_depth_ok = true;
    if ((rc = ctx_ensure_grid(ctx, scale, g.npx, g.npy, g.start_x, g.start_y)) != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->node_valid, 0, (size_t)ctx->num_nodes, ctx->stream));
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->nodes, 0, (size_t)ctx->num_nodes * 4 * sizeof(double),
        ctx->stream));
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->patch_valid, 0, (size_t)ctx->num_patches, ctx->stream));
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->patch_vis, 0, (size_t)ctx->num_patches * sizeof(uint32_t),
        ctx->stream));
    SurfArgs A;
    fill_surf_args(ctx, &A);
    if ((rc = launch_fill_from_depth(ctx, A)) != SMVS_OK)
        return rc;
    ctx->has_surface = true;
    surface_changed(ctx);
    return read_counts(ctx, A, num_valid_patches, nullptr);
}

extern "C" int
smvs_surface_fill_patches_from_depth(smvs_ctx *ctx, int *num_valid_patches)
{
    int rc = require_surface(ctx, "smvs_surface_fill_patches_from_depth");
    if (rc != SMVS_OK)
        return rc;
    if (!ctx->surf_depth_ok) {
        set_error("smvs_surface_fill_patches_from_depth: the context holds no initial "
            "depth map (smvs_surface_create)");
        return SMVS_ERR_STATE;
    }
    SurfArgs A;
    fill_surf_args(ctx, &A);
    if ((rc = launch_fill_from_depth(ctx, A)) != SMVS_OK)
        return rc;
    surface_changed(ctx);
    return read_counts(ctx, A, num_valid_patches, nullptr);
}

extern "C" int
smvs_surface_subdivide(smvs_ctx *ctx, int *num_valid_patches)
{
    int rc = require_surface(ctx, "smvs_surface_subdivide");
    if (rc != SMVS_OK)
        return rc;
    SMVS_REQUIRE(ctx->scale >= 1, "the surface is at scale 0");
    Grid old;
    old.width = ctx->width; old.height = ctx->height;
    old.scale = ctx->scale; old.ps = ctx->patchsize;
    old.npx = ctx->npx; old.npy = ctx->npy;
    old.start_x = ctx->start_x; old.start_y = ctx->start_y;
    int off_x = 0, off_y = 0;
    Grid const g = smvs_surf::grid_subdivided(old, &off_x, &off_y);
    // the old surface moves to the scratch buffers (the grid buffers may be
    // reallocated for the finer grid)
    size_t const oldN = (size_t)ctx->num_nodes, oldP = (size_t)ctx->num_patches;
    if ((rc = ensure_tmp(ctx, oldN * 4, oldN + oldP)) != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->surf_tmp, ctx->nodes, oldN * 4 * sizeof(double),
        hipMemcpyDeviceToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->surf_tmp_bytes, ctx->node_valid, oldN,
        hipMemcpyDeviceToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->surf_tmp_bytes + oldN, ctx->patch_valid, oldP,
        hipMemcpyDeviceToDevice, ctx->stream));
    bool const had = ctx->has_surface;
    if ((rc = ctx_ensure_grid(ctx, g.scale, g.npx, g.npy, g.start_x, g.start_y)) != SMVS_OK)
        return rc;
    ctx->has_surface = had;
    SurfArgs A;
    fill_surf_args(ctx, &A);
    A.old_nodes = ctx->surf_tmp;
    A.old_node_valid = ctx->surf_tmp_bytes;
    A.old_patch_valid = ctx->surf_tmp_bytes + oldN;
    A.old_npx = old.npx;
    A.old_npy = old.npy;
    A.off_x = off_x;
    A.off_y = off_y;
    hipLaunchKernelGGL(surf_subdivide_kernel, dim3(blocks_for((size_t)A.num_nodes)),
        dim3(256), 0, ctx->stream, A);
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->patch_valid, 0, (size_t)ctx->num_patches, ctx->stream));
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->patch_vis, 0, (size_t)ctx->num_patches * sizeof(uint32_t),
        ctx->stream));
    launch_fill_holes(ctx, A);
    launch_remove_nodes(ctx, A);
    SMVS_HIP_CHECK(hipGetLastError());
    ctx->has_surface = true;
    surface_changed(ctx);
    return read_counts(ctx, A, num_valid_patches, nullptr);
}

extern "C" int
smvs_surface_expand(smvs_ctx *ctx, int *num_filled, int *num_valid_patches)
{
    int rc = require_surface(ctx, "smvs_surface_expand");
    if (rc != SMVS_OK)
        return rc;
    size_t const N = (size_t)ctx->num_nodes;
    if ((rc = ensure_tmp(ctx, N, N)) != SMVS_OK)
        return rc;
    SurfArgs A;
    fill_surf_args(ctx, &A);
    A.proposal = ctx->surf_tmp;
    A.proposed = ctx->surf_tmp_bytes;
    SMVS_HIP_CHECK(hipMemsetAsync(A.proposal, 0, N * sizeof(double), ctx->stream));
    SMVS_HIP_CHECK(hipMemsetAsync(A.proposed, 0, N, ctx->stream));
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->status + I_SURF_CHANGED, 0, sizeof(int), ctx->stream));
    for (int round = 0; round < 2; ++round) {
        hipLaunchKernelGGL(surf_expand_propose_kernel, dim3(blocks_for(N)), dim3(256), 0,
            ctx->stream, A);
        hipLaunchKernelGGL(surf_expand_apply_kernel, dim3(blocks_for(N)), dim3(256), 0,
            ctx->stream, A);
    }
    launch_fill_holes(ctx, A);
    launch_remove_nodes(ctx, A);
    SMVS_HIP_CHECK(hipGetLastError());
    surface_changed(ctx);
    return read_counts(ctx, A, num_valid_patches, num_filled);
}

extern "C" int
smvs_surface_remove_isolated_patches(smvs_ctx *ctx, int *num_valid_patches)
{
    int rc = require_surface(ctx, "smvs_surface_remove_isolated_patches");
    if (rc != SMVS_OK)
        return rc;
    SurfArgs A;
    fill_surf_args(ctx, &A);
    int const wpc = (A.g.npy + 2 + 31) / 32;
    // (the patches as they are and the deletions, one bit each)
    size_t const bytes = 2 * (size_t)(A.g.npx + 2) * wpc * sizeof(unsigned);
    unsigned *global_bits = nullptr;
    size_t lds = bytes;
    // The bit columns live in LDS when they fit (the kernel's static
    // __shared__ word counts against the CU's 160 KB too) and the device grants
    // that much dynamic LDS for this launch -- asked for per launch, only as
    // much as the grid needs; otherwise in global memory.
    if (bytes > (size_t)150 * 1024
        || allow_dynamic_lds(ctx->device,
               reinterpret_cast<const void *>(surf_isolated_kernel), bytes) != SMVS_OK) {
        if (ctx->surf_bits_cap < bytes) {
            ctx->surf_bits_cap = 0;
            if ((rc = device_alloc(&ctx->surf_bits, bytes / sizeof(unsigned))) != SMVS_OK)
                return rc;
            ctx->surf_bits_cap = bytes;
        }
        global_bits = ctx->surf_bits;
        lds = 0;
    }
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->status + I_SURF_CHANGED, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(surf_isolated_kernel, dim3(1), dim3(1024), lds, ctx->stream, A,
        global_bits, wpc);
    launch_remove_nodes(ctx, A);
    SMVS_HIP_CHECK(hipGetLastError());
    surface_changed(ctx);
    return read_counts(ctx, A, num_valid_patches, nullptr);
}

extern "C" int
smvs_surface_delete_unseen_patches(smvs_ctx *ctx, int *num_deleted, int *num_valid_patches)
{
    int rc = require_surface(ctx, "smvs_surface_delete_unseen_patches");
    if (rc != SMVS_OK)
        return rc;
    SurfArgs A;
    fill_surf_args(ctx, &A);
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->status + I_SURF_CHANGED, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(surf_delete_unseen_kernel, dim3(blocks_for((size_t)A.num_patches)),
        dim3(256), 0, ctx->stream, A, ctx->patch_vis);
    launch_remove_nodes(ctx, A);
    SMVS_HIP_CHECK(hipGetLastError());
    surface_changed(ctx);
    return read_counts(ctx, A, num_valid_patches, num_deleted);
}

extern "C" int
smvs_surface_delete_every(smvs_ctx *ctx, int every, int *num_valid_patches)
{
    int rc = require_surface(ctx, "smvs_surface_delete_every");
    if (rc != SMVS_OK)
        return rc;
    SMVS_REQUIRE(every >= 1, "every must be >= 1");
    SurfArgs A;
    fill_surf_args(ctx, &A);
    hipLaunchKernelGGL(surf_delete_every_kernel, dim3(1), dim3(1024), 0, ctx->stream, A, every);
    launch_remove_nodes(ctx, A);
    SMVS_HIP_CHECK(hipGetLastError());
    surface_changed(ctx);
    return read_counts(ctx, A, num_valid_patches, nullptr);
}

extern "C" int
smvs_surface_info(smvs_ctx *ctx, smvs_surface_geometry *out, int *num_valid_patches)
{
    int rc = require_surface(ctx, "smvs_surface_info");
    if (rc != SMVS_OK)
        return rc;
    if (out != nullptr) {
        out->scale = ctx->scale;
        out->patchsize = ctx->patchsize;
        out->npx = ctx->npx;
        out->npy = ctx->npy;
        out->start_x = ctx->start_x;
        out->start_y = ctx->start_y;
    }
    if (num_valid_patches == nullptr)
        return SMVS_OK;
    SurfArgs A;
    fill_surf_args(ctx, &A);
    return read_counts(ctx, A, num_valid_patches, nullptr);
}

extern "C" int
smvs_surface_download(smvs_ctx *ctx, double *nodes, uint8_t *node_valid,
    uint8_t *patch_valid, uint32_t *patch_vis)
{
    int rc = require_surface(ctx, "smvs_surface_download");
    if (rc != SMVS_OK)
        return rc;
    size_t const N = (size_t)ctx->num_nodes, P = (size_t)ctx->num_patches;
    if (nodes != nullptr
        && (rc = ctx_download(ctx, nodes, ctx->nodes, N * 4 * sizeof(double))) != SMVS_OK)
        return rc;
    if (node_valid != nullptr
        && (rc = ctx_download(ctx, node_valid, ctx->node_valid, N)) != SMVS_OK)
        return rc;
    if (patch_valid != nullptr
        && (rc = ctx_download(ctx, patch_valid, ctx->patch_valid, P)) != SMVS_OK)
        return rc;
    if (patch_vis != nullptr
        && (rc = ctx_download(ctx, patch_vis, ctx->patch_vis, P * sizeof(uint32_t))) != SMVS_OK)
        return rc;
    return SMVS_OK;
}
