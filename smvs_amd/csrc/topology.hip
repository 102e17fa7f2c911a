// Per-patch topology tests of DepthOptimizer between Newton batches
// (SURVEY.md row (f)-2), on the device:
//
//   * create_subview_surfaces (lib/depth_optimizer.cc:433-604): z-buffer of
//     the current surface (and of the SGM depth) in every neighbour, then per
//     (patch, neighbour) the image-border / occlusion test, the warp
//     anisotropy test and ncc_for_patch (:792-912) -> visibility bit mask;
//   * mse_for_patch (:747-790) for every valid patch, the quantity
//     cut_boundaries (:360-431) thresholds.
//
// The host keeps the topology itself (which patches and nodes exist): these
// kernels only produce the per-patch numbers the host decides on, and they do
// so with the arithmetic of csrc/host/topo_math.h (the source the C++ host
// mirror compiles too).  The z-buffer minimum is order independent:
// (float)min(d) == min((float)d) because rounding is monotone.
#include "common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "host/topo_math.h"

#include <vector>

namespace smvs_hip {

using smvs_topo::NccSample;
using smvs_topo::Warp;

// three packed floats (4-byte aligned): one global_load_dwordx3
struct __attribute__((packed, aligned(4))) float3_r { float x, y, z; };

// Several quotients over one denominator.  a / d as the compiler expands it is
// v_div_scale (twice), v_rcp_f64, two Newton steps on the reciprocal, the
// quotient a * r, ONE residual correction (v_div_fmas) and v_div_fixup: thirteen
// instructions, five of which depend only on d.  For a denominator well inside
// the normal range the scaling steps are the identity, so the same quotient --
// bit for bit, it is the same sequence of roundings -- comes from the shared
// refined reciprocal in three instructions plus the fix-up (zero, infinite and
// NaN numerators).  Any other denominator (zero, NaN, 1e-70 ...) is `plain ==
// false`: the caller divides.  The guarantee is conditional on the NUMERATOR
// as well: v_div_scale also rescales for numerators whose exponent is extreme
// (|a| beyond ~1e230 or denormal, quotients near overflow / underflow), and
// only the denominator is tested here.  The numerators of this file are
// projective coordinates and their pixel derivatives -- products of image
// coordinates (< 1e5), camera entries and depths, |a| < 1e20 and either exactly
// zero or > 1e-60 for any scene the surface tests accept -- so the condition
// holds; SMVS_TOPO_DIVIDE=exact runs the divisions themselves and the GPU
// suite demands identical masks.  The warp of a pixel divides ten times by d and
// d * d (Correspondence, topo_math.h): the visibility kernel issues a vector
// instruction every cycle it can, and a seventh of them were these.
struct SharedDivisor {
    double d, inv;
    bool plain;
    __device__ __forceinline__ explicit SharedDivisor(double d_, bool allow = true) : d(d_)
    {
        double const m = fabs(d_);
        plain = allow && m > 1e-70 && m < 1e70;   // (false for NaN)
        double r = __builtin_amdgcn_rcp(d_);
        r = __builtin_fma(__builtin_fma(-d_, r, 1.0), r, r);
        inv = __builtin_fma(__builtin_fma(-d_, r, 1.0), r, r);
    }
    // a / d, `plain` denominators only (no branch: straight-line callers)
    __device__ __forceinline__ double under(double a) const
    {
        double const q = a * inv;
        double const rem = __builtin_fma(-q, d, a);
        return __builtin_amdgcn_div_fixup(__builtin_fma(rem, inv, q), d, a);
    }
    // a / d for any denominator
    __device__ __forceinline__ double quotient(double a) const
    {
        return plain ? under(a) : a / d;
    }
};

// Warp::x, Warp::y and Warp::jacobian (topo_math.h; lib/correspondence.cc:20-51,
// 88-100): SHARED = the reciprocals of d and d * d computed once (only when
// plain()), otherwise the divisions of topo_math.h.  The same operations in
// the same order, every quotient the one the division gives.
template <bool SHARED>
struct WarpQuotients {
    SharedDivisor by_d, by_d2;
    __device__ __forceinline__ explicit WarpQuotients(Warp const &wp, bool allow = true)
        : by_d(wp.d, allow), by_d2(wp.d * wp.d, allow) {}
    __device__ __forceinline__ bool plain(void) const { return by_d.plain && by_d2.plain; }
    __device__ __forceinline__ double x(Warp const &wp) const
    {
        return SHARED ? by_d.under(wp.a) : wp.x();
    }
    __device__ __forceinline__ double y(Warp const &wp) const
    {
        return SHARED ? by_d.under(wp.b) : wp.y();
    }
    __device__ __forceinline__ void
    jacobian(Warp const &wp, const double *M, double w, double wx, double wy,
        double *jac) const
    {
#pragma clang fp contract(off)
        if (!SHARED) {
            wp.jacobian(M, w, wx, wy, jac);
            return;
        }
        jac[0] = by_d.under(wx * wp.p + w * M[0]) - by_d2.under(wp.a * (wx * wp.r + w * M[6]));
        jac[2] = by_d.under(wy * wp.p + w * M[1]) - by_d2.under(wp.a * (wy * wp.r + w * M[7]));
        jac[1] = by_d.under(wx * wp.q + w * M[3]) - by_d2.under(wp.b * (wx * wp.r + w * M[6]));
        jac[3] = by_d.under(wy * wp.q + w * M[4]) - by_d2.under(wp.b * (wy * wp.r + w * M[7]));
    }
};

// linear_at (topo_math.h) on both channels of a gradient plane: the taps once,
// four 8-byte loads, per channel linear_at's arithmetic term for term.
__device__ __forceinline__ void
linear_at_pair(const float2 *data, int w, int h, float x, float y, float *c0, float *c1)
{
#pragma clang fp contract(off)
    x = x < 0.0f ? 0.0f : (x > (float)(w - 1) ? (float)(w - 1) : x);
    y = y < 0.0f ? 0.0f : (y > (float)(h - 1) ? (float)(h - 1) : y);
    int const fx = (int)x, fy = (int)y;
    int const fx1 = fx + 1 < w - 1 ? fx + 1 : w - 1;
    int const fy1 = fy + 1 < h - 1 ? fy + 1 : h - 1;
    float const w1 = x - (float)fx, w0 = 1.0f - w1;
    float const w3 = y - (float)fy, w2 = 1.0f - w3;
    float2 const v00 = data[(long)fy * w + fx];
    float2 const v10 = data[(long)fy * w + fx1];
    float2 const v01 = data[(long)fy1 * w + fx];
    float2 const v11 = data[(long)fy1 * w + fx1];
    *c0 = v00.x * (w0 * w2) + v10.x * (w1 * w2) + v01.x * (w0 * w3) + v11.x * (w1 * w3);
    *c1 = v00.y * (w0 * w2) + v10.y * (w1 * w2) + v01.y * (w0 * w3) + v11.y * (w1 * w3);
}

struct TopoView {
    int w, h, c;
    const float *image;   // interleaved float image (bytes / 255)
};

struct TopoArgs {
    const double *nodes;
    const uint8_t *patch_valid;
    const uint32_t *patch_vis;      // input of the mse kernel
    uint32_t *vis_out;
    double *mse_out;
    const DeviceCameras *cams;
    TopoView views[1 + SMVS_MAX_SUBS];   // [0] main, [1 + j] neighbour j
    const float2 *main_grad;
    const SubPlanes *subs;
    float *zbuf[SMVS_MAX_SUBS];     // [(h + 1)][(w + 1)] z-buffer (3 x 3 splats)
    float *zraw[SMVS_MAX_SUBS];     // same shape: nearest depth per centre cell
    // zbuf holds the 5 x 5 minimum of zraw -- the minimum of the 3 x 3 z-buffer
    // cells the visibility test compares with -- instead of the 3 x 3 one
    // (topo_dilate5_kernel; SMVS_ZBUF_WINDOW=3: the z-buffer itself, nine lookups)
    int zbuf5;
    const float *sgm_depth;         // [H][W] or nullptr
    const NccSample *ncc;           // 32 concatenated templates
    int ncc_off[33];
    int W, H, npx, npy, stride, ps, start_x, start_y, n_subs, num_patches;
    int ps_log2;       // ps = 1 << ps_log2 (Surface: patchsize = 2^scale): shifts and an
                       // exact reciprocal instead of integer and double divisions
    double inv_ps;     // 1.0 / ps
    int use_ncc;
    // cut_boundaries
    uint8_t *patch_valid_rw;
    uint8_t *node_valid_rw;
    int *deleted;               // status word
    float invproj[9];
    int num_nodes;
    // cut_boundaries: nodes with more than one missing neighbour node (the
    // state before the pass), and whether the mse kernel may skip the patches
    // that touch none of them
    uint8_t *border_node;
    int only_candidates;
    int *mse_list;      // patches topo_mse_kernel evaluates
    int *mse_count;     // status word: entries of mse_list
    // create_subview_surfaces: depth and its pixel derivatives of the surface at
    // every pixel of a valid patch, [H][W][3] doubles (topo_pixel_surface_kernel)
    double *pix;
    // SMVS_TOPO_DIVIDE=exact: every quotient by the division itself
    // (SharedDivisor; the test that both give the same bits)
    int exact_divisions;
    // SMVS_NCC_PAIRS=0: the NCC samples of a lane one after the other
    int ncc_pairs;
    // the two halves of the visibility test as launches of their own
    // (topo_visibility_kernel<1> / <2>): what the geometric half decided per
    // (patch, neighbour), [num_patches][n_subs]
    uint8_t *pair_alive;
    // cut_boundaries passes enqueued ahead of their predecessor's count: a pass
    // whose predecessor deleted at most 10 patches does nothing
    // (`while (deleted > 10)`, depth_optimizer.cc:186-190)
    const int *pass_gate;
    // topo_visibility_kernel: lanes per (patch, neighbour) and the stash slots
    // per thread its launch reserves (vis_launch_shape)
    int vis_group, ncc_stash_slots;
    // ... and what else of its dynamic LDS is in use: doubles of the staged
    // depths (0: read from memory), entries of the staged interior template
    int lds_depth_doubles, lds_tpl_n;
    // topo_mse_kernel at patch sizes 32 / 64: chunks of 256 pixels per patch,
    // their partial sums [item][chunk][2] and arrival counters [item] (zero
    // between launches)
    int mse_chunks;
    double *mse_parts;
    int *mse_arrived;
};

__device__ __forceinline__ void
load_patch_nodes(TopoArgs const &A, int p, double n16[16])
{
    int const ix = p % A.npx, iy = p / A.npx;
    int const n00 = iy * A.stride + ix;
    int const ids[4] = { n00, n00 + 1, n00 + A.stride, n00 + A.stride + 1 };
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            n16[4 * n + k] = A.nodes[4 * (size_t)ids[n] + k];
}

// min over floats of any sign with integer atomics
__device__ __forceinline__ void
atomic_min_float(float *addr, float v)
{
    if (v >= 0.0f)
        atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else
        atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// the per-centre minima of every neighbour's z-buffer start at 10000
// (depth_optimizer.cc:441-446)
__global__ void __launch_bounds__(256)
topo_clear_kernel(TopoArgs A)
{
    int const s = blockIdx.z;
    TopoView const sv = A.views[1 + s];
    size_t const cells = (size_t)(sv.w + 1) * (sv.h + 1);
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells)
        A.zraw[s][i] = 10000.0f;
    // ... and the masks the visibility kernel ORs into start at zero (the first
    // neighbour's blocks; one runtime fill kernel per call less)
    if (s == 0)
        for (size_t p = i; p < (size_t)A.num_patches; p += (size_t)gridDim.x * blockDim.x)
            A.vis_out[p] = 0u;
}

// ---- z-buffer splat (depth_optimizer.cc:441-470), one thread per pixel ----
__global__ void __launch_bounds__(256)
topo_splat_kernel(TopoArgs A)
{
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= A.W || y >= A.H)
        return;
    float depths[2] = { 0.0f, 0.0f };
    // Surface::get_depth_map (surface.cc:155-168): float of the patch value
    int const gx = x - A.start_x, gy = y - A.start_y;
    if (gx >= 0 && gy >= 0 && gx < A.npx * A.ps && gy < A.npy * A.ps) {
        int const ix = gx >> A.ps_log2, iy = gy >> A.ps_log2;
        int const p = iy * A.npx + ix;
        if (A.patch_valid[p]) {
            double n16[16];
            load_patch_nodes(A, p, n16);
            int const i = gx - ix * A.ps, j = gy - iy * A.ps;
            depths[0] = (float)smvs_topo::patch_eval(n16, (i + 0.5) * A.inv_ps,
                (j + 0.5) * A.inv_ps, 0, 0);
        }
    }
    if (A.sgm_depth != nullptr)
        depths[1] = A.sgm_depth[(size_t)y * A.W + x];
    // Round 6: the two depths of a pixel (the surface's and the SGM map's) mostly
    // land in the same cell of a neighbour -- the surface starts as the SGM map --
    // and the L2 serves one atomic per clock and channel, 32 M of them per call
    // with SGM: where both centres agree ONE atomic carries the smaller depth (the
    // minimum is exact and order free, so the buffer is the same to the bit).
    bool const on[2] = { depths[0] != 0.0f, depths[1] != 0.0f };   // (NaN splats like the reference: no effect)
    if (!on[0] && !on[1])
        return;
    for (int s = 0; s < A.n_subs; ++s) {
        int const sw = A.views[1 + s].w, sh = A.views[1 + s].h;
        size_t cell[2] = { 0, 0 };
        float df[2] = { 0.0f, 0.0f };
        bool hit[2] = { false, false };
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!on[k])
                continue;
            double const w = depths[k];
            Warp wp(A.cams->M[s], A.cams->t[s], x + 0.5, y + 0.5, w);
            double const qx = wp.x() - 0.5, qy = wp.y() - 0.5;
            double const cutoffset = 3.0;
            if (qx < cutoffset || qx >= sw - cutoffset || qy < cutoffset
                || qy >= sh - cutoffset)
                continue;
            int const cx = (int)qx, cy = (int)qy;
            df[k] = (float)wp.d;
            if (!(df[k] == df[k]))
                continue;
            // The reference writes df into the 3 x 3 cells around (cx, cy)
            // (:455-462).  min is exact and order free, so the same buffer is
            // the 3 x 3 minimum filter of the per-centre minima: one atomic per
            // (pixel, neighbour) here instead of nine, the dilate kernel does
            // the rest.
            cell[k] = (size_t)cy * (sw + 1) + cx;
            hit[k] = true;
        }
        if (hit[0] && hit[1] && cell[0] == cell[1]) {
            atomic_min_float(A.zraw[s] + cell[0], fminf(df[0], df[1]));
        } else {
            if (hit[0])
                atomic_min_float(A.zraw[s] + cell[0], df[0]);
            if (hit[1])
                atomic_min_float(A.zraw[s] + cell[1], df[1]);
        }
    }
}

// zbuf = 3 x 3 minimum filter of zraw (cells outside the buffer do not exist).
// The minimum is separable and exact: a thread takes DILATE_ROWS rows of one
// column, forms the three-column minimum of the DILATE_ROWS + 2 rows it needs
// once and combines three of them per output: 4.5 loads per cell instead of 9
// (the kernel is bound by its cached loads).
constexpr int DILATE_ROWS = 4;
__global__ void __launch_bounds__(256)
topo_dilate_kernel(TopoArgs A)
{
    int const s = blockIdx.z;
    int const zw = A.views[1 + s].w + 1, zh = A.views[1 + s].h + 1;
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y0 = blockIdx.y * DILATE_ROWS;
    if (x >= zw || y0 >= zh)
        return;
    const float *raw = A.zraw[s];
    float rows[DILATE_ROWS + 2];
#pragma unroll
    for (int r = 0; r < DILATE_ROWS + 2; ++r) {
        int const yy = y0 - 1 + r;
        float m = 10000.0f;
        if (yy >= 0 && yy < zh) {
            const float *row = raw + (size_t)yy * zw;
            m = fminf(m, row[x]);
            if (x > 0)
                m = fminf(m, row[x - 1]);
            if (x + 1 < zw)
                m = fminf(m, row[x + 1]);
        }
        rows[r] = m;
    }
#pragma unroll
    for (int r = 0; r < DILATE_ROWS; ++r)
        if (y0 + r < zh)
            A.zbuf[s][(size_t)(y0 + r) * zw + x]
                = fminf(fminf(rows[r], rows[r + 1]), rows[r + 2]);
}

// The visibility test looks at the 3 x 3 z-buffer cells around a pixel's
// projection and fails when ANY of them is nearer than 0.95 of the pixel's depth
// (depth_optimizer.cc:792-830) -- i.e. when their MINIMUM is: a > b_i for some i
// <=> a > min b_i (no NaN is ever splatted).  The minimum of 3 x 3 cells of the
// 3 x 3 minimum filter is the 5 x 5 minimum filter of zraw, so this kernel leaves
// THAT in zbuf and the test is one lookup per (pixel, neighbour) instead of nine
// (round 6: 16 M x 9 four-byte loads per call at 1920 x 1080 x 8 were most of the
// vector-memory instructions of the visibility kernel's pixel pass).  Cells
// outside the buffer do not exist, as in the 3 x 3 filter; the test only looks
// at cells whose 3 x 3 neighbourhood is inside (its 3 % border).
// A workgroup loads DIL5_ROWS + 4 rows of 256 columns once (coalesced, into
// LDS), forms the five-column minima per row and the five-row minima of those:
// 252 x DIL5_ROWS cells per workgroup, 1.5 loads per cell.
constexpr int DIL5_ROWS = 8;
constexpr int DIL5_COLS = 252;
__global__ void __launch_bounds__(256)
topo_dilate5_kernel(TopoArgs A)
{
    __shared__ float tile[DIL5_ROWS + 4][256];
    int const s = blockIdx.z;
    int const zw = A.views[1 + s].w + 1, zh = A.views[1 + s].h + 1;
    int const t = (int)threadIdx.x;
    int const x0 = (int)blockIdx.x * DIL5_COLS;        // first output column; tile column j is x0 - 2 + j
    int const y0 = (int)blockIdx.y * DIL5_ROWS;
    if (x0 >= zw || y0 >= zh)
        return;
    const float *raw = A.zraw[s];
    int const gx = x0 - 2 + t;
    bool const col_ok = gx >= 0 && gx < zw;
#pragma unroll
    for (int r = 0; r < DIL5_ROWS + 4; ++r) {
        int const gy = y0 - 2 + r;
        // (a cell that does not exist takes no part in a minimum: +inf)
        tile[r][t] = col_ok && gy >= 0 && gy < zh ? raw[(size_t)gy * zw + gx] : __builtin_inff();
    }
    __syncthreads();
    if (t < 2 || t >= 2 + DIL5_COLS || gx >= zw)
        return;
    float rows[DIL5_ROWS + 4];
#pragma unroll
    for (int r = 0; r < DIL5_ROWS + 4; ++r)
        rows[r] = fminf(fminf(fminf(tile[r][t - 2], tile[r][t - 1]), tile[r][t]),
            fminf(tile[r][t + 1], tile[r][t + 2]));
#pragma unroll
    for (int r = 0; r < DIL5_ROWS; ++r)
        if (y0 + r < zh)
            A.zbuf[s][(size_t)(y0 + r) * zw + gx] = fminf(fminf(fminf(rows[r], rows[r + 1]),
                rows[r + 2]), fminf(rows[r + 3], rows[r + 4]));
}

// A group of G = min(64, ps^2) consecutive lanes works on one (patch,
// neighbour) resp. one patch: the pixels / samples are dealt round-robin to
// the lanes and the group combines its partial results with xor-shuffles in a
// fixed order (deterministic).  The reference's loops are sequential; every
// quantity here is a conjunction, a maximum or a sum, so only the summation
// order differs (by rounding, far below the 0.05 / 8.0 / 0.0 thresholds the
// results are compared with).
__device__ __host__ __forceinline__ int
group_size(int ps, int whole_workgroup_from)
{
    // ps is a power of two: 1, 4, 16, 64 lanes for ps = 1, 2, 4, 8 and above.
    // From ps = whole_workgroup_from on the whole 256-thread workgroup works
    // on one item: at the coarse scales a few hundred patches of thousands of
    // pixels each are a latency chain per lane, not a throughput problem
    // (measured, 341 patches at scale 6: mse 554 -> 190 us, visibility
    // 705 -> 585 us; at ps = 16 the barriers of the workgroup-wide reductions
    // cost the visibility kernel more than the shorter chains save:
    // 840 -> 1640 us, so it switches at 64, the mse kernel at 16).
    // (Round 6: the visibility kernel takes this rule only for ps = 1, 32, 64
    // and up; in between it runs 2 / 4 / 8 / 32 lanes at ps = 2 / 4 / 8 / 16,
    // smvs_topology_subviews.)
    int const pp = ps * ps;
    if (ps >= whole_workgroup_from)
        return 256;
    return pp >= 64 ? 64 : pp;
}
constexpr int VIS_WORKGROUP_FROM = 64;   // topo_visibility_kernel
// samples of ncc_for_patch a lane keeps in registers between the two passes;
// the following NCC_STASH_MAX live in LDS (48 KB per workgroup at most: three
// workgroups per CU, what the kernel's registers allow), any beyond are
// recomputed
constexpr int NCC_KEEP = 4;
constexpr int NCC_STASH_MAX = 16;
constexpr int MSE_WORKGROUP_FROM = 16;   // topo_mse_kernel

// Reductions over a lane group (the patch-MSE kernel's; the visibility
// kernel's are vis_lanes_reduce below).  G <= 64: xor-shuffles inside the wave.
// G == 256: the workgroup is the group -- per-wave results meet in LDS (every
// thread of the workgroup must call; `red` holds 4 doubles).
template <typename T>
__device__ __forceinline__ T
group_sum(T v, int G, double *red)
{
    for (int off = (G < 64 ? G : 64) >> 1; off > 0; off >>= 1)
        v += __shfl_xor(v, off);
    if (G > 64) {
        __syncthreads();   // (the previous reduction's readers are done)
        if ((threadIdx.x & 63) == 0)
            red[threadIdx.x >> 6] = (double)v;
        __syncthreads();
        v = (T)(((red[0] + red[1]) + red[2]) + red[3]);
    }
    return v;
}

__device__ __forceinline__ bool
group_all(bool ok, int G, int lane, double *red)
{
    unsigned long long const b = __ballot(ok);
    if (G > 64)
        return __syncthreads_and(b == ~0ull) != 0;
    (void)red;
    unsigned long long const gmask = G >= 64 ? ~0ull
        : (((1ull << G) - 1ull) << ((lane / G) * G));
    return (b & gmask) == gmask;
}

// ---- lane-group reductions of the visibility kernel on the VALU (round 6) ----
// __shfl_xor of a double is two ds_bpermute_b32 through the CU's one LDS pipe
// and a round trip per step; the kernel's eleven reductions of up to six steps
// each were ~5 us of every wave's life at patch size 8, more than its samples
// (wave life = 5.0 us + 1.9 us per pixel / sample slot, from the per-call times
// of a --no-sgm view).  DPP moves inside a row of 16 lanes and gfx950's
// v_permlane16_swap / v_permlane32_swap across the rows do the same butterflies
// without LDS.  Every lane of the group gets the result, bit-identical in all
// of them (each step adds / compares the same two numbers in both lanes of a
// pair).
template <int CTRL>
__device__ __forceinline__ double
vis_dpp(double v)
{
    unsigned long long const b = (unsigned long long)__double_as_longlong(v);
    int const lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    int const hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf,
        false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32)
        | (unsigned)lo));
}

// the partner's value across rows (HALF = 16: rows 2k <-> 2k + 1) or halves of
// the wave (HALF = 32): both values of the pair, lower lane's first
template <int HALF>
__device__ __forceinline__ void
vis_swap(double v, double &lower, double &upper)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    unsigned long long const b = (unsigned long long)__double_as_longlong(v);
    u2 lo, hi;
    if constexpr (HALF == 32) {
        lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
        hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false,
            false);
    } else {
        lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
        hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false,
            false);
    }
    lower = __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x));
    upper = __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
}

// op over the min(G, 64) lanes of a group inside the wave (G a power of two)
template <typename Op>
__device__ __forceinline__ double
vis_lanes_reduce(double v, int G, Op const &op)
{
    if (G >= 2)
        v = op(v, vis_dpp<0xB1>(v));      // quad_perm [1, 0, 3, 2]
    if (G >= 4)
        v = op(v, vis_dpp<0x4E>(v));      // quad_perm [2, 3, 0, 1]
    if (G >= 8)
        v = op(v, vis_dpp<0x141>(v));     // row_half_mirror
    if (G >= 16)
        v = op(v, vis_dpp<0x140>(v));     // row_mirror
    if (G >= 32) {
        double a, b;
        vis_swap<16>(v, a, b);
        v = op(a, b);
    }
    if (G >= 64) {
        double a, b;
        vis_swap<32>(v, a, b);
        v = op(a, b);
    }
    return v;
}

// K sums over the group at once; G == 256: the per-wave sums of all K meet in
// LDS behind ONE pair of barriers (`red` holds K x 4 doubles; every thread of
// the workgroup must call)
template <int K>
__device__ __forceinline__ void
vis_group_sums(double (&v)[K], int G, double *red)
{
    auto const add = [](double a, double b) { return a + b; };
#pragma unroll
    for (int k = 0; k < K; ++k)
        v[k] = vis_lanes_reduce(v[k], G, add);
    if (G > 64) {
        __syncthreads();   // (the previous reduction's readers are done)
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int k = 0; k < K; ++k)
                red[k * 4 + (threadIdx.x >> 6)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k)
            v[k] = ((red[k * 4] + red[k * 4 + 1]) + red[k * 4 + 2]) + red[k * 4 + 3];
    }
}

__device__ __forceinline__ double
vis_group_max(double v, int G, double *red)
{
    auto const larger = [](double a, double b) { return a < b ? b : a; };
    v = vis_lanes_reduce(v, G, larger);
    if (G > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
            red[threadIdx.x >> 6] = v;
        __syncthreads();
        for (int w = 0; w < 4; ++w)
            v = v < red[w] ? red[w] : v;
    }
    return v;
}

// ---- the surface at every pixel of every valid patch: depth w and its pixel
// derivatives wx, wy.  The visibility kernel needs them per (pixel, neighbour)
// and, for the NCC samples, per (sample, neighbour): evaluated here ONCE per
// pixel with the expressions that kernel used per neighbour (patch_eval of
// topo_math.h: the same bits), 8 x less bicubic arithmetic for 8 neighbours.
__global__ void __launch_bounds__(256)
topo_pixel_surface_kernel(TopoArgs A)
{
#pragma clang fp contract(off)
    int const pp = A.ps * A.ps;
    long long const gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int const p = (int)(gid >> (2 * A.ps_log2));
    int const k = (int)(gid & (pp - 1));
    if (p >= A.num_patches || !A.patch_valid[p])
        return;
    double n16[16];
    load_patch_nodes(A, p, n16);
    int const i = k & (A.ps - 1), j = k >> A.ps_log2;
    // (x / ps == x * (1 / ps) exactly: ps is a power of two)
    double const u = (i + 0.5) * A.inv_ps, v = (j + 0.5) * A.inv_ps;
    int const x = A.start_x + (p % A.npx) * A.ps + i;
    int const y = A.start_y + (p / A.npx) * A.ps + j;
    double *out = A.pix + ((size_t)y * A.W + x) * 3;
    out[0] = smvs_topo::patch_eval(n16, u, v, 0, 0);
    out[1] = smvs_topo::patch_eval(n16, u, v, 1, 0) * A.inv_ps;
    out[2] = smvs_topo::patch_eval(n16, u, v, 0, 1) * A.inv_ps;
}

// ---- visibility of every patch in every neighbour (:472-590), incl.
// ncc_for_patch (:792-912) ----
// (153 VGPRs: three waves per SIMD -- round 5: 159, round 4: 192 and two waves.
// Launch bounds that force 128 VGPRs and four waves put 116-128 bytes per lane
// into scratch: 610 -> 762 us when it was measured in round 5.)
// Round 6 (profiles/r6_visibility_groups.txt): the group's reductions on the
// VALU (vis_lanes_reduce), the lanes per (patch, neighbour) chosen by patch size
// on the host (A.vis_group), the NCC's warped colours beyond the kept ones in an
// LDS stash, a sample's depth and template entry from LDS, the neighbour
// wave-uniform (blockIdx.y): 9.1 -> 5.8 ms per --no-sgm view, masks unchanged.
//
// PART 0: the whole test in one launch (the default).  SMVS_VIS_SPLIT=1
// (round 6, an experiment that did not pay: 3 % slower) runs the two halves
// as launches of their own with the NCC on, PART 1 = the pass over the patch's pixels (borders,
// z-buffer, warp anisotropy) leaving its verdict per (patch, neighbour) in
// `pair_alive`, PART 2 = ncc_for_patch for the pairs that are still alive:
// the same statements in the same order (one body, `if constexpr`), so the
// masks are those of PART 0 bit for bit -- but each half is compiled for its
// own registers: the waves of the fused kernel waited for memory half of
// their life at three per SIMD (profiles/r5_visibility_counters.txt), and the
// NCC half without the Jacobian's live range fits more of them.
template <int PART>
__global__ void __launch_bounds__(256, 2)
topo_visibility_kernel(TopoArgs A)
{
#pragma clang fp contract(off)
    __shared__ double red[6 * 4];
    // Dynamic LDS, sized by the host (vis launch shape in smvs_topology_subviews):
    //  * the depths ncc_for_patch's samples take -- the surface at the patch's
    //    pixels and its four corner nodes, [group of the workgroup][ps^2 + 4]
    //    doubles, left there by the pass over the pixels;
    //  * the sample template of an interior patch (all five border predicates),
    //    two ints per entry;
    //  * the warped colours of the samples a lane does not keep in registers,
    //    [slot][channel][thread] floats.
    // A sample was three dependent round trips to memory (template entry ->
    // depth -> taps) in a kernel whose waves wait for memory half of their life;
    // with the first two in LDS it is one.
    extern __shared__ double vis_lds[];
    double *const lds_depth = vis_lds;
    int *const lds_tpl = reinterpret_cast<int *>(vis_lds + A.lds_depth_doubles);
    float *const ncc_stash = reinterpret_cast<float *>(lds_tpl + 2 * A.lds_tpl_n);
    int const ps = A.ps;
    int const G = A.vis_group;
    int const lane = threadIdx.x & 63;
    int const g_log2 = 31 - __clz(G);           // G = 1 << g_log2
    int const gl = threadIdx.x & (G - 1);     // lane inside the group
    int const dstride = ps * ps + 4;
    // (PART 2 does not walk the pixels)
    bool const depth_in_lds = PART != 2 && A.lds_depth_doubles > 0;
    double *const my_depths = lds_depth + (threadIdx.x >> g_log2) * dstride;
    if (PART != 1 && A.lds_tpl_n > 0) {
        const NccSample *src = A.ncc + A.ncc_off[31];
        for (int i = threadIdx.x; i < A.lds_tpl_n; i += 256) {
            NccSample const e = src[i];
            lds_tpl[2 * i] = (int)((unsigned)(unsigned short)e.dx | ((unsigned)(unsigned short)e.dy << 16));
            lds_tpl[2 * i + 1] = e.src;
        }
        __syncthreads();
    }
    // (group index < num_patches * n_subs: 32 bits)
    // (Round 6 measured two other orders of the groups, because the kernel
    // fetches 1,007 MB per call at 1920 x 1080 for ~340 MB of planes
    // (profiles/r6_hbm_traffic.txt): neighbour-major -- the groups in flight read
    // ONE neighbour's image -- and the workgroups dealt to the XCDs in contiguous
    // bands of the patch grid, as the patch kernel's are.  Neither changed the
    // traffic (1,007 MB) or the time (610 / 623 against 605-611 us): the fetches are
    // 12-byte taps and 4-byte z-buffer cells out of 128-byte lines, not lines
    // fetched by several XCDs.  Plain order.)
    // The neighbour is the workgroup's (blockIdx.y): its camera, image size and
    // pointers are wave-uniform -- scalar registers and scalar loads, operands
    // of the vector arithmetic instead of 30 vector registers of every lane.
    int const s = (int)blockIdx.y;
    int const p = (int)(((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2);
    // (index of the pair: < num_patches * n_subs, 32 bits)
    unsigned const gid = (unsigned)p * (unsigned)A.n_subs + (unsigned)s;
    bool alive = p < A.num_patches && A.patch_valid[p];
    int const pc = alive ? p : 0;
    int const px = A.start_x + (pc % A.npx) * ps;
    int const py = A.start_y + (pc / A.npx) * ps;
    const double *M = A.cams->M[s];
    const double *t = A.cams->t[s];
    TopoView const mv = A.views[0], sv = A.views[1 + s];
    double const sw = sv.w, sh = sv.h;
    int const zw = sv.w + 1;
    double const cutoffset = 0.03 * (sw < sh ? sh : sw);
    const float *zbuf = A.zbuf[s];

    // border / occlusion test and the warp anisotropy, one pass over the
    // patch's pixels
    bool visible = true;
    double worst = 0.0;
    // (one pixel with either kind of quotients; false: outside the neighbour's
    // image, the reference stops looking at the patch)
    auto const pixel = [&](auto const &wq, Warp const &wp, const double *sp, double w) -> bool {
        double const qx = wq.x(wp) - 0.5, qy = wq.y(wp) - 0.5;
        if (qx < cutoffset || qx >= sw - cutoffset || qy < cutoffset
            || qy >= sh - cutoffset) {
            visible = false;
            return false;
        }
        int const cx = (int)qx, cy = (int)qy;
        if (A.zbuf5) {
            // the minimum of the nine cells, formed once per cell (topo_dilate5_kernel)
            if (wp.d * 0.95 > zbuf[(unsigned)cy * (unsigned)zw + (unsigned)cx])
                visible = false;
        } else {
            for (int dx = -1; dx < 2; ++dx)
                for (int dy = -1; dy < 2; ++dy)
                    if (wp.d * 0.95 > zbuf[(unsigned)(cy + dy) * (unsigned)zw + (unsigned)(cx + dx)])
                        visible = false;
        }
        // ratio of the squared singular values of the warp Jacobian
        double const wx = sp[1], wy = sp[2];
        double jac[4];
        wq.jacobian(wp, M, w, wx, wy, jac);
        double const e = sqrt((jac[0] - jac[3]) * (jac[0] - jac[3])
            + (jac[1] + jac[2]) * (jac[1] + jac[2]));
        double const g = sqrt((jac[0] + jac[3]) * (jac[0] + jac[3])
            + (jac[1] - jac[2]) * (jac[1] - jac[2]));
        double const s0 = (e + g) / 2.0;
        double const s1 = fabs(s0 - e);
        double const hi = s0 < s1 ? s1 : s0, lo = s1 < s0 ? s1 : s0;
        double const ratio = (hi * hi) / (lo * lo);
        // std::max(worst, ratio): a NaN ratio leaves worst unchanged
        worst = worst < ratio ? ratio : worst;
        return true;
    };
    if (PART != 2 && alive) {
        if (depth_in_lds) {
            int const n00 = (pc / A.npx) * A.stride + pc % A.npx;
            for (int c = gl; c < 4; c += G)
                my_depths[ps * ps + c] = A.nodes[4 * (size_t)(n00 + (c & 1) + (c >> 1) * A.stride)];
        }
        for (int k = gl; k < ps * ps; k += G) {
            int const i = k & (ps - 1), j = k >> A.ps_log2;
            // depth and pixel derivatives of the surface (topo_pixel_surface_kernel)
            const double *sp = A.pix + ((unsigned)(py + j) * (unsigned)A.W + (unsigned)(px + i)) * 3u;
            double const w = sp[0];
            if (depth_in_lds)
                my_depths[k] = w;
            Warp wp(M, t, px + i + 0.5, py + j + 0.5, w);
            WarpQuotients<true> const wq(wp, A.exact_divisions == 0);
            bool const go_on = wq.plain() ? pixel(wq, wp, sp, w)
                : pixel(WarpQuotients<false>(wp), wp, sp, w);
            if (!go_on)
                break;
        }
    }
    // (the depths are read by the other lanes of the group: LDS operations of a
    // wave complete in order, the fences keep the compiler from moving them; a
    // group of 256 meets in group_all's barrier below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (PART != 2) {
        visible = group_all(visible, G, lane, red);
        worst = vis_group_max(worst, G, red);
        alive = alive && visible && !(worst > 8.0);
    } else {
        // (the verdict of the geometric half: the whole group reads one byte)
        alive = alive && A.pair_alive[gid] != 0;   // (alive implies gid in range)
    }
    if constexpr (PART == 1) {
        if (gl == 0 && p < A.num_patches)
            A.pair_alive[gid] = alive ? 1 : 0;
        return;
    }

    // ncc_for_patch
    double ncc = 1.0;
    if (PART != 1 && A.use_ncc) {
        int const flags = smvs_topo::ncc_flags(px, py, ps, mv.w, mv.h);
        const NccSample *tpl = A.ncc + A.ncc_off[flags];
        int const n = A.ncc_off[flags + 1] - A.ncc_off[flags];
        bool const tpl_in_lds = A.lds_tpl_n > 0 && flags == 31;
        auto const sample_at = [&](int i) -> NccSample {
            if (tpl_in_lds) {
                int const a = lds_tpl[2 * i], b = lds_tpl[2 * i + 1];
                return NccSample{ (short)(a & 0xffff), (short)(a >> 16), (short)b };
            }
            return tpl[i];
        };
        bool inside = true;
        double sum0[3] = { 0, 0, 0 }, sum1[3] = { 0, 0, 0 };
        double mean0[3], mean1[3];
        double n0 = 0.0, n1 = 0.0, dot = 0.0;
        // The colours of a sample, main view and warped neighbour.  The second
        // pass (centred products) needs the same values as the first (sums):
        // a lane keeps its first NCC_KEEP samples in registers -- all of them
        // at the fine scales, where the border samples make the templates
        // 2 - 3 x the patch -- and recomputes the rest.
        // (float: the kept values ARE floats -- an image value, linear_at's
        // float result -- widened when they are used)
        float keep_m[NCC_KEEP][3], keep_s[NCC_KEEP][3];
        // Round 6: what does not fit the registers goes to LDS instead of being
        // recomputed in the second pass -- at patch sizes 32 and 64 a lane has
        // ~18 samples, so the second pass warped and interpolated 14 of them
        // again (only the thread itself reads its slots: no barrier)
        auto const stash_put = [&](int slot, float const (&cs)[3]) {
            if (slot - NCC_KEEP < A.ncc_stash_slots)
                for (int c = 0; c < 3; ++c)
                    ncc_stash[((slot - NCC_KEEP) * 3 + c) * 256 + threadIdx.x] = cs[c];
        };
        auto colours = [&](int i, double (&cm)[3], double (&cs)[3], bool check) -> bool {
            NccSample const smp = sample_at(i);
            // the depth of grid sample src is the surface at that pixel; the
            // corner samples take the corner node's depth (:803-857)
            double depth;
            if (depth_in_lds) {
                depth = my_depths[smp.src >= 0 ? smp.src : ps * ps - 1 - smp.src];
            } else if (smp.src >= 0) {
                // (32-bit offsets: the host checks that the planes have fewer than
                // 2^31 elements; 64-bit multiply-adds run at a quarter of the rate)
                depth = A.pix[((unsigned)(py + (smp.src >> A.ps_log2)) * (unsigned)A.W
                    + (unsigned)(px + (smp.src & (ps - 1)))) * 3u];
            } else {
                int const corner = -1 - smp.src;
                int const n00 = (pc / A.npx) * A.stride + pc % A.npx;
                depth = A.nodes[4 * (size_t)(n00 + (corner & 1) + (corner >> 1) * A.stride)];
            }
            double const sx = (double)(px + smp.dx);
            double const sy = (double)(py + smp.dy);
            Warp wp(M, t, sx + 0.5, sy + 0.5, depth);
            SharedDivisor const by_d(wp.d, A.exact_divisions == 0);
            double const qx = by_d.quotient(wp.a) - 0.5, qy = by_d.quotient(wp.b) - 0.5;
            if (check && (qx < 1 || qx > sv.w - 2 || qy < 1 || qy > sv.h - 2))
                return false;
            if (mv.c == 3 && sv.c == 3) {
                // RGB views: the three channels of a tap lie side by side, so a
                // sample is 1 + 4 twelve-byte loads instead of 3 + 12 four-byte
                // ones; per channel the arithmetic is linear_at's
                // (topo_math.h), term for term
                float3_r const m3 = *reinterpret_cast<const float3_r *>(mv.image
                    + ((unsigned)(py + smp.dy) * (unsigned)mv.w + (unsigned)(px + smp.dx)) * 3u);
                cm[0] = m3.x; cm[1] = m3.y; cm[2] = m3.z;
                float x = (float)qx, y = (float)qy;
                x = x < 0.0f ? 0.0f : (x > (float)(sv.w - 1) ? (float)(sv.w - 1) : x);
                y = y < 0.0f ? 0.0f : (y > (float)(sv.h - 1) ? (float)(sv.h - 1) : y);
                int const fx = (int)x, fy = (int)y;
                int const fx1 = fx + 1 < sv.w - 1 ? fx + 1 : sv.w - 1;
                int const fy1 = fy + 1 < sv.h - 1 ? fy + 1 : sv.h - 1;
                float const w1 = x - (float)fx, w0 = 1.0f - w1;
                float const w3 = y - (float)fy, w2 = 1.0f - w3;
                float const k00 = w0 * w2, k10 = w1 * w2, k01 = w0 * w3, k11 = w1 * w3;
                const float *img = sv.image;
                unsigned const row0 = (unsigned)fy * (unsigned)sv.w, row1 = (unsigned)fy1 * (unsigned)sv.w;
                float3_r const v00 = *reinterpret_cast<const float3_r *>(img + (row0 + (unsigned)fx) * 3u);
                float3_r const v10 = *reinterpret_cast<const float3_r *>(img + (row0 + (unsigned)fx1) * 3u);
                float3_r const v01 = *reinterpret_cast<const float3_r *>(img + (row1 + (unsigned)fx) * 3u);
                float3_r const v11 = *reinterpret_cast<const float3_r *>(img + (row1 + (unsigned)fx1) * 3u);
                cs[0] = v00.x * k00 + v10.x * k10 + v01.x * k01 + v11.x * k11;
                cs[1] = v00.y * k00 + v10.y * k10 + v01.y * k01 + v11.y * k11;
                cs[2] = v00.z * k00 + v10.z * k10 + v01.z * k01 + v11.z * k11;
                return true;
            }
            for (int c = 0; c < 3; ++c) {
                int const cmi = c < mv.c - 1 ? c : mv.c - 1;
                int const csi = c < sv.c - 1 ? c : sv.c - 1;
                cm[c] = mv.image[((size_t)(py + smp.dy) * mv.w + (px + smp.dx)) * mv.c + cmi];
                cs[c] = smvs_topo::linear_at(sv.image, sv.w, sv.h, sv.c, (float)qx,
                    (float)qy, csi);
            }
            return true;
        };
        // Two samples of an RGB pair of views side by side: both template
        // entries, then both depths, both warps, the nine taps of both, the
        // arithmetic.  A lane's samples were three chains of three dependent
        // round trips each (template entry -> depth -> taps), one after the
        // other: the waves of this kernel wait for memory more than half of
        // their life (profiles/r5_visibility_counters.txt).  Values and the
        // order they are summed in are those of `colours`; a sample outside the
        // neighbour's image reads clamped taps (the sums of such a patch are
        // never used: ncc = -1).
        bool const rgb = mv.c == 3 && sv.c == 3;
        struct NccTaps { unsigned o00, o10, o01, o11; float k00, k10, k01, k11; };
        auto const taps_at = [&](double qx, double qy) -> NccTaps {
            float x = (float)qx, y = (float)qy;
            x = x == x ? x : 0.0f;   // (outside anyway; keeps the conversion defined)
            y = y == y ? y : 0.0f;
            x = x < 0.0f ? 0.0f : (x > (float)(sv.w - 1) ? (float)(sv.w - 1) : x);
            y = y < 0.0f ? 0.0f : (y > (float)(sv.h - 1) ? (float)(sv.h - 1) : y);
            int const fx = (int)x, fy = (int)y;
            int const fx1 = fx + 1 < sv.w - 1 ? fx + 1 : sv.w - 1;
            int const fy1 = fy + 1 < sv.h - 1 ? fy + 1 : sv.h - 1;
            float const w1 = x - (float)fx, w0 = 1.0f - w1;
            float const w3 = y - (float)fy, w2 = 1.0f - w3;
            unsigned const row0 = (unsigned)fy * (unsigned)sv.w, row1 = (unsigned)fy1 * (unsigned)sv.w;
            NccTaps tp;
            tp.o00 = (row0 + (unsigned)fx) * 3u;  tp.o10 = (row0 + (unsigned)fx1) * 3u;
            tp.o01 = (row1 + (unsigned)fx) * 3u;  tp.o11 = (row1 + (unsigned)fx1) * 3u;
            tp.k00 = w0 * w2; tp.k10 = w1 * w2; tp.k01 = w0 * w3; tp.k11 = w1 * w3;
            return tp;
        };
        auto const depth_of = [&](NccSample const &smp) -> const double * {
            int const corner = -1 - smp.src;
            int const n00 = (pc / A.npx) * A.stride + pc % A.npx;
            const double *at_pixel = A.pix + ((unsigned)(py + (smp.src >> A.ps_log2)) * (unsigned)A.W
                + (unsigned)(px + (smp.src & (ps - 1)))) * 3u;
            const double *at_node = A.nodes + 4 * (size_t)(n00 + (corner & 1)
                + (corner >> 1) * A.stride);
            return smp.src >= 0 ? at_pixel : at_node;
        };
        auto const pair = [&](int ia, int ib, float (&ma)[3], float (&sa)[3], bool &oka,
                float (&mb)[3], float (&sb)[3], bool &okb) {
            NccSample const a = sample_at(ia), b = sample_at(ib);
            double da, db;
            if (depth_in_lds) {
                da = my_depths[a.src >= 0 ? a.src : ps * ps - 1 - a.src];
                db = my_depths[b.src >= 0 ? b.src : ps * ps - 1 - b.src];
            } else {
                const double *pa = depth_of(a), *pb = depth_of(b);
                da = *pa;
                db = *pb;
            }
            // (the scheduler would sink the second sample's loads below the first
            // one's arithmetic to save registers: both are asked for first)
            __builtin_amdgcn_sched_barrier(0);
            Warp const wa(M, t, (double)(px + a.dx) + 0.5, (double)(py + a.dy) + 0.5, da);
            Warp const wb(M, t, (double)(px + b.dx) + 0.5, (double)(py + b.dy) + 0.5, db);
            SharedDivisor const qa(wa.d, A.exact_divisions == 0), qb(wb.d, A.exact_divisions == 0);
            double ax, ay, bx, by;
            if (qa.plain && qb.plain) {
                ax = qa.under(wa.a) - 0.5;  ay = qa.under(wa.b) - 0.5;
                bx = qb.under(wb.a) - 0.5;  by = qb.under(wb.b) - 0.5;
            } else {
                ax = wa.x() - 0.5;  ay = wa.y() - 0.5;
                bx = wb.x() - 0.5;  by = wb.y() - 0.5;
            }
            oka = !(ax < 1 || ax > sv.w - 2 || ay < 1 || ay > sv.h - 2);
            okb = !(bx < 1 || bx > sv.w - 2 || by < 1 || by > sv.h - 2);
            NccTaps const ta = taps_at(ax, ay), tb = taps_at(bx, by);
            const float *img = sv.image;
            float3_r const am = *reinterpret_cast<const float3_r *>(mv.image
                + ((unsigned)(py + a.dy) * (unsigned)mv.w + (unsigned)(px + a.dx)) * 3u);
            float3_r const bm = *reinterpret_cast<const float3_r *>(mv.image
                + ((unsigned)(py + b.dy) * (unsigned)mv.w + (unsigned)(px + b.dx)) * 3u);
            float3_r const a00 = *reinterpret_cast<const float3_r *>(img + ta.o00);
            float3_r const a10 = *reinterpret_cast<const float3_r *>(img + ta.o10);
            float3_r const a01 = *reinterpret_cast<const float3_r *>(img + ta.o01);
            float3_r const a11 = *reinterpret_cast<const float3_r *>(img + ta.o11);
            float3_r const b00 = *reinterpret_cast<const float3_r *>(img + tb.o00);
            float3_r const b10 = *reinterpret_cast<const float3_r *>(img + tb.o10);
            float3_r const b01 = *reinterpret_cast<const float3_r *>(img + tb.o01);
            float3_r const b11 = *reinterpret_cast<const float3_r *>(img + tb.o11);
            __builtin_amdgcn_sched_barrier(0);
            ma[0] = am.x; ma[1] = am.y; ma[2] = am.z;
            mb[0] = bm.x; mb[1] = bm.y; mb[2] = bm.z;
            sa[0] = a00.x * ta.k00 + a10.x * ta.k10 + a01.x * ta.k01 + a11.x * ta.k11;
            sa[1] = a00.y * ta.k00 + a10.y * ta.k10 + a01.y * ta.k01 + a11.y * ta.k11;
            sa[2] = a00.z * ta.k00 + a10.z * ta.k10 + a01.z * ta.k01 + a11.z * ta.k11;
            sb[0] = b00.x * tb.k00 + b10.x * tb.k10 + b01.x * tb.k01 + b11.x * tb.k11;
            sb[1] = b00.y * tb.k00 + b10.y * tb.k10 + b01.y * tb.k01 + b11.y * tb.k11;
            sb[2] = b00.z * tb.k00 + b10.z * tb.k10 + b01.z * tb.k01 + b11.z * tb.k11;
        };
        if (alive && rgb && A.ncc_pairs != 0) {
            // pass 0 of the loop below, two samples at a time
            int slot = 0;
            for (int i = gl; i < n; i += 2 * G, slot += 2) {
                int const i2 = i + G;
                bool const two = i2 < n;
                float ma[3], sa[3], mb[3], sb[3];
                bool oka, okb;
                pair(i, two ? i2 : i, ma, sa, oka, mb, sb, okb);
                inside = oka && inside;
#pragma unroll
                for (int k = 0; k < NCC_KEEP; ++k)
                    if (slot == k)
                        for (int c = 0; c < 3; ++c) {
                            keep_m[k][c] = ma[c];
                            keep_s[k][c] = sa[c];
                        }
                if (slot >= NCC_KEEP)
                    stash_put(slot, sa);
                for (int c = 0; c < 3; ++c) {
                    sum0[c] += (double)ma[c];
                    sum1[c] += (double)sa[c];
                }
                if (two) {
                    inside = okb && inside;
#pragma unroll
                    for (int k = 0; k < NCC_KEEP; ++k)
                        if (slot + 1 == k)
                            for (int c = 0; c < 3; ++c) {
                                keep_m[k][c] = mb[c];
                                keep_s[k][c] = sb[c];
                            }
                    if (slot + 1 >= NCC_KEEP)
                        stash_put(slot + 1, sb);
                    for (int c = 0; c < 3; ++c) {
                        sum0[c] += (double)mb[c];
                        sum1[c] += (double)sb[c];
                    }
                }
            }
        }
        for (int pass = 0; pass < 2; ++pass) {
            if (alive && inside && !(pass == 0 && rgb && A.ncc_pairs != 0)) {
                int slot = 0;
                for (int i = gl; i < n; i += G, ++slot) {
                    double cm[3], cs[3];
                    if (pass == 0) {
                        // (no early exit when a sample falls outside: the taps are
                        // clamped into the image, the sums of such a patch are never
                        // used (ncc = -1), and a loop without an exit lets the loads
                        // of the next sample start under the arithmetic of this one)
                        inside = colours(i, cm, cs, true) && inside;
#pragma unroll
                        for (int k = 0; k < NCC_KEEP; ++k)
                            if (slot == k)
                                for (int c = 0; c < 3; ++c) {
                                    keep_m[k][c] = (float)cm[c];
                                    keep_s[k][c] = (float)cs[c];
                                }
                        if (slot >= NCC_KEEP) {
                            // (cs[] ARE floats widened: linear_at's results)
                            float const sf[3] = { (float)cs[0], (float)cs[1], (float)cs[2] };
                            stash_put(slot, sf);
                        }
                        for (int c = 0; c < 3; ++c) {
                            sum0[c] += cm[c];
                            sum1[c] += cs[c];
                        }
                    } else {
                        if (slot < NCC_KEEP) {
#pragma unroll
                            for (int k = 0; k < NCC_KEEP; ++k)
                                if (slot == k)
                                    for (int c = 0; c < 3; ++c) {
                                        cm[c] = keep_m[k][c];
                                        cs[c] = keep_s[k][c];
                                    }
                        } else if (slot - NCC_KEEP < A.ncc_stash_slots) {
                            // the neighbour's colour from the stash, the main
                            // view's read again (one load against a warp, a
                            // division and four taps)
                            NccSample const smp = sample_at(i);
                            size_t const at = (size_t)(py + smp.dy) * mv.w + (px + smp.dx);
                            for (int c = 0; c < 3; ++c) {
                                int const cmi = c < mv.c - 1 ? c : mv.c - 1;
                                cm[c] = mv.image[at * mv.c + cmi];
                                cs[c] = ncc_stash[((slot - NCC_KEEP) * 3 + c) * 256 + threadIdx.x];
                            }
                        } else {
                            (void)colours(i, cm, cs, false);
                        }
                        for (int c = 0; c < 3; ++c) {
                            double const a = cm[c] - mean0[c];
                            double const b = cs[c] - mean1[c];
                            n0 += a * a;
                            n1 += b * b;
                            dot += a * b;
                        }
                    }
                }
            }
            if (pass == 0) {
                inside = group_all(inside, G, lane, red);
                SharedDivisor const by_n((double)n, A.exact_divisions == 0);
                double six[6] = { sum0[0], sum0[1], sum0[2], sum1[0], sum1[1], sum1[2] };
                vis_group_sums<6>(six, G, red);
                for (int c = 0; c < 3; ++c) {
                    mean0[c] = by_n.quotient(six[c]);
                    mean1[c] = by_n.quotient(six[3 + c]);
                }
            }
        }
        double three[3] = { n0, n1, dot };
        vis_group_sums<3>(three, G, red);
        n0 = sqrt(three[0]);
        n1 = sqrt(three[1]);
        dot = three[2];
        if (!inside)
            ncc = -1.0;
        else if (n0 + n1 < 0.001 * n)
            ncc = 1.0;
        else
            ncc = dot / (n0 * n1);
    }
    if (alive && gl == 0 && !(ncc < 0))
        atomicOr(&A.vis_out[p], 1u << s);
}

// ---- mse_for_patch (:747-790) ----
// Which patches are asked about: all valid ones, or (cut_boundaries) those
// with a node that has lost more than one neighbour (:401-428) -- the rim of
// the surface; the others are not evaluated (0: never above 0.05).  One thread
// per patch writes the answer of everything that is not evaluated and appends
// the rest to a list, so that the kernel doing the arithmetic is launched over
// the few per cent that need it: sixteen lanes per patch of a 129 k-patch grid
// were 32 k waves that each waited for two dependent loads to learn that they
// had nothing to do -- that, not the arithmetic, was the 35-40 us of a pass.
// (The order of the list is whatever the atomics make it; an entry's result
// does not depend on its place.)
__global__ void __launch_bounds__(256)
topo_mse_candidates_kernel(TopoArgs A)
{
    // (a pass enqueued ahead whose predecessor ended the loop: every thread of
    // the launch reads the same word)
    if (A.pass_gate != nullptr && *A.pass_gate <= 10)
        return;
    int const p = blockIdx.x * blockDim.x + threadIdx.x;
    bool const in_range = p < A.num_patches;
    bool const valid = in_range && A.patch_valid[p];
    bool alive = valid;
    if (valid && A.only_candidates) {
        int const n00 = (p / A.npx) * A.stride + p % A.npx;
        alive = (A.border_node[n00] | A.border_node[n00 + 1] | A.border_node[n00 + A.stride]
            | A.border_node[n00 + A.stride + 1]) != 0;
    }
    if (in_range && !alive)
        A.mse_out[p] = valid ? 0.0 : -1.0;
    // one atomic per workgroup: the list's end is one word for the whole grid
    __shared__ int wave_count[4];
    __shared__ int block_base;
    unsigned long long const mask = __ballot(alive);
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        wave_count[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        int const total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        block_base = total > 0 ? atomicAdd(A.mse_count, total) : 0;
    }
    __syncthreads();
    if (alive) {
        int at = block_base + __popcll(mask & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w)
            at += wave_count[w];
        A.mse_list[at] = p;
    }
}

// One lane group per listed patch; the launch is bounded and a group takes
// every (number of groups)-th entry.  AT_ONCE: neighbours whose warps and
// gathers are issued together.
template <int AT_ONCE>
__global__ void __launch_bounds__(256)
topo_mse_kernel(TopoArgs A)
{
#pragma clang fp contract(off)
    __shared__ double red[4];
    // (a pass enqueued ahead whose predecessor ended the loop: every thread of
    // the launch reads the same word)
    if (A.pass_gate != nullptr && *A.pass_gate <= 10)
        return;
    int const ps = A.ps;
    int const G = group_size(ps, MSE_WORKGROUP_FROM);
    int const gl = threadIdx.x & (G - 1);
    int const g_log2 = 31 - __clz(G);
    int const group = (int)(((unsigned)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2);
    int const groups = (int)(((unsigned)gridDim.x * blockDim.x) >> g_log2);
    int const count = *A.mse_count;
    // Patch sizes 32 and 64: a patch is 4 resp. 16 CHUNKS of 256 pixels, each a
    // workgroup's item of its own (round 6).  The two coarsest scales have a
    // few hundred candidates at most, so a workgroup per patch left most of the
    // chip idle behind chains of 16 pixels x 8 neighbours per lane (140-170 us
    // per launch at patch size 64).  A chunk leaves its two sums in
    // `mse_parts`; the chunk that arrives last (one atomic per chunk) adds them
    // in chunk order -- a fixed order, whichever workgroup does it -- and
    // clears the counter for the next launch.
    int const chunks = A.mse_chunks;
    // (G == 256: the group is the workgroup, its threads loop together and
    // meet in group_sum's barriers; smaller groups only shuffle among
    // themselves)
    for (int work = group; work < count * chunks; work += groups) {
        int const item = chunks > 1 ? work / chunks : work;
        int const chunk = chunks > 1 ? work - item * chunks : 0;
        int const p = A.mse_list[item];
        double n16[16];
        load_patch_nodes(A, p, n16);
        int const px = A.start_x + (p % A.npx) * ps;
        int const py = A.start_y + (p / A.npx) * ps;
        // (bits of neighbours the context does not have are not looked at)
        uint32_t const vis = A.patch_vis[p] & ((1u << A.n_subs) - 1u);
        double error = 0.0, counter = 0.0;
        int const k_begin = chunks > 1 ? chunk * 256 : 0;
        int const k_end = chunks > 1 ? k_begin + 256 : ps * ps;
        for (int k = k_begin + gl; k < k_end; k += G) {
            int const i = k & (ps - 1), j = k >> A.ps_log2;
            // (asked for before the surface is evaluated: a cold round trip)
            float2 const gm = A.main_grad[(size_t)(py + j) * A.W + (px + i)];
            // (x / ps == x * (1 / ps) exactly: ps is a power of two)
            double const u = (i + 0.5) * A.inv_ps, v = (j + 0.5) * A.inv_ps;
            double const w = smvs_topo::patch_eval(n16, u, v, 0, 0);
            double const wx = smvs_topo::patch_eval(n16, u, v, 1, 0) * A.inv_ps;
            double const wy = smvs_topo::patch_eval(n16, u, v, 0, 1) * A.inv_ps;
            double const gm0 = gm.x, gm1 = gm.y;
            // The neighbours AT_ONCE at a time: the warps, then the gathers of
            // all of them, then the sum in the neighbours' order (the few rim
            // patches this kernel is asked about make a launch as long as one
            // lane's chain of dependent divisions and gathers).
            for (int s0 = 0; s0 < A.n_subs; s0 += AT_ONCE) {
                uint32_t const some = (vis >> s0) & ((1u << AT_ONCE) - 1u);
                if (some == 0u)
                    continue;
                double jac[AT_ONCE][4];
                float g0[AT_ONCE], g1[AT_ONCE];
                auto const gather = [&](auto tag) {
#pragma unroll
                    for (int e = 0; e < AT_ONCE; ++e) {
                        // (an unseen neighbour: the first one's planes at pixel 0,
                        // loaded and not used)
                        bool const on = ((some >> e) & 1u) != 0u;
                        int const sc = on ? s0 + e : s0;
                        const double *M = A.cams->M[sc];
                        Warp wp(M, A.cams->t[sc], px + i + 0.5, py + j + 0.5, w);
                        WarpQuotients<decltype(tag)::value> const wq(wp);
                        wq.jacobian(wp, M, w, wx, wy, jac[e]);
                        float const qx = on ? (float)(wq.x(wp) - 0.5) : 0.0f;
                        float const qy = on ? (float)(wq.y(wp) - 0.5) : 0.0f;
                        SubPlanes const sp = A.subs[sc];
                        linear_at_pair(sp.grad, sp.width, sp.height, qx, qy, &g0[e], &g1[e]);
                    }
                };
                bool plain = A.exact_divisions == 0;
#pragma unroll
                for (int e = 0; e < AT_ONCE; ++e) {
                    int const sc = ((some >> e) & 1u) != 0u ? s0 + e : s0;
                    Warp wp(A.cams->M[sc], A.cams->t[sc], px + i + 0.5, py + j + 0.5, w);
                    plain = plain && WarpQuotients<true>(wp).plain();
                }
                if (plain)
                    gather(std::true_type());
                else
                    gather(std::false_type());
#pragma unroll
                for (int e = 0; e < AT_ONCE; ++e) {
                    if (((some >> e) & 1u) == 0u)
                        continue;
                    double const d0 = gm0 - (jac[e][0] * (double)g0[e] + jac[e][1] * (double)g1[e]);
                    double const d1 = gm1 - (jac[e][2] * (double)g0[e] + jac[e][3] * (double)g1[e]);
                    error += sqrt(d0 * d0 + d1 * d1);
                    counter += 1.0;
                }
            }
        }
        error = group_sum(error, G, red);
        counter = group_sum(counter, G, red);
        if (chunks == 1) {
            if (gl == 0)
                A.mse_out[p] = counter == 0.0 ? 1.0 : error / counter;
            continue;
        }
        // (chunks > 1 only with G == 256: the workgroup is the group)
        if (threadIdx.x == 0) {
            double *mine = A.mse_parts + 2 * ((size_t)item * chunks + chunk);
            __hip_atomic_store(mine, error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1, counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int const before = __hip_atomic_fetch_add(A.mse_arrived + item, 1, __ATOMIC_ACQ_REL,
                __HIP_MEMORY_SCOPE_AGENT);
            if (before == chunks - 1) {
                double e = 0.0, c = 0.0;
                for (int q = 0; q < chunks; ++q) {
                    const double *part = A.mse_parts + 2 * ((size_t)item * chunks + q);
                    e += __hip_atomic_load(part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    c += __hip_atomic_load(part + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                A.mse_out[p] = c == 0.0 ? 1.0 : e / c;
                __hip_atomic_store(A.mse_arrived + item, 0, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- cut_boundaries (:401-428): the nodes with more than one missing
// neighbour node (outside the grid counts as missing) ----
__global__ void __launch_bounds__(256)
topo_border_nodes_kernel(TopoArgs A)
{
    // The first kernel of a cut pass also clears the pass's two counters
    // (patches deleted, candidates listed): the kernels that count run behind
    // this one on the stream, and the host has read the previous pass's values
    // before it enqueues this launch -- one runtime fill kernel per pass less
    // (50 per --no-sgm view) in a loop whose cost is its launches.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *A.deleted = 0;
        *A.mse_count = 0;
    }
    // (a pass enqueued ahead whose predecessor ended the loop: every thread of
    // the launch reads the same word)
    if (A.pass_gate != nullptr && *A.pass_gate <= 10)
        return;
    int const n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.num_nodes)
        return;
    int const nx = n % A.stride, ny = n / A.stride;
    int missing = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy)
                continue;
            int const mx = nx + dx, my = ny + dy;
            bool const exists = mx >= 0 && my >= 0 && mx <= A.npx && my <= A.npy
                && A.node_valid_rw[(size_t)my * A.stride + mx] != 0;
            missing += exists ? 0 : 1;
        }
    A.border_node[n] = missing > 1 ? 1 : 0;
}

// ---- one pass of cut_boundaries (:360-431), patches ----
__global__ void __launch_bounds__(256)
topo_cut_patches_kernel(TopoArgs A)
{
#pragma clang fp contract(off)
    // (a pass enqueued ahead whose predecessor ended the loop: every thread of
    // the launch reads the same word)
    if (A.pass_gate != nullptr && *A.pass_gate <= 10)
        return;
    int const p = blockIdx.x * blockDim.x + threadIdx.x;
    bool remove = false;
    if (p < A.num_patches && A.patch_valid_rw[p]) {
        int const ix = p % A.npx, iy = p / A.npx;
        int const n00 = iy * A.stride + ix;
        int const ids[4] = { n00, n00 + 1, n00 + A.stride, n00 + A.stride + 1 };
        // depth discontinuity (:371-399)
        double f[4];
        for (int k = 0; k < 4; ++k)
            f[k] = A.nodes[4 * (size_t)ids[k]];
        int lo = 0, hi = 0;  // first minimum, last maximum (multimap order)
        for (int i = 1; i < 4; ++i) {
            if (f[i] < f[lo])
                lo = i;
            if (f[i] >= f[hi])
                hi = i;
        }
        double dd_factor = 5.0;
        if (lo + hi == 3)
            dd_factor *= 1.41421356237309504880;
        int const px = A.start_x + ix * A.ps, py = A.start_y + iy * A.ps;
        float const fx = (float)px + 0.5f, fy = (float)py + 0.5f;
        float v[3];
        for (int r = 0; r < 3; ++r)
            v[r] = A.invproj[3 * r] * fx + A.invproj[3 * r + 1] * fy
                + A.invproj[3 * r + 2] * 1.0f;
        float const vnorm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        double const threshold = dd_factor * f[lo] * A.invproj[0] * A.ps / vnorm;
        if (f[hi] - f[lo] > threshold)
            remove = true;
        // high-error patch on the border of the surface (:401-428): a node of
        // it has more than one missing neighbour node (topo_border_nodes_kernel:
        // node validity as it was before this pass)
        if (!remove && A.mse_out[p] > 0.05) {
            for (int k = 0; k < 4 && !remove; ++k)
                if (A.border_node[ids[k]])
                    remove = true;
        }
        if (remove)
            A.patch_valid_rw[p] = 0;
    }
    int const cnt = __syncthreads_count(remove);
    if (threadIdx.x == 0 && cnt != 0)
        atomicAdd(A.deleted, cnt);
}

// ---- Surface::remove_nodes_without_patch (surface.cc:762-869) ----
__global__ void __launch_bounds__(256)
topo_cut_nodes_kernel(TopoArgs A)
{
    // (a pass enqueued ahead whose predecessor ended the loop: every thread of
    // the launch reads the same word)
    if (A.pass_gate != nullptr && *A.pass_gate <= 10)
        return;
    int const n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.num_nodes || !A.node_valid_rw[n])
        return;
    int const idx = n % A.stride, idy = n / A.stride;
    bool any = false;
    for (int dy = -1; dy <= 0; ++dy)
        for (int dx = -1; dx <= 0; ++dx) {
            int const qx = idx + dx, qy = idy + dy;
            if (qx >= 0 && qy >= 0 && qx < A.npx && qy < A.npy
                && A.patch_valid_rw[(size_t)qy * A.npx + qx])
                any = true;
        }
    if (!any)
        A.node_valid_rw[n] = 0;
}

// ---- a cut pass in three launches instead of five (round 6) ----
// A pass was border nodes -> candidates -> errors -> cut patches -> cut nodes,
// five launches of ~5 us around one of ~20 us, and a view makes 24-50 passes:
// what a pass costs is its launches (profiles/r6_cut_passes_ahead.txt).  Both
// ends are fused by RECOMPUTATION -- the dependences are local:
//  * topo_border_candidates_kernel: thread i writes the border flag of node i
//    and decides the candidacy of patch i from the flags of its four nodes,
//    which it forms itself from the 4 x 4 node validities around the patch (the
//    same predicate on the same bytes: nobody writes node validity here);
//  * topo_cut_fused_kernel: thread n evaluates the removal predicate of the (up
//    to) four patches around node n -- the one whose first node it is, it also
//    deletes and counts -- and keeps the node iff one of them stays.  A thread
//    may read a patch's validity before or after its owner cleared it: the
//    predicate does not depend on any validity, so both give the same answer.
// The pass's two counters cannot be cleared by its first kernel any more (other
// workgroups of the same launch add to them): passes alternate between two
// pairs of words and the LAST kernel of a pass clears the pair of the next one,
// whose previous values the host has read (it waits for every pass).
// SMVS_CUT_FUSED=0: the five launches.
__device__ __forceinline__ bool
node_on_border(TopoArgs const &A, int n)
{
    int const nx = n % A.stride, ny = n / A.stride;
    int missing = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy)
                continue;
            int const mx = nx + dx, my = ny + dy;
            bool const exists = mx >= 0 && my >= 0 && mx <= A.npx && my <= A.npy
                && A.node_valid_rw[(size_t)my * A.stride + mx] != 0;
            missing += exists ? 0 : 1;
        }
    return missing > 1;
}

__global__ void __launch_bounds__(256)
topo_border_candidates_kernel(TopoArgs A)
{
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.num_nodes)
        A.border_node[i] = node_on_border(A, i) ? 1 : 0;
    int const p = i;
    bool const in_range = p < A.num_patches;
    bool const valid = in_range && A.patch_valid[p];
    bool alive = valid;
    if (valid && A.only_candidates) {
        int const n00 = (p / A.npx) * A.stride + p % A.npx;
        alive = node_on_border(A, n00) || node_on_border(A, n00 + 1)
            || node_on_border(A, n00 + A.stride) || node_on_border(A, n00 + A.stride + 1);
    }
    if (in_range && !alive)
        A.mse_out[p] = valid ? 0.0 : -1.0;
    // one atomic per workgroup: the list's end is one word for the whole grid
    __shared__ int wave_count[4];
    __shared__ int block_base;
    unsigned long long const mask = __ballot(alive);
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        wave_count[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        int const total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        block_base = total > 0 ? atomicAdd(A.mse_count, total) : 0;
    }
    __syncthreads();
    if (alive) {
        int at = block_base + __popcll(mask & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w)
            at += wave_count[w];
        A.mse_list[at] = p;
    }
}

// topo_cut_patches_kernel's predicate for a VALID patch (:371-428)
__device__ __forceinline__ bool
cut_removes_patch(TopoArgs const &A, int p)
{
#pragma clang fp contract(off)
    int const ix = p % A.npx, iy = p / A.npx;
    int const n00 = iy * A.stride + ix;
    int const ids[4] = { n00, n00 + 1, n00 + A.stride, n00 + A.stride + 1 };
    double f[4];
    for (int k = 0; k < 4; ++k)
        f[k] = A.nodes[4 * (size_t)ids[k]];
    int lo = 0, hi = 0;  // first minimum, last maximum (multimap order)
    for (int i = 1; i < 4; ++i) {
        if (f[i] < f[lo])
            lo = i;
        if (f[i] >= f[hi])
            hi = i;
    }
    double dd_factor = 5.0;
    if (lo + hi == 3)
        dd_factor *= 1.41421356237309504880;
    int const px = A.start_x + ix * A.ps, py = A.start_y + iy * A.ps;
    float const fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float v[3];
    for (int r = 0; r < 3; ++r)
        v[r] = A.invproj[3 * r] * fx + A.invproj[3 * r + 1] * fy
            + A.invproj[3 * r + 2] * 1.0f;
    float const vnorm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double const threshold = dd_factor * f[lo] * A.invproj[0] * A.ps / vnorm;
    if (f[hi] - f[lo] > threshold)
        return true;
    if (A.mse_out[p] > 0.05)
        for (int k = 0; k < 4; ++k)
            if (A.border_node[ids[k]])
                return true;
    return false;
}

__global__ void __launch_bounds__(256)
topo_cut_fused_kernel(TopoArgs A, int *next_counters)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        next_counters[0] = 0;
        next_counters[1] = 0;
    }
    int const n = blockIdx.x * blockDim.x + threadIdx.x;
    bool removed_own = false;
    if (n < A.num_nodes) {
        int const idx = n % A.stride, idy = n / A.stride;
        bool const node_valid = A.node_valid_rw[n] != 0;
        bool any = false;
        for (int dy = -1; dy <= 0; ++dy)
            for (int dx = -1; dx <= 0; ++dx) {
                int const qx = idx + dx, qy = idy + dy;
                if (!(qx >= 0 && qy >= 0 && qx < A.npx && qy < A.npy))
                    continue;
                bool const own = dx == 0 && dy == 0;
                // (the patches around an invalid node matter only to their owners)
                if (!own && !node_valid)
                    continue;
                int const q = qy * A.npx + qx;
                if (!A.patch_valid_rw[q])
                    continue;
                bool const remove = cut_removes_patch(A, q);
                if (!remove)
                    any = true;
                else if (own) {
                    A.patch_valid_rw[q] = 0;
                    removed_own = true;
                }
            }
        if (node_valid && !any)
            A.node_valid_rw[n] = 0;
    }
    int const cnt = __syncthreads_count(removed_own);
    if (threadIdx.x == 0 && cnt != 0)
        atomicAdd(A.deleted, cnt);
}

static int
fill_args(smvs_ctx *ctx, TopoArgs *A, const char *who)
{
    if (!ctx->has_surface || !ctx->has_cameras) {
        set_error("%s: cameras and surface must be set first", who);
        return SMVS_ERR_STATE;
    }
    for (int v = 0; v <= ctx->n_subs; ++v)
        if (!((ctx->image_ok >> v) & 1u)) {
            set_error("%s: view %d has no image (smvs_ctx_upload_image)", who,
                v - 1);
            return SMVS_ERR_STATE;
        }
    {
        // (images handed over by smvs_ctx_upload_image_async and not read yet)
        int const rc = ctx_materialise_images(ctx, ~0u);
        if (rc != SMVS_OK)
            return rc;
    }
    A->nodes = ctx->nodes;
    A->patch_valid = ctx->patch_valid;
    A->patch_vis = ctx->patch_vis;
    A->vis_out = ctx->patch_vis;
    A->mse_out = ctx->topo_mse;
    A->cams = ctx->cams;
    for (int v = 0; v <= ctx->n_subs; ++v) {
        A->views[v].w = ctx->images[v].w;
        A->views[v].h = ctx->images[v].h;
        A->views[v].c = ctx->images[v].c;
        A->views[v].image = ctx->images[v].data;
    }
    A->main_grad = ctx->main_grad;
    A->subs = ctx->subs_dev;
    for (int s = 0; s < SMVS_MAX_SUBS; ++s) {
        A->zbuf[s] = ctx->topo_zbuf[s];
        A->zraw[s] = ctx->topo_zbuf[s] == nullptr ? nullptr
            : ctx->topo_zbuf[s] + (size_t)(ctx->images[1 + s].w + 1)
                * (ctx->images[1 + s].h + 1);
    }
    A->sgm_depth = nullptr;
    {
        const char *mode = std::getenv("SMVS_TOPO_DIVIDE");
        A->exact_divisions = mode != nullptr && std::strcmp(mode, "exact") == 0 ? 1 : 0;
        const char *pairs = std::getenv("SMVS_NCC_PAIRS");
        A->ncc_pairs = pairs != nullptr && std::atoi(pairs) == 0 ? 0 : 1;
        const char *window = std::getenv("SMVS_ZBUF_WINDOW");
        A->zbuf5 = window != nullptr && std::atoi(window) == 3 ? 0 : 1;
    }
    A->pair_alive = nullptr;
    A->pass_gate = nullptr;

    A->ncc = ctx->topo_ncc;
    for (int i = 0; i < 33; ++i)
        A->ncc_off[i] = ctx->topo_ncc_off[i];
    A->W = ctx->width;
    A->H = ctx->height;
    A->npx = ctx->npx;
    A->npy = ctx->npy;
    A->stride = ctx->node_stride;
    A->ps = ctx->patchsize;
    A->ps_log2 = 0;
    while ((1 << A->ps_log2) < ctx->patchsize)
        A->ps_log2 += 1;
    A->inv_ps = 1.0 / (double)ctx->patchsize;
    A->start_x = ctx->start_x;
    A->start_y = ctx->start_y;
    A->n_subs = ctx->n_subs;
    A->num_patches = ctx->num_patches;
    A->use_ncc = 0;
    A->patch_valid_rw = ctx->patch_valid;
    A->node_valid_rw = ctx->node_valid;
    A->deleted = ctx->status + I_TOPO_DELETED;
    A->mse_list = ctx->topo_mse_list;
    A->mse_count = ctx->status + I_TOPO_CANDIDATES;
    A->num_nodes = ctx->num_nodes;
    A->border_node = ctx->topo_border;
    A->only_candidates = 0;
    A->pix = ctx->topo_pix;
    for (int i = 0; i < 9; ++i)
        A->invproj[i] = 0.0f;
    return SMVS_OK;
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_topology_subviews(smvs_ctx *ctx, const float *sgm_depth, int use_ncc,
    uint32_t *patch_vis_out)
{
    SMVS_REQUIRE(ctx != nullptr, "null argument");
    SMVS_HIP_CHECK(set_device(ctx->device));
    if ((ctx->image_ok & 1u) != 0u
        && (ctx->images[0].w != ctx->width || ctx->images[0].h != ctx->height)) {
        set_error("smvs_topology_subviews: main image size differs from the context");
        return SMVS_ERR_INVALID;
    }
    // (the kernels index the per-pixel planes and the float images with 32-bit offsets)
    if ((size_t)ctx->width * ctx->height * 3 >= ((size_t)1 << 31)) {
        set_error("smvs_topology_subviews: images of 2^31 / 3 pixels and more are not supported");
        return SMVS_ERR_INVALID;
    }
    for (int j = 0; j <= ctx->n_subs; ++j)
        if ((size_t)ctx->images[j].w * ctx->images[j].h * 3 >= ((size_t)1 << 31)) {
            set_error("smvs_topology_subviews: images of 2^31 / 3 pixels and more are not supported");
            return SMVS_ERR_INVALID;
        }
    int rc;
    // the 32 sample templates of ncc_for_patch for this patch size
    if (ctx->topo_ncc_ps != ctx->patchsize && ctx->has_surface) {
        std::vector<smvs_topo::NccSample> all;
        for (int f = 0; f < 32; ++f) {
            ctx->topo_ncc_off[f] = (int)all.size();
            std::vector<smvs_topo::NccSample> const one
                = smvs_topo::build_ncc_template(ctx->patchsize, f);
            all.insert(all.end(), one.begin(), one.end());
        }
        ctx->topo_ncc_off[32] = (int)all.size();
        if ((rc = device_alloc(&ctx->topo_ncc, all.size())) != SMVS_OK)
            return rc;
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->topo_ncc, all.data(),
            all.size() * sizeof(smvs_topo::NccSample), hipMemcpyHostToDevice,
            ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->topo_ncc_ps = ctx->patchsize;
    }
    for (int s = 0; s < ctx->n_subs; ++s) {
        size_t const n = (size_t)(ctx->images[1 + s].w + 1)
            * (ctx->images[1 + s].h + 1);
        if (n > ctx->topo_zbuf_cap[s]) {
            // (the z-buffer and, behind it, the per-centre minima)
            if ((rc = device_alloc(&ctx->topo_zbuf[s], 2 * n)) != SMVS_OK)
                return rc;
            ctx->topo_zbuf_cap[s] = n;
        }
    }
    size_t const npix = (size_t)ctx->width * ctx->height;
    if (ctx->topo_pix_cap < npix * 3) {
        ctx->topo_pix_cap = 0;
        if ((rc = device_alloc(&ctx->topo_pix, npix * 3)) != SMVS_OK)
            return rc;
        ctx->topo_pix_cap = npix * 3;
    }
    if (sgm_depth != nullptr && ctx->topo_sgm_cap < npix) {
        if ((rc = device_alloc(&ctx->topo_sgm, npix)) != SMVS_OK)
            return rc;
        ctx->topo_sgm_cap = npix;
    }
    TopoArgs A;
    if ((rc = fill_args(ctx, &A, "smvs_topology_subviews")) != SMVS_OK)
        return rc;
    A.use_ncc = use_ncc ? 1 : 0;
    if (sgm_depth != nullptr) {
        if ((rc = ctx_upload(ctx, ctx->topo_sgm, sgm_depth, npix * sizeof(float)))
                != SMVS_OK)
            return rc;
        ctx->sgm_resident = false;   // (overwritten by the caller's map)
        A.sgm_depth = ctx->topo_sgm;
    } else if (ctx->sgm_resident) {
        // the map smvs_ctx_sgm_init_depth left on the device
        A.sgm_depth = ctx->topo_sgm;
    }
    {
        // every neighbour's per-centre minima start at 10000 (one launch; a
        // memset per neighbour is two runtime kernels each)
        size_t cells = 0;
        for (int s = 0; s < ctx->n_subs; ++s)
            cells = std::max(cells, (size_t)(ctx->images[1 + s].w + 1)
                * (ctx->images[1 + s].h + 1));
        hipLaunchKernelGGL(topo_clear_kernel, dim3((unsigned)((cells + 255) / 256), 1,
            (unsigned)ctx->n_subs), dim3(256), 0, ctx->stream, A);
    }
    hipLaunchKernelGGL(topo_splat_kernel, dim3((ctx->width + 255) / 256,
        ctx->height), dim3(256), 0, ctx->stream, A);
    {
        int zw = 0, zh = 0;
        for (int s = 0; s < ctx->n_subs; ++s) {
            zw = std::max(zw, ctx->images[1 + s].w + 1);
            zh = std::max(zh, ctx->images[1 + s].h + 1);
        }
        // (column blocks padded to a multiple of 8: vertically adjacent row blocks
        // then share an XCD's L2, csrc/scale.hip launch_blur_ks)
        if (A.zbuf5)
            hipLaunchKernelGGL(topo_dilate5_kernel,
                dim3((((unsigned)zw + DIL5_COLS - 1) / DIL5_COLS + 7u) & ~7u,
                    (zh + DIL5_ROWS - 1) / DIL5_ROWS, ctx->n_subs), dim3(256), 0, ctx->stream, A);
        else
            hipLaunchKernelGGL(topo_dilate_kernel, dim3((((unsigned)zw + 255u) / 256u + 7u) & ~7u,
                (zh + DILATE_ROWS - 1) / DILATE_ROWS, ctx->n_subs), dim3(256), 0, ctx->stream, A);
    }
    {
        long long const pixels = (long long)ctx->num_patches * ctx->patchsize * ctx->patchsize;
        hipLaunchKernelGGL(topo_pixel_surface_kernel, dim3((unsigned)((pixels + 255) / 256)),
            dim3(256), 0, ctx->stream, A);
    }
    // Lanes per (patch, neighbour), measured per patch size on a --no-sgm view at
    // 1920 x 1080 x 8 (profiles/r6_visibility_groups.txt; SMVS_VIS_GROUP_<ps>=<lanes>
    // is the A/B switch).  Few lanes win wherever there are enough groups to fill
    // the chip: a lane's pixels and samples are independent chains either way, and
    // the group's set-up and eleven reductions are paid once per group --
    // patch size 4 (16 pixels, 44 samples): 4 lanes 627 us, 8: 704, 16: 880;
    // patch size 8: 8 lanes 443, 16: 460, 32: 525, 64: 680; 16: 32 lanes 365, 64: 385;
    // 32: 64 lanes 310, 32: 385 (a lane's samples outgrow the stash), 256: 365;
    // 64: the workgroup 290, 64 lanes 415.
    long long group = group_size(ctx->patchsize, VIS_WORKGROUP_FROM);
    switch (ctx->patchsize) {
    case 2: group = 2; break;
    case 4: group = 4; break;
    case 8: group = 8; break;
    case 16: group = 32; break;
    default: break;
    }
    {
        char name[32];
        std::snprintf(name, sizeof(name), "SMVS_VIS_GROUP_%d", ctx->patchsize);
        const char *e = std::getenv(name);
        int const g = e != nullptr ? std::atoi(e) : 0;
        if (g == 256 || (g >= 1 && g <= 64 && (g & (g - 1)) == 0))
            group = g;
    }
    long long const items = (long long)ctx->num_patches * group;   // (per neighbour: grid.y)
    A.vis_group = (int)group;
    A.ncc_stash_slots = 0;
    if (use_ncc) {
        int n_max = 0;
        for (int f = 0; f < 32; ++f)
            n_max = std::max(n_max, ctx->topo_ncc_off[f + 1] - ctx->topo_ncc_off[f]);
        static bool const no_stash = [] {
            const char *e = std::getenv("SMVS_NCC_STASH");
            return e != nullptr && e[0] == '0';
        }();
        int const per_lane = (int)((n_max + group - 1) / group);
        A.ncc_stash_slots = no_stash ? 0 : std::min(NCC_STASH_MAX, std::max(0, per_lane - NCC_KEEP));
    }
    size_t stash_bytes = (size_t)A.ncc_stash_slots * 3 * 256 * sizeof(float);
    A.lds_depth_doubles = 0;
    A.lds_tpl_n = 0;
    if (use_ncc) {
        // (three workgroups per CU -- what the kernel's registers allow -- leave
        // each 53 KB of the 160)
        size_t const budget = 52 * 1024;
        static bool const no_lds = [] {
            const char *e = std::getenv("SMVS_NCC_LDS");
            return e != nullptr && e[0] == '0';
        }();
        size_t const depth_doubles = (size_t)(256 / group) * ((size_t)ctx->patchsize * ctx->patchsize + 4);
        size_t const tpl_n = (size_t)(ctx->topo_ncc_off[32] - ctx->topo_ncc_off[31]);
        if (!no_lds && group <= 256 && stash_bytes + depth_doubles * 8 <= budget) {
            A.lds_depth_doubles = (int)depth_doubles;
            stash_bytes += depth_doubles * 8;
        }
        if (!no_lds && stash_bytes + tpl_n * 8 <= budget) {
            A.lds_tpl_n = (int)tpl_n;
            stash_bytes += tpl_n * 8;
        }
    }
    // SMVS_VIS_SPLIT=1: the two halves as launches of their own.  Measured
    // (profiles/r6_visibility_split.txt): 165 + 461 us against 606 us fused --
    // the NCC half keeps its 159 VGPRs, and the fused kernel overlaps the two
    // halves' waits; the default stays fused.
    static bool const split = [] {
        const char *e = std::getenv("SMVS_VIS_SPLIT");
        return e != nullptr && e[0] == '1';
    }();
    if (use_ncc && split) {
        size_t const pairs = (size_t)ctx->num_patches * ctx->n_subs;
        if (pairs > ctx->topo_pair_cap) {
            ctx->topo_pair_cap = 0;
            if ((rc = device_alloc(&ctx->topo_pair_alive, pairs)) != SMVS_OK)
                return rc;
            ctx->topo_pair_cap = pairs;
        }
        A.pair_alive = ctx->topo_pair_alive;
        hipLaunchKernelGGL(topo_visibility_kernel<1>,
            dim3((unsigned)((items + 255) / 256), (unsigned)ctx->n_subs), dim3(256), 0, ctx->stream, A);
        hipLaunchKernelGGL(topo_visibility_kernel<2>,
            dim3((unsigned)((items + 255) / 256), (unsigned)ctx->n_subs), dim3(256), stash_bytes, ctx->stream, A);
    } else {
        hipLaunchKernelGGL(topo_visibility_kernel<0>,
            dim3((unsigned)((items + 255) / 256), (unsigned)ctx->n_subs), dim3(256), stash_bytes, ctx->stream, A);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    if (patch_vis_out == nullptr)
        return SMVS_OK;   // (the masks stay on the device: surface.hip)
    SMVS_HIP_CHECK(hipMemcpyAsync(patch_vis_out, ctx->patch_vis,
        sizeof(uint32_t) * ctx->num_patches, hipMemcpyDeviceToHost,
        ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

static int
prepare_patch_mse(smvs_ctx *ctx, TopoArgs *A, const char *who)
{
    if (ctx->main_grad == nullptr) {
        set_error("%s: no gradient planes", who);
        return SMVS_ERR_STATE;
    }
    for (int j = 0; j < ctx->n_subs; ++j)
        if (!((ctx->planes_ok >> j) & 1u)) {
            set_error("%s: sub view %d has no planes", who, j);
            return SMVS_ERR_STATE;
        }
    int rc;
    if ((size_t)ctx->num_patches > ctx->topo_mse_cap) {
        ctx->topo_mse_cap = 0;
        if ((rc = device_alloc(&ctx->topo_mse, (size_t)ctx->num_patches))
            != SMVS_OK)
            return rc;
        ctx->topo_mse_cap = (size_t)ctx->num_patches;
    }
    // (device_alloc frees the old buffer first: the capacity goes to zero with
    // it, so that a failed allocation is tried again by the next call instead of
    // leaving a null pointer behind a capacity that says it is there)
    if ((size_t)ctx->num_nodes > ctx->topo_border_cap) {
        ctx->topo_border_cap = 0;
        if ((rc = device_alloc(&ctx->topo_border, (size_t)ctx->num_nodes)) != SMVS_OK)
            return rc;
        ctx->topo_border_cap = (size_t)ctx->num_nodes;
    }
    if ((size_t)ctx->num_patches > ctx->topo_mse_list_cap) {
        ctx->topo_mse_list_cap = 0;
        if ((rc = device_alloc(&ctx->topo_mse_list, (size_t)ctx->num_patches)) != SMVS_OK)
            return rc;
        ctx->topo_mse_list_cap = (size_t)ctx->num_patches;
    }
    // patch sizes 32 and up: the kernel works in chunks of 256 pixels
    // (SMVS_MSE_CHUNKS=0: a workgroup per patch, as before round 6)
    static bool const no_chunks = [] {
        const char *e = std::getenv("SMVS_MSE_CHUNKS");
        return e != nullptr && e[0] == '0';
    }();
    int const pp = ctx->patchsize * ctx->patchsize;
    int const chunks = !no_chunks && group_size(ctx->patchsize, MSE_WORKGROUP_FROM) == 256
        && pp > 256 ? pp / 256 : 1;
    if (chunks > 1) {
        size_t const parts = (size_t)ctx->num_patches * chunks * 2;
        if (parts > ctx->topo_mse_parts_cap) {
            ctx->topo_mse_parts_cap = 0;
            if ((rc = device_alloc(&ctx->topo_mse_parts, parts)) != SMVS_OK)
                return rc;
            ctx->topo_mse_parts_cap = parts;
        }
        if ((size_t)ctx->num_patches > ctx->topo_mse_arrived_cap) {
            ctx->topo_mse_arrived_cap = 0;
            if ((rc = device_alloc(&ctx->topo_mse_arrived, (size_t)ctx->num_patches)) != SMVS_OK)
                return rc;
            SMVS_HIP_CHECK(hipMemsetAsync(ctx->topo_mse_arrived, 0,
                sizeof(int) * (size_t)ctx->num_patches, ctx->stream));
            ctx->topo_mse_arrived_cap = (size_t)ctx->num_patches;
        }
    }
    rc = fill_args(ctx, A, who);
    A->mse_chunks = chunks;
    A->mse_parts = ctx->topo_mse_parts;
    A->mse_arrived = ctx->topo_mse_arrived;
    return rc;
}

static int launch_patch_mse_listed(smvs_ctx *ctx, TopoArgs const &A);

// The candidate list, then the errors of its entries.  count_is_zero: the
// caller has cleared I_TOPO_CANDIDATES on the stream (with its own words).
static int
launch_patch_mse(smvs_ctx *ctx, TopoArgs const &A, bool count_is_zero)
{
    if (!count_is_zero)
        SMVS_HIP_CHECK(hipMemsetAsync(ctx->status + I_TOPO_CANDIDATES, 0, sizeof(int),
            ctx->stream));
    hipLaunchKernelGGL(topo_mse_candidates_kernel,
        dim3((unsigned)((ctx->num_patches + 255) / 256)), dim3(256), 0, ctx->stream, A);
    return launch_patch_mse_listed(ctx, A);
}

// The errors of the patches on the list (the candidates are on the stream).
static int
launch_patch_mse_listed(smvs_ctx *ctx, TopoArgs const &A)
{
    // enough groups for every CU to hold its fill of waves, never more than the
    // patches: a group walks the list with that stride
    long long const group = group_size(ctx->patchsize, MSE_WORKGROUP_FROM);
    long long const items = (long long)ctx->num_patches * group * A.mse_chunks;
    long long blocks = (items + 255) / 256;
    if (blocks > 1024)
        blocks = 1024;
    // SMVS_MSE_SUBS=1: the neighbours of a pixel one after the other (eight at
    // a time measured slower than four: 34 against 31 us)
    const char *subs = std::getenv("SMVS_MSE_SUBS");
    if (subs != nullptr && std::atoi(subs) == 1)
        hipLaunchKernelGGL(topo_mse_kernel<1>, dim3((unsigned)blocks), dim3(256), 0,
            ctx->stream, A);
    else
        hipLaunchKernelGGL(topo_mse_kernel<4>, dim3((unsigned)blocks), dim3(256), 0,
            ctx->stream, A);
    SMVS_HIP_CHECK(hipGetLastError());
    return SMVS_OK;
}

extern "C" int
smvs_topology_patch_mse(smvs_ctx *ctx, double *mse_out)
{
    SMVS_REQUIRE(ctx && mse_out, "null argument");
    SMVS_HIP_CHECK(set_device(ctx->device));
    TopoArgs A;
    int rc = prepare_patch_mse(ctx, &A, "smvs_topology_patch_mse");
    if (rc == SMVS_OK)
        rc = launch_patch_mse(ctx, A, false);
    if (rc != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemcpyAsync(mse_out, ctx->topo_mse,
        sizeof(double) * ctx->num_patches, hipMemcpyDeviceToHost, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_topology_cut_boundaries(smvs_ctx *ctx, const float *inv_calibration9,
    uint8_t *patch_valid_out, uint8_t *node_valid_out, int *total_deleted)
{
    SMVS_REQUIRE(ctx && inv_calibration9, "null argument");
    SMVS_HIP_CHECK(set_device(ctx->device));
    TopoArgs A;
    int const rc = prepare_patch_mse(ctx, &A, "smvs_topology_cut_boundaries");
    if (rc != SMVS_OK)
        return rc;
    for (int i = 0; i < 9; ++i)
        A.invproj[i] = inv_calibration9[i];
    // mse_for_patch only where a pass can ask for it: the patches that touch a
    // node with more than one missing neighbour, as the surface stands before
    // the pass (every pass creates new ones).  A patch's error does not change
    // between passes, so this equals evaluating every patch once up front --
    // which cost 125 us per call at 1920x1080 for the few per cent that matter.
    A.only_candidates = 1;
    int total = 0;
    int deleted = 11;
    bool const trace = std::getenv("SMVS_TOPO_TRACE") != nullptr;
    // `while (deleted > 10) deleted = cut_boundaries();` (depth_optimizer.cc:
    // 186-190, 323-337).  The passes CAN be enqueued ahead, SMVS_TOPO_AHEAD=2..4
    // at a time: pass k + 1 is gated on pass k's count on the device (its
    // kernels leave at once when that count is <= 10) and the host reads the
    // counts of the whole chunk with one synchronisation.  Measured in round 6
    // (profiles/r6_cut_passes_ahead.txt): four ahead is SLOWER, a warm optimize()
    // 18.5 -> 19.4 ms with SGM and 30.9 -> 34.0 ms without -- most calls end
    // after their first pass (24 passes in ~20 calls per view), and the 118
    // launches that then do nothing cost more than the round trips they save.
    // The default is one pass per synchronisation, as in round 5.
    static int const ahead = [] {
        const char *e = std::getenv("SMVS_TOPO_AHEAD");
        int const v = e != nullptr ? std::atoi(e) : 1;
        return v < 1 ? 1 : (v > TOPO_AHEAD ? TOPO_AHEAD : v);
    }();
    // (read per call: a test runs both forms in one process)
    const char *fused_env = std::getenv("SMVS_CUT_FUSED");
    bool const fused_passes = !(fused_env != nullptr && fused_env[0] == '0');
    if (fused_passes && ahead == 1) {
        // three launches per pass (topo_border_candidates_kernel); the pass's
        // counters alternate between the word pairs 0 and 1, each cleared by the
        // last kernel of the pass before
        int *const words = ctx->status + I_TOPO_PASS0;
        if (!ctx->topo_slots_clean) {
            SMVS_HIP_CHECK(hipMemsetAsync(words, 0, 4 * sizeof(int), ctx->stream));
            ctx->topo_slot = 0;
        }
        ctx->topo_slots_clean = false;
        unsigned const cover = (unsigned)((std::max(ctx->num_nodes, ctx->num_patches) + 255) / 256);
        while (deleted > 10) {
            int const slot = ctx->topo_slot;
            TopoArgs P = A;
            P.deleted = words + 2 * slot;
            P.mse_count = words + 2 * slot + 1;
            P.pass_gate = nullptr;
            hipLaunchKernelGGL(topo_border_candidates_kernel, dim3(cover), dim3(256), 0,
                ctx->stream, P);
            int const mrc = launch_patch_mse_listed(ctx, P);
            if (mrc != SMVS_OK)
                return mrc;
            hipLaunchKernelGGL(topo_cut_fused_kernel,
                dim3((unsigned)((ctx->num_nodes + 255) / 256)), dim3(256), 0, ctx->stream, P,
                words + 2 * (1 - slot));
            SMVS_HIP_CHECK(hipGetLastError());
            SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host + I_TOPO_PASS0 + 2 * slot,
                words + 2 * slot, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            ctx->topo_slot = 1 - slot;
            deleted = ctx->status_host[I_TOPO_PASS0 + 2 * slot];
            total += deleted;
            if (trace)
                std::fprintf(stderr, "[smvs topo] cut pass: %d of %d patches evaluated, %d deleted\n",
                    ctx->status_host[I_TOPO_PASS0 + 2 * slot + 1], ctx->num_patches, deleted);
        }
        ctx->topo_slots_clean = true;
    }
    while (deleted > 10) {
        // (the counters of a pass are cleared by its first kernel; the word pairs
        // are then no longer what the three-launch form expects to find)
        ctx->topo_slots_clean = false;
        for (int k = 0; k < ahead; ++k) {
            TopoArgs P = A;
            P.deleted = ctx->status + I_TOPO_PASS0 + 2 * k;
            P.mse_count = ctx->status + I_TOPO_PASS0 + 2 * k + 1;
            P.pass_gate = k == 0 ? nullptr : ctx->status + I_TOPO_PASS0 + 2 * (k - 1);
            hipLaunchKernelGGL(topo_border_nodes_kernel,
                dim3((unsigned)((ctx->num_nodes + 255) / 256)), dim3(256), 0,
                ctx->stream, P);
            int const mrc = launch_patch_mse(ctx, P, true);
            if (mrc != SMVS_OK)
                return mrc;
            hipLaunchKernelGGL(topo_cut_patches_kernel,
                dim3((unsigned)((ctx->num_patches + 255) / 256)), dim3(256), 0,
                ctx->stream, P);
            hipLaunchKernelGGL(topo_cut_nodes_kernel,
                dim3((unsigned)((ctx->num_nodes + 255) / 256)), dim3(256), 0,
                ctx->stream, P);
        }
        SMVS_HIP_CHECK(hipGetLastError());
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host + I_TOPO_PASS0,
            ctx->status + I_TOPO_PASS0, 2 * TOPO_AHEAD * sizeof(int), hipMemcpyDeviceToHost,
            ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < ahead && deleted > 10; ++k) {
            deleted = ctx->status_host[I_TOPO_PASS0 + 2 * k];
            total += deleted;
            if (trace)
                std::fprintf(stderr, "[smvs topo] cut pass: %d of %d patches evaluated, %d deleted\n",
                    ctx->status_host[I_TOPO_PASS0 + 2 * k + 1], ctx->num_patches, deleted);
        }
    }
    if (patch_valid_out != nullptr)
        SMVS_HIP_CHECK(hipMemcpyAsync(patch_valid_out, ctx->patch_valid,
            (size_t)ctx->num_patches, hipMemcpyDeviceToHost, ctx->stream));
    if (node_valid_out != nullptr)
        SMVS_HIP_CHECK(hipMemcpyAsync(node_valid_out, ctx->node_valid,
            (size_t)ctx->num_nodes, hipMemcpyDeviceToHost, ctx->stream));
    if (patch_valid_out != nullptr || node_valid_out != nullptr)
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (total_deleted != nullptr)
        *total_deleted = total;
    return SMVS_OK;
}
