// Preconditioned conjugate gradient on the block-stencil matrix.
//
// Replaces ConjugateGradient::solve (reference: lib/conjugate_gradient.h:72-202),
// BlockSparseMatrix<4>::multiply (lib/block_sparse_matrix.h:276-298) and the
// SSEVector kernels (lib/sse_vector.cc).  The node grid is regular, so the
// block-CSC matrix of the reference becomes a 9-point stencil of 4x4 blocks.
// H is symmetric block for block, so only the diagonal and the four "upper"
// neighbour slots are stored, slot-major H[5][node][16]: no index arrays,
// fully coalesced; the lower slots are read as transposes from the neighbour.
//
// Two kernels per iteration, no atomics, no intra-kernel fences:
//   A_k  finishes iteration k-1 (termination tests, beta), forms
//        d_k = z + beta d_{k-1}, computes Ad_k and the d.Ad partials;
//   B_k  alpha = rr / d.Ad; x += alpha d; r -= alpha Ad; z = P r; partials of
//        r.r, x.(b + r), z.r.
// Every block re-reduces the (<= 512) per-block partials of the previous
// kernel in the same fixed order, so all blocks derive bit-identical scalars
// and the result is independent of scheduling.  The CG state (rr, Q0, iter,
// done) is double-buffered: block 0 of A_k writes bank k&1 while the other
// blocks read bank (k-1)&1.  The host never synchronises inside a solve: it
// paces its launches on progress words the kernels publish in pinned host
// memory (cg_solve_launch).
#include "common.h"

#include <chrono>
#include <cmath>
#include <cstdlib>

namespace smvs_hip {

constexpr int CG_THREADS = 512;
constexpr int CG_MAX_BLOCKS = 512;

typedef double double4_v __attribute__((ext_vector_type(4)));

// broadcast lane (CTRL & 3) of every quad to the quad (DPP quad_perm)
template <int CTRL>
__device__ __forceinline__ double
quad_bcast(double v)
{
    long long const bits = __double_as_longlong(v);
    int lo = (int)(bits & 0xFFFFFFFFll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

struct CgState {
    double rr;       // z.r (r_dot_r of the reference)
    double q0;
    double tol;
    double gnorm;
    int iter;        // iteration this state belongs to
    int done;
    int info;
    int pad;
};

struct CgArgs {
    const double *H9;
    const double *Pinv;
    const double *g;
    const uint8_t *active;   // rows / columns of inactive nodes are zero
    uint16_t *mask;          // bit s: stencil slot s of the node is in the matrix
    double *x, *r, *z, *Ad, *b;
    double *dbuf[2];
    double *partials;      // [2][CG_PARTIAL_ROWS][CG_MAX_BLOCKS]: high words, low words
    CgState *state;        // [2]
    int *status;
    int *progress;         // pinned host memory, see cg_solve_launch
    int solve_tag;         // solve id << 16
    int num_nodes, stride, rows_total;
    int k;                 // iteration index of this launch
    int max_iterations;
    double q_tolerance;
    double fixed_tolerance;  // < 0: 0.01 * ||g||
};

// Progress words in pinned host memory (the host paces its launches on them
// instead of synchronising): [0] = tag | k once A_k has settled the solver
// state, [1] = tag | 1 once the solve is finished, [2] = info, [3] = the
// iteration count.  The tag (solve id << 16) keeps late writes of an earlier
// solve's trailing no-op launches from being mistaken for this solve's.
__device__ __forceinline__ void
publish_progress(CgArgs const &A, int done, int info, int iter)
{
    if (done) {
        __hip_atomic_store(A.progress + 2, info, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.progress + 3, iter, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.progress + 1, A.solve_tag | 1, __ATOMIC_RELEASE,
            __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(A.progress + 0, A.solve_tag | A.k, __ATOMIC_RELEASE,
        __HIP_MEMORY_SCOPE_SYSTEM);
}

// The dot products: every product is rounded to double as in the reference
// (sse_vector.cc:215-262), and the SUM of the rounded products is exact --
// accumulated in twice the working precision (TwoSum: the exact error of
// every addition collected in a second double, combined across lanes, waves
// and blocks in the same arithmetic).  The value every block derives is the
// correctly rounded sum of the reference's own products: it no longer depends
// on how the elements are dealt to threads and blocks, and it differs from the
// reference's sequential sum by that sum's own accumulated rounding only.
// Why: the block partition is a property of the launch (grid size, XCD bands),
// and the sums should not be.  What it does NOT buy is the oracle's iteration
// count on the ill-conditioned systems of the fuzz sweep: there the count flips
// between two exits (68 / 61 iterations) when g moves by 1e-12 -- in the
// oracle's own solve too (tools/cg_association.py, profiles/r5_cg_association.txt,
// tests/test_oracle_solver_math.py) -- and the device's g is 1e-12 from the
// oracle's.  The streaming kernels are bound by memory latency, the extra
// arithmetic is free.
struct DD {
    double hi, lo;
};

// s += a * b, the product rounded once
__device__ __forceinline__ void
dd_fma(DD &s, double a, double b)
{
#pragma clang fp contract(off)
    double const p = a * b;
    double const t = s.hi + p;
    double const zz = t - s.hi;
    double const te = (s.hi - (t - zz)) + (p - zz);
    s.hi = t;
    s.lo += te;
}

// (TwoSum's error term is the exact error of the rounded sum, whichever
// operand comes first: dd_add is commutative to the bit, so a butterfly leaves
// the same value in every lane)
__device__ __forceinline__ DD
dd_add(DD a, DD b)
{
#pragma clang fp contract(off)
    double const t = a.hi + b.hi;
    double const zz = t - a.hi;
    double const te = (a.hi - (t - zz)) + (b.hi - zz);
    DD r;
    r.hi = t;
    r.lo = (a.lo + b.lo) + te;
    return r;
}

__device__ __forceinline__ DD
dd_shfl_xor(DD v, int off)
{
    DD r;
    r.hi = __shfl_xor(v.hi, off);
    r.lo = __shfl_xor(v.lo, off);
    return r;
}

// rows of the partial buffer: the high words in rows 0 .. CG_PARTIAL_ROWS - 1,
// the low words CG_PARTIAL_ROWS rows behind.  Rows in use: 0 (d.Ad), 1-3 (r.r,
// x.(b + r), z.r), 4-5 (the init kernel's z.r, g.g); A_1 loads three rows from
// row 4 on (branch-free: it reads row 6 and ignores it), so eight rows.
constexpr int CG_PARTIAL_ROWS = 8;

// Sum NV arrays of `nb` (<= CG_THREADS) per-block partials in a fixed order;
// every thread of every block gets the same values.  The loads are split
// from the reduction so that a kernel can request them together with its
// other operands and pay one memory round trip instead of two.
template <int NV>
__device__ __forceinline__ void
load_partials(const double *partials, int nb, DD (&mine)[NV])
{
#pragma unroll
    // branch-free and unmasked (reduce_loaded masks): the loads stay
    // countable for s_waitcnt and nothing waits for them here
    for (int k = 0; k < NV; ++k) {
        size_t const at = (size_t)k * CG_MAX_BLOCKS + min((int)threadIdx.x, nb - 1);
        mine[k].hi = partials[at];
        mine[k].lo = partials[at + (size_t)CG_PARTIAL_ROWS * CG_MAX_BLOCKS];
    }
}

template <int NV>
__device__ __forceinline__ void
reduce_loaded(DD (&mine)[NV], int nb, double (&out)[NV])
{
    __shared__ DD red[NV][CG_THREADS / 64];
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        DD s = { 0.0, 0.0 };
        if ((int)threadIdx.x < nb)
            s = mine[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s = dd_add(s, dd_shfl_xor(s, off));
        if (lane == 0)
            red[k][wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        DD s = red[k][0];
#pragma unroll
        for (int wv = 1; wv < CG_THREADS / 64; ++wv)
            s = dd_add(s, red[k][wv]);
        out[k] = s.hi + s.lo;
    }
    __syncthreads();
}

template <int NV>
__device__ __forceinline__ void
reduce_partials(const double *partials, int nb, double (&out)[NV])
{
    DD mine[NV];
    load_partials<NV>(partials, nb, mine);
    reduce_loaded<NV>(mine, nb, out);
}

// Block-level sum of per-thread values -> partials[k][blockIdx.x]
template <int NV>
__device__ __forceinline__ void
store_partials(DD (&v)[NV], double *partials)
{
    __shared__ DD red[NV][CG_THREADS / 64];
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        DD s = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s = dd_add(s, dd_shfl_xor(s, off));
        if (lane == 0)
            red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            DD s = red[k][0];
#pragma unroll
            for (int wv = 1; wv < CG_THREADS / 64; ++wv)
                s = dd_add(s, red[k][wv]);
            size_t const at = (size_t)k * CG_MAX_BLOCKS + blockIdx.x;
            partials[at] = s.hi;
            partials[at + (size_t)CG_PARTIAL_ROWS * CG_MAX_BLOCKS] = s.lo;
        }
    }
}

// b = -g, x = 0, r = b, z = P r, d_1 = z; partials of z.r and g.g
__global__ void __launch_bounds__(CG_THREADS)
cg_init_kernel(CgArgs A)
{
    int const items = A.num_nodes * 4;
    DD v[2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
    for (int gid = blockIdx.x * CG_THREADS + threadIdx.x; gid < items;
         gid += gridDim.x * CG_THREADS) {
        int const n = gid >> 2, row = gid & 3;
        double const gi = A.g[gid];
        double const bi = -gi;
        A.b[gid] = bi;
        A.x[gid] = 0.0;
        A.r[gid] = bi;
        const double *gn = A.g + (size_t)n * 4;
        const double *P = A.Pinv + (size_t)n * 16 + row * 4;
        double zi;
        {
            // block_sparse_matrix.h:289-295 accumulation order
#pragma clang fp contract(off)
            zi = 0.0;
            zi += P[0] * (-gn[0]);
            zi += P[1] * (-gn[1]);
            zi += P[2] * (-gn[2]);
            zi += P[3] * (-gn[3]);
        }
        A.z[gid] = zi;
        A.dbuf[1][gid] = zi;
        dd_fma(v[0], zi, bi);
        dd_fma(v[1], gi, gi);
        if (row == 0) {
            // Inactive nodes have no row and no column in the reference's
            // matrix (gauss_newton_step.cc:91-105): their b, r, z, d and x
            // stay zero, and nothing is read for them.
            unsigned mask = 0u;
            if (A.active == nullptr || A.active[n] != 0) {
                int const ix = n % A.stride;
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    int const dx = s % 3 - 1, dy = s / 3 - 1;
                    int const mx = ix + dx;
                    int const m = n + dy * A.stride + dx;
                    if (mx < 0 || mx >= A.stride || m < 0 || m >= A.num_nodes)
                        continue;
                    if (A.active != nullptr && !A.active[m])
                        continue;
                    mask |= 1u << s;
                }
            }
            A.mask[n] = (uint16_t)mask;
        }
    }
    // rows 4, 5 of the partial buffer: A_1 reduces them while its fast
    // blocks already write row 0
    store_partials<2>(v, A.partials + 4 * CG_MAX_BLOCKS);
}

// A_k: finish iteration k-1, form d_k, Ad_k = A d_k, partial d.Ad.
// One thread per (node, row) of a 16 x 8 node tile; a wave reads 16
// consecutive nodes' rows of one slot as one contiguous 2 KiB segment.  The
// direction d_k = z_k + beta d_{k-1} of the tile and its one-node halo is
// formed once and shared through LDS.  The product follows the reference's
// accumulation order (ascending block column, then column inside the block,
// separate mul/add) and is bit-identical to BlockSparseMatrix::multiply.
//
// A launch is a chain of dependent memory round trips (a few microseconds
// each at this size), not a bandwidth problem, so the kernel needs three:
// {solver state, partial sums, stencil masks of both tiles}, then the first
// tile's matrix rows and direction operands (requested before the reduction
// of the partial sums), then the second tile's.  All loads are unconditional
// (absent operands are redirected to a valid, already cached address): the
// tile body is straight-line code whose loads are in flight together.
constexpr int SPMV_TX = 16, SPMV_TY = CG_THREADS / 4 / SPMV_TX;
constexpr int SPMV_HALO = 2 * (SPMV_TX + 2) + 2 * SPMV_TY;

__global__ void __launch_bounds__(CG_THREADS, 4)
cg_spmv_kernel(CgArgs A, int nb)
{
    constexpr int TX = SPMV_TX, TY = SPMV_TY, LW = TX + 2;
    __shared__ double dtile[(TY + 2) * LW][4];
    const double *__restrict__ d_old = A.dbuf[(A.k - 1) & 1];
    double *__restrict__ d_new = A.dbuf[A.k & 1];
    const double *__restrict__ H = A.H9;
    double *__restrict__ Ad = A.Ad;
    bool const first = A.k == 1;
    unsigned const Nu = (unsigned)A.num_nodes;
    // Tiles are handed out in row-major bands per XCD (blockIdx % 8) so that
    // the transposed blocks stored at vertical neighbours across a tile edge
    // mostly come from the same XCD's L2.
    // (8 bands, or one for grids of fewer than 8 blocks; shifts and 32-bit
    // divisions only: this prologue is on every launch's critical path)
    unsigned const rows_total = A.rows_total;
    unsigned const tiles_x = ((unsigned)A.stride + TX - 1) / TX;
    unsigned const tiles_y = (rows_total + TY - 1) / TY;
    unsigned const num_tiles = tiles_x * tiles_y;
    unsigned const gshift = gridDim.x >= 8u ? 3u : 0u;
    unsigned const groups = 1u << gshift;
    unsigned const xcd = blockIdx.x & (groups - 1u);
    unsigned const slot_in_xcd = blockIdx.x >> gshift;
    int const per_xcd_blocks
        = (int)((gridDim.x + groups - 1u - xcd) >> gshift);
    int const band_begin = (int)((num_tiles * xcd) >> gshift);
    int const band_end = (int)((num_tiles * (xcd + 1u)) >> gshift);
    int const tid = threadIdx.x;
    int const lnode = tid >> 2, row = tid & 3;
    int const lx = lnode & (TX - 1), ly = lnode / TX;
    int const lcore = (ly + 1) * LW + lx + 1;
    // halo position served by this thread (threads 0 .. 4 * SPMV_HALO - 1)
    bool const has_halo = lnode < SPMV_HALO;
    int hx, hy;
    if (lnode < LW) {
        hx = lnode; hy = 0;
    } else if (lnode < 2 * LW) {
        hx = lnode - LW; hy = TY + 1;
    } else if (lnode < 2 * LW + TY) {
        hx = 0; hy = lnode - 2 * LW + 1;
    } else {
        hx = TX + 1; hy = lnode - 2 * LW - TY + 1;
    }
    int const lhalo = has_halo ? hy * LW + hx : 0;

    // Node of this thread in a tile (0 and in_grid = false outside the grid)
    auto tile_node = [&](int tile, bool &in_grid) -> int {
        unsigned const ty = (unsigned)tile / tiles_x;
        unsigned const tx = (unsigned)tile - ty * tiles_x;
        int const ix = (int)tx * TX + lx, iy = (int)ty * TY + ly;
        in_grid = tile < band_end && ix < A.stride && iy < (int)rows_total;
        return in_grid ? iy * A.stride + ix : 0;
    };
    // halo node of this thread in a tile (-1: none)
    auto halo_node = [&](int tile) -> int {
        unsigned const ty = (unsigned)tile / tiles_x;
        unsigned const tx = (unsigned)tile - ty * tiles_x;
        int const ix = (int)tx * TX + hx - 1, iy = (int)ty * TY + hy - 1;
        bool const ok = has_halo && tile < band_end && ix >= 0
            && ix < A.stride && iy >= 0 && iy < (int)rows_total;
        return ok ? iy * A.stride + ix : -1;
    };

    // first round trip: stencil masks of this block's first two tiles, the
    // partial sums and the solver state, all requested before any is used
    int tile = band_begin + slot_in_xcd;
    bool grid_cur, grid_next;
    int n_cur = tile_node(tile, grid_cur);
    int n_next = tile_node(tile + per_xcd_blocks, grid_next);
    unsigned const raw_cur = A.mask[n_cur], raw_next = A.mask[n_next];
    // (A_1 reduces the two sums of the init kernel: z.r and g.g)
    DD mine[3];
    load_partials<3>(A.partials + (first ? 4 : 1) * CG_MAX_BLOCKS, nb, mine);
    // A launch after convergence is a no-op.  It takes no early exit here (a
    // branch would serialise the state load ahead of everything else): its
    // operand requests are all redirected to one cached line instead.
    // (A_1 has no predecessor state: it builds the initial one below.)
    CgState prev = A.state[(A.k - 1) & 1];
    bool const idle = !first && prev.done != 0;
    // stencil mask: 0 outside the grid, 0x8000 in the grid but inactive
    unsigned mask_cur = grid_cur && !idle ? (0x8000u | raw_cur) : 0u;
    unsigned mask_next = grid_next && !idle ? (0x8000u | raw_next) : 0u;
    n_cur = idle ? 0 : n_cur;

    // in the first iteration d_1 = z was written by the init kernel
    const double *__restrict__ dir_a = first ? d_new : A.z;
    const double *__restrict__ dir_b = first ? d_new : d_old;

    // Operands of one work item; none depends on beta, so the first tile's
    // are requested before the reduction.
    struct Operands {
        double h[9][4];
        double za, zb, ha, hb;   // direction operands: own node, halo node
    };
    auto load_operands = [&](int tile_, int n, unsigned mask, Operands &o) {
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            bool const present = (mask >> s) & 1u;
            int const dx = s % 3 - 1, dy = s / 3 - 1;
            int const m = n + dy * A.stride + dx;
            // symmetric storage: slots 4..8 are stored at the node itself,
            // slots 0..3 are the transposed blocks stored at the neighbour
            // (32-bit element offsets from the uniform base: H has fewer
            // than 2^32 elements)
            if (s >= 4) {
                unsigned const off = ((unsigned)(present ? s - 4 : 0) * Nu
                    + (unsigned)n) * 16u + (unsigned)row * 4u;
                double4_v const hrow
                    = *reinterpret_cast<const double4_v *>(H + off);
                o.h[s][0] = hrow.x; o.h[s][1] = hrow.y;
                o.h[s][2] = hrow.z; o.h[s][3] = hrow.w;
            } else {
                unsigned const off = present
                    ? ((unsigned)(4 - s) * Nu + (unsigned)m) * 16u + (unsigned)row
                    : (unsigned)n * 16u + (unsigned)row;
                const double *blk = H + off;
                o.h[s][0] = blk[0]; o.h[s][1] = blk[4];
                o.h[s][2] = blk[8]; o.h[s][3] = blk[12];
            }
        }
        unsigned const own_off = (unsigned)n * 4u + (unsigned)row;
        o.za = dir_a[own_off];
        o.zb = dir_b[own_off];
        int const hn = idle ? -1 : halo_node(tile_);
        unsigned const halo_off = (unsigned)(hn < 0 ? n : hn) * 4u
            + (unsigned)row;
        o.ha = dir_a[halo_off];
        o.hb = dir_b[halo_off];
        if (hn < 0) {
            o.ha = 0.0;
            o.hb = 0.0;
        }
    };
    Operands op;
    load_operands(tile, n_cur, mask_cur, op);

    if (idle) {
        if (blockIdx.x == 0 && tid == 0) {
            A.state[A.k & 1] = prev;
            publish_progress(A, 0, 0, 0);
        }
        return;
    }
    double beta = 0.0;
    if (!first) {
        // termination tests of iteration k-1 (conjugate_gradient.h:136-198)
        double v[3];
        reduce_loaded<3>(mine, nb, v);
        double const new_rr = v[0];
        double const Q1 = -1.0 * v[1];
        int const it = prev.iter;  // == k - 1
        int done = 0, info = SMVS_CG_MAX_ITERATIONS, iter_out = it + 1;
        if (new_rr < prev.tol) {
            done = 1; info = SMVS_CG_CONVERGENCE; iter_out = it;
        } else {
            double const zeta = it * (Q1 - prev.q0) / Q1;
            if (zeta < A.q_tolerance) {
                done = 1; info = SMVS_CG_CONVERGENCE; iter_out = it;
            } else if (it + 1 >= A.max_iterations) {
                done = 1;  // loop ran out: CG_MAX_ITERATIONS, count = max
            }
        }
        beta = v[2] / prev.rr;
        if (blockIdx.x == 0 && tid == 0) {
            CgState s = prev;
            s.rr = v[2];
            s.q0 = Q1;
            s.iter = iter_out;
            s.done = done;
            s.info = info;
            A.state[A.k & 1] = s;
            if (done) {
                A.status[I_DONE] = 1;
                A.status[I_INFO] = info;
                A.status[I_ITER] = iter_out;
            }
            publish_progress(A, done, info, iter_out);
        }
        if (done)
            return;
    } else {
        // initial state: x = 0, r = b = -g, z = P r, d = z
        // (conjugate_gradient.h:86-118)
        double v[2];
        DD init[2] = { mine[0], mine[1] };
        reduce_loaded<2>(init, nb, v);
        prev.rr = v[0];
        prev.q0 = -0.0;  // -1.0 * x.(b + r) with x = 0
        prev.gnorm = sqrt(v[1]);
        prev.tol = A.fixed_tolerance < 0.0 ? prev.gnorm * 0.01
            : A.fixed_tolerance;
        prev.iter = 1;
        // loop condition `num_iterations < max_iterations` fails at once
        prev.done = A.max_iterations <= 1 ? 1 : 0;
        prev.info = SMVS_CG_MAX_ITERATIONS;
        prev.pad = 0;
        if (blockIdx.x == 0 && tid == 0) {
            A.state[1] = prev;
            A.state[0] = prev;
            A.status[I_DONE] = prev.done;
            A.status[I_INFO] = prev.info;
            A.status[I_ITER] = 1;
            publish_progress(A, prev.done, prev.info, 1);
        }
        if (prev.done)
            return;
    }

    DD v[1] = { { 0.0, 0.0 } };
    bool first_tile = true;
    for (; tile < band_end; tile += per_xcd_blocks) {
        if (!first_tile) {
            n_cur = n_next;
            mask_cur = mask_next;
            load_operands(tile, n_cur, mask_cur, op);
            bool grid;
            n_next = tile_node(tile + per_xcd_blocks, grid);
            unsigned const raw = A.mask[n_next];
            mask_next = grid ? (0x8000u | raw) : 0u;
            __syncthreads();   // the previous tile's readers are done
        }
        first_tile = false;
        int const n = n_cur;
        int const gid = n * 4 + row;
        double own, halo;
        {
#pragma clang fp contract(off)
            own = first ? op.za : op.za + beta * op.zb;
            halo = first ? op.ha : op.ha + beta * op.hb;
        }
        dtile[lcore][row] = own;
        if (has_halo)
            dtile[lhalo][row] = halo;
        __syncthreads();
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            bool const present = (mask_cur >> s) & 1u;
            int const dx = s % 3 - 1, dy = s / 3 - 1;
            double4_v const dm = *reinterpret_cast<const double4_v *>(
                &dtile[lcore + dy * LW + dx][0]);
            double t = acc;
            {
#pragma clang fp contract(off)
                t += op.h[s][0] * dm.x;
                t += op.h[s][1] * dm.y;
                t += op.h[s][2] * dm.z;
                t += op.h[s][3] * dm.w;
            }
            acc = present ? t : acc;
        }
        if (mask_cur != 0u) {
            // inactive nodes: no row in the matrix, d stays zero
            double const d_own = (mask_cur & 0x10u) ? own : 0.0;
            Ad[gid] = acc;
            if (!first)
                d_new[gid] = d_own;
            dd_fma(v[0], d_own, acc);
        }
    }
    store_partials<1>(v, A.partials);
}

// B_k: x += alpha d; r -= alpha Ad; z = P r; partials of r.r, x.(b + r), z.r
// All operands of the (at most two) rounds are requested together with the
// d.Ad partials: one memory round trip per launch.
__global__ void __launch_bounds__(CG_THREADS)
cg_update_kernel(CgArgs A, int nb)
{
    const double *d = A.dbuf[A.k & 1];
    int const items = A.num_nodes * 4;
    // every thread runs the same number of rounds so the DPP exchanges stay
    // convergent
    int const rounds = (items + gridDim.x * CG_THREADS - 1)
        / (gridDim.x * CG_THREADS);
    struct Item {
        bool in_range;
        unsigned mask;
        double x, d, r, Ad, b;
        double4_v P;
    };
    // A launch after convergence takes no early exit (the branch would put
    // the state load ahead of everything else): its requests all go to the
    // first line of each array instead.
    CgState const st = A.state[A.k & 1];
    bool const idle = st.done != 0;
    auto load_item = [&](int round, Item &it) {
        int const gid = (round * gridDim.x + blockIdx.x) * CG_THREADS
            + threadIdx.x;
        bool const inside = gid < items && !idle;
        unsigned const at = inside ? (unsigned)gid : (unsigned)(gid & 3);
        unsigned const n = at >> 2, row = at & 3u;
        it.in_range = inside;
        it.mask = A.mask[n];
        it.x = A.x[at];
        it.d = d[at];
        it.r = A.r[at];
        it.Ad = A.Ad[at];
        it.b = A.b[at];
        it.P = *reinterpret_cast<const double4_v *>(A.Pinv + n * 16u + row * 4u);
    };
    // (rounds beyond the grid are redirected like idle ones: no branches
    // between the requests)
    constexpr int PRE = 2;
    DD mine[1];
    double dad[1];
    load_partials<1>(A.partials, nb, mine);
    Item pre[PRE];
#pragma unroll
    for (int i = 0; i < PRE; ++i)
        load_item(i, pre[i]);
    if (idle)
        return;
    reduce_loaded<1>(mine, nb, dad);
    double const alpha = st.rr / dad[0];
    DD v[3] = { { 0.0, 0.0 }, { 0.0, 0.0 }, { 0.0, 0.0 } };
    auto process = [&](int round, Item const &item) {
        int const gid = (round * gridDim.x + blockIdx.x) * CG_THREADS
            + threadIdx.x;
        bool const in_range = item.in_range && (item.mask & 0x1FFu) != 0u;
        double xi = 0.0, ri = 0.0;
        if (in_range) {
#pragma clang fp contract(off)
            xi = item.x + alpha * item.d;
            ri = item.r - alpha * item.Ad;
        }
        // z = P r needs the node's whole residual: the four row-lanes of a
        // node are neighbours in the wave.
        double rn[4];
        rn[0] = quad_bcast<0x00>(ri);
        rn[1] = quad_bcast<0x55>(ri);
        rn[2] = quad_bcast<0xAA>(ri);
        rn[3] = quad_bcast<0xFF>(ri);
        if (in_range) {
            A.x[gid] = xi;
            A.r[gid] = ri;
            double zi;
            {
#pragma clang fp contract(off)
                zi = 0.0;
                zi += item.P.x * rn[0];
                zi += item.P.y * rn[1];
                zi += item.P.z * rn[2];
                zi += item.P.w * rn[3];
            }
            A.z[gid] = zi;
            double tmp;
            {
                // conjugate_gradient.h:176-178: tmp = b + r, rounded, then x . tmp
#pragma clang fp contract(off)
                tmp = item.b + ri;
            }
            dd_fma(v[0], ri, ri);
            dd_fma(v[1], xi, tmp);
            dd_fma(v[2], zi, ri);
        }
    };
#pragma unroll
    for (int i = 0; i < PRE; ++i)
        process(i, pre[i]);
    for (int round = PRE; round < rounds; ++round) {
        Item item;
        load_item(round, item);
        process(round, item);
    }
    store_partials<3>(v, A.partials + CG_MAX_BLOCKS);
}

int
cg_solve_launch(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info)
{
    // conjugate_gradient.h:123-125: the iteration counter starts at 1 and the
    // loop runs while it is below max_iterations.  The launch index shares a
    // 32-bit progress word with the solve id (16 bits each).
    SMVS_REQUIRE(max_iterations >= 0 && max_iterations <= 0xFFFF,
        "max_iterations must be in [0, 65535]");
    SMVS_REQUIRE(q_tolerance >= 0.0, "negative q_tolerance");
    {
        // whole solve in one launch with H resident in registers when the
        // node grid fits the chip (cg_resident.hip); otherwise, or when its
        // workgroups could not all be resident, the streaming kernels below
        bool ran = false;
        int const rc = cg_resident_solve(ctx, max_iterations, error_tolerance,
            q_tolerance, num_iterations, info, &ran);
        if (rc != SMVS_OK || ran)
            return rc;
    }
    CgArgs A;
    A.H9 = ctx->H9;
    A.Pinv = ctx->Pinv;
    A.g = ctx->g;
    A.active = ctx->cg_use_active ? ctx->active : nullptr;
    A.mask = ctx->cg_mask;
    A.x = ctx->x;
    A.r = ctx->r;
    A.z = ctx->z;
    A.Ad = ctx->Ad;
    A.b = ctx->b;
    A.dbuf[0] = ctx->d;
    A.dbuf[1] = ctx->d2;
    A.partials = ctx->partials;
    A.state = reinterpret_cast<CgState *>(ctx->cg_state);
    A.status = ctx->status;
    A.progress = ctx->cg_progress;
    ctx->cg_solve_id = (ctx->cg_solve_id + 1) & 0x7FFF;
    if (ctx->cg_solve_id == 0)
        ctx->cg_solve_id = 1;
    A.solve_tag = ctx->cg_solve_id << 16;
    A.num_nodes = ctx->num_nodes;
    A.stride = ctx->node_stride;
    A.rows_total = ctx->num_nodes / ctx->node_stride;
    A.k = 0;
    A.max_iterations = max_iterations;
    A.q_tolerance = q_tolerance;
    A.fixed_tolerance = error_tolerance;

    // the kernels index H with 32-bit element offsets
    if ((size_t)ctx->num_nodes * 5 * 16 >= (size_t)1 << 32) {
        set_error("cg_solve_launch: %d nodes exceed the 32-bit offsets of the "
            "SpMV kernel", ctx->num_nodes);
        return SMVS_ERR_INVALID;
    }
    size_t const items = (size_t)ctx->num_nodes * 4;
    int nb = (int)((items + CG_THREADS - 1) / CG_THREADS);
    if (nb > CG_MAX_BLOCKS)
        nb = CG_MAX_BLOCKS;
    {
        ScopedKernelTimer timer(ctx, SMVS_K_CG_INIT);
        hipLaunchKernelGGL(cg_init_kernel, dim3(nb), dim3(CG_THREADS), 0,
            ctx->stream, A);
    }
    SMVS_HIP_CHECK(hipGetLastError());

    if (max_iterations <= 1) {
        // The loop body never runs (conjugate_gradient.h:121-125, 198-199):
        // x = 0, one "iteration", CG_MAX_ITERATIONS.  x was zeroed above.
        ctx->last_cg_iterations = 1;
        if (num_iterations != nullptr)
            *num_iterations = 1;
        if (info != nullptr)
            *info = SMVS_CG_MAX_ITERATIONS;
        return SMVS_OK;
    }

    // A_k for k = 1 .. max_iterations (A_max only finishes iteration max-1),
    // B_k for k = 1 .. max_iterations - 1.
    // The host does not synchronise: the kernels publish their progress in
    // pinned host memory and the host stays AHEAD launches in front of the
    // last iteration it has seen settle.  Launches that run after convergence
    // are no-ops, at most AHEAD of them per solve.
    constexpr int AHEAD = 2;
    volatile int *progress = ctx->cg_progress;
    int k = 1;
    auto const t_start = std::chrono::steady_clock::now();
    long spins = 0;
    for (;;) {
        int const done_word = __atomic_load_n(&progress[1], __ATOMIC_ACQUIRE);
        if (done_word == (A.solve_tag | 1))
            break;
        int const seen_word = __atomic_load_n(&progress[0], __ATOMIC_ACQUIRE);
        int const seen = (seen_word & ~0xFFFF) == A.solve_tag
            ? (seen_word & 0xFFFF) : 0;
        if (k <= max_iterations && k <= seen + AHEAD) {
            A.k = k;
            {
                ScopedKernelTimer timer(ctx, SMVS_K_CG_SPMV);
                hipLaunchKernelGGL(cg_spmv_kernel, dim3(nb),
                    dim3(CG_THREADS), 0, ctx->stream, A, nb);
            }
            if (k < max_iterations) {
                ScopedKernelTimer timer(ctx, SMVS_K_CG_UPDATE);
                hipLaunchKernelGGL(cg_update_kernel, dim3(nb),
                    dim3(CG_THREADS), 0, ctx->stream, A, nb);
            }
            ++k;
            continue;
        }
        if (k > max_iterations && seen >= max_iterations) {
            // A_max settles the state as done (the done word is published
            // before the progress word); reaching this means it did not
            if (__atomic_load_n(&progress[1], __ATOMIC_ACQUIRE)
                == (A.solve_tag | 1))
                break;
            set_error("cg_solve_launch: solver did not report completion");
            return SMVS_ERR_STATE;
        }
        __builtin_ia32_pause();
        if ((++spins & 0xFFFF) == 0) {
            SMVS_HIP_CHECK(hipGetLastError());
            // a device fault surfaces here instead of an endless wait
            hipError_t const q = hipStreamQuery(ctx->stream);
            if (q != hipSuccess && q != hipErrorNotReady)
                SMVS_HIP_CHECK(q);
            if (q == hipSuccess
                && __atomic_load_n(&progress[1], __ATOMIC_ACQUIRE)
                    != (A.solve_tag | 1)
                && __atomic_load_n(&progress[0], __ATOMIC_ACQUIRE)
                    == seen_word) {
                // the stream is idle, nothing is in flight and the solver
                // has not finished: the launches above were lost
                if (k > max_iterations) {
                    set_error("cg_solve_launch: solver did not report completion");
                    return SMVS_ERR_STATE;
                }
            }
            auto const dt = std::chrono::steady_clock::now() - t_start;
            if (dt > std::chrono::seconds(60)) {
                set_error("cg_solve_launch: timed out waiting for the device");
                return SMVS_ERR_STATE;
            }
        }
    }
    SMVS_HIP_CHECK(hipGetLastError());
    int const iters = progress[3];
    int const solve_info = progress[2];
    ctx->last_cg_iterations = iters;
    if (num_iterations != nullptr)
        *num_iterations = iters;
    if (info != nullptr)
        *info = solve_info;
    return SMVS_OK;
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_cg_solve(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_system) {
        set_error("smvs_cg_solve: no system constructed");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    return cg_solve_launch(ctx, max_iterations, error_tolerance, q_tolerance,
        num_iterations, info);
}

extern "C" int
smvs_cg_download_x(smvs_ctx *ctx, double *x)
{
    SMVS_REQUIRE(ctx && x, "null argument");
    if (!ctx->has_surface) {
        set_error("smvs_cg_download_x: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    SMVS_HIP_CHECK(hipMemcpyAsync(x, ctx->x,
        (size_t)ctx->num_nodes * 4 * sizeof(double), hipMemcpyDeviceToHost,
        ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_cg_upload_x(smvs_ctx *ctx, const double *x)
{
    SMVS_REQUIRE(ctx && x, "null argument");
    if (!ctx->has_surface) {
        set_error("smvs_cg_upload_x: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->x, x,
        (size_t)ctx->num_nodes * 4 * sizeof(double), hipMemcpyHostToDevice,
        ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}
