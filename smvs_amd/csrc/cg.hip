// Preconditioned conjugate gradient on the block-stencil matrix.
//
// Replaces ConjugateGradient::solve (reference: lib/conjugate_gradient.h:72-202),
// BlockSparseMatrix<4>::multiply (lib/block_sparse_matrix.h:276-298) and the
// SSEVector kernels (lib/sse_vector.cc).  The node grid is regular, so the
// block-CSC matrix of the reference becomes a 9-point stencil of 4x4 blocks,
// stored slot-major H9[slot][node][16]: no index arrays, fully coalesced.
//
// Two kernels per iteration, no atomics, no intra-kernel fences:
//   A_k  finishes iteration k-1 (termination tests, beta), forms
//        d_k = z + beta d_{k-1} on the fly, computes Ad_k and the d.Ad partials;
//   B_k  alpha = rr / d.Ad; x += alpha d; r -= alpha Ad; z = P r; partials of
//        r.r, x.(b + r), z.r.
// Every block re-reduces the (<= 1024) per-block partials of the previous
// kernel in the same fixed order, so all blocks derive bit-identical scalars
// and the result is independent of scheduling.  The CG state (rr, Q0, iter,
// done) is double-buffered: block 0 of A_k writes bank k&1 while the other
// blocks read bank (k-1)&1.  The host only polls a "done" word per chunk.
#include "common.h"

#include <cmath>

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
    double *x, *r, *z, *Ad, *b;
    double *dbuf[2];
    double *partials;      // [4][CG_MAX_BLOCKS]
    CgState *state;        // [2]
    int *status;
    int num_nodes, stride;
    int k;                 // iteration index of this launch
    int max_iterations;
    double q_tolerance;
    double fixed_tolerance;  // < 0: 0.01 * ||g||
};

// Sum NV arrays of `nb` per-block partials in a fixed order; every thread of
// every block gets the same values.
template <int NV>
__device__ __forceinline__ void
reduce_partials(const double *partials, int nb, double (&out)[NV])
{
    __shared__ double red[NV][CG_THREADS / 64];
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nb; i += CG_THREADS)
            s += partials[(size_t)k * CG_MAX_BLOCKS + i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s += __shfl_xor(s, off);
        if (lane == 0)
            red[k][wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < CG_THREADS / 64; ++wv)
            s += red[k][wv];
        out[k] = s;
    }
    __syncthreads();
}

// Block-level sum of per-thread values -> partials[k][blockIdx.x]
template <int NV>
__device__ __forceinline__ void
store_partials(double (&v)[NV], double *partials)
{
    __shared__ double red[NV][CG_THREADS / 64];
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s += __shfl_xor(s, off);
        if (lane == 0)
            red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = 0.0;
#pragma unroll
            for (int wv = 0; wv < CG_THREADS / 64; ++wv)
                s += red[k][wv];
            partials[(size_t)k * CG_MAX_BLOCKS + blockIdx.x] = s;
        }
    }
}

// b = -g, x = 0, r = b, z = P r, d_1 = z; partials of z.r and g.g
__global__ void __launch_bounds__(CG_THREADS)
cg_init_kernel(CgArgs A)
{
    int const items = A.num_nodes * 4;
    double v[2] = { 0.0, 0.0 };
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
        v[0] += zi * bi;
        v[1] += gi * gi;
    }
    store_partials<2>(v, A.partials);
}

__global__ void __launch_bounds__(CG_THREADS)
cg_init_finalize_kernel(CgArgs A, int nb)
{
    double v[2];
    reduce_partials<2>(A.partials, nb, v);
    if (threadIdx.x != 0)
        return;
    CgState s;
    s.rr = v[0];
    s.q0 = -0.0;  // -1.0 * x.(b + r) with x = 0
    s.gnorm = sqrt(v[1]);
    s.tol = A.fixed_tolerance < 0.0 ? s.gnorm * 0.01 : A.fixed_tolerance;
    s.iter = 1;
    // loop condition `num_iterations < max_iterations` fails at once
    s.done = A.max_iterations <= 1 ? 1 : 0;
    s.info = SMVS_CG_MAX_ITERATIONS;
    s.pad = 0;
    A.state[1] = s;
    A.state[0] = s;
    A.status[I_DONE] = s.done;
    A.status[I_INFO] = s.info;
    A.status[I_ITER] = 1;
}

// A_k: finish iteration k-1, form d_k, Ad_k = A d_k, partial d.Ad.
// One thread per (node, row); a wave reads 16 consecutive nodes' rows of one
// slot as one contiguous 2 KiB segment.  The product follows the reference's
// accumulation order (ascending block column, then column inside the block,
// separate mul/add) and is bit-identical to BlockSparseMatrix::multiply.
__global__ void __launch_bounds__(CG_THREADS)
cg_spmv_kernel(CgArgs A, int nb)
{
    CgState const prev = A.state[(A.k - 1) & 1];
    if (prev.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            A.state[A.k & 1] = prev;
        return;
    }
    double beta = 0.0;
    if (A.k > 1) {
        // termination tests of iteration k-1 (conjugate_gradient.h:136-198)
        double v[3];
        reduce_partials<3>(A.partials + CG_MAX_BLOCKS, nb, v);
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
        if (blockIdx.x == 0 && threadIdx.x == 0) {
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
        }
        if (done)
            return;
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.state[A.k & 1] = prev;
    }

    const double *__restrict__ d_old = A.dbuf[(A.k - 1) & 1];
    double *__restrict__ d_new = A.dbuf[A.k & 1];
    const double *__restrict__ H = A.H9;
    const double *__restrict__ zv = A.z;
    double *__restrict__ Ad = A.Ad;
    const uint8_t *__restrict__ act = A.active;
    bool const first = A.k == 1;
    int const items = A.num_nodes * 4;
    size_t const N = (size_t)A.num_nodes;
    double v[1] = { 0.0 };
    // 2-D tiles of 16 x (CG_THREADS / 64) nodes: with symmetric storage the
    // lower stencil slots re-read blocks stored at neighbouring nodes, which
    // stay in L1 / L2 when the neighbour belongs to the same tile.  Tiles
    // are handed out in row-major bands per XCD (blockIdx % 8) so vertical
    // neighbours across a tile edge mostly share an XCD's L2.
    constexpr int TX = 16, TY = CG_THREADS / 64;
    int const rows_total = A.num_nodes / A.stride;
    int const tiles_x = (A.stride + TX - 1) / TX;
    int const tiles_y = (rows_total + TY - 1) / TY;
    int const num_tiles = tiles_x * tiles_y;
    int const groups = min(8, (int)gridDim.x);
    int const xcd = blockIdx.x % groups, slot_in_xcd = blockIdx.x / groups;
    int const per_xcd_blocks = ((int)gridDim.x + groups - 1 - xcd) / groups;
    int const band_begin = (int)((long long)num_tiles * xcd / groups);
    int const band_end = (int)((long long)num_tiles * (xcd + 1) / groups);
    int const lnode = threadIdx.x >> 2, row = threadIdx.x & 3;
    int const lx = lnode & (TX - 1), ly = lnode / TX;
    (void)items;
    for (int tile = band_begin + slot_in_xcd; tile < band_end;
         tile += per_xcd_blocks) {
        int const tx = tile % tiles_x, ty = tile / tiles_x;
        int const ix = tx * TX + lx, iy = ty * TY + ly;
        if (ix >= A.stride || iy >= rows_total)
            continue;
        int const n = iy * A.stride + ix;
        int const gid = n * 4 + row;
        // Inactive nodes have no row and no column in the reference's matrix
        // (gauss_newton_step.cc:91-105): their b, r, z, d and x stay zero, so
        // nothing has to be read for them.
        if (act != nullptr && !act[n]) {
            Ad[gid] = 0.0;
            if (!first)
                d_new[gid] = 0.0;
            continue;
        }
        double acc = 0.0;
        double d_own = 0.0;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            int const dx = s % 3 - 1, dy = s / 3 - 1;
            int const mx = ix + dx;
            int const m = n + dy * A.stride + dx;
            if (mx < 0 || mx >= A.stride || m < 0 || m >= A.num_nodes)
                continue;
            if (act != nullptr && !act[m])
                continue;
            // d_k of the neighbour: every row-lane forms its own component
            // and the quad shares the four values by DPP broadcast (one
            // 8-byte load per lane instead of two 32-byte loads)
            double own;
            if (first)
                own = d_new[(size_t)m * 4 + row];
            else {
#pragma clang fp contract(off)
                own = zv[(size_t)m * 4 + row] + beta * d_old[(size_t)m * 4 + row];
            }
            double dm[4];
            dm[0] = quad_bcast<0x00>(own);
            dm[1] = quad_bcast<0x55>(own);
            dm[2] = quad_bcast<0xAA>(own);
            dm[3] = quad_bcast<0xFF>(own);
            if (s == 4)
                d_own = own;
            // symmetric storage: slots 4..8 are stored at the node itself,
            // slots 0..3 are the transposed blocks stored at the neighbour
            double h0, h1, h2, h3;
            if (s >= 4) {
                double4_v const hrow = *reinterpret_cast<const double4_v *>(
                    H + ((size_t)(s - 4) * N + n) * 16 + row * 4);
                h0 = hrow.x; h1 = hrow.y; h2 = hrow.z; h3 = hrow.w;
            } else {
                const double *blk = H + ((size_t)(4 - s) * N + m) * 16 + row;
                h0 = blk[0]; h1 = blk[4]; h2 = blk[8]; h3 = blk[12];
            }
            {
#pragma clang fp contract(off)
                acc += h0 * dm[0];
                acc += h1 * dm[1];
                acc += h2 * dm[2];
                acc += h3 * dm[3];
            }
        }
        Ad[gid] = acc;
        if (!first)
            d_new[gid] = d_own;
        v[0] += d_own * acc;
    }
    store_partials<1>(v, A.partials);
}

// B_k: x += alpha d; r -= alpha Ad; z = P r; partials of r.r, x.(b + r), z.r
__global__ void __launch_bounds__(CG_THREADS)
cg_update_kernel(CgArgs A, int nb)
{
    CgState const st = A.state[A.k & 1];
    if (st.done)
        return;
    double dad[1];
    reduce_partials<1>(A.partials, nb, dad);
    double const alpha = st.rr / dad[0];
    const double *d = A.dbuf[A.k & 1];
    int const items = A.num_nodes * 4;
    // every thread runs the same number of rounds so the shuffles stay
    // convergent
    int const rounds = (items + gridDim.x * CG_THREADS - 1)
        / (gridDim.x * CG_THREADS);
    double v[3] = { 0.0, 0.0, 0.0 };
    for (int round = 0; round < rounds; ++round) {
        int const gid = (round * gridDim.x + blockIdx.x) * CG_THREADS
            + threadIdx.x;
        int const n = gid >> 2, row = gid & 3;
        bool const in_range = gid < items
            && (A.active == nullptr || A.active[n] != 0);
        double xi = 0.0, ri = 0.0;
        if (in_range) {
#pragma clang fp contract(off)
            xi = A.x[gid] + alpha * d[gid];
            ri = A.r[gid] - alpha * A.Ad[gid];
        }
        // z = P r needs the node's whole residual: the four row-lanes of a
        // node are neighbours in the wave.
        int const base_lane = (threadIdx.x & 63) & ~3;
        double rn[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            rn[c] = __shfl(ri, base_lane + c);
        if (in_range) {
            A.x[gid] = xi;
            A.r[gid] = ri;
            const double *P = A.Pinv + (size_t)n * 16 + row * 4;
            double zi;
            {
#pragma clang fp contract(off)
                zi = 0.0;
                zi += P[0] * rn[0];
                zi += P[1] * rn[1];
                zi += P[2] * rn[2];
                zi += P[3] * rn[3];
            }
            A.z[gid] = zi;
            v[0] += ri * ri;
            v[1] += xi * (A.b[gid] + ri);
            v[2] += zi * ri;
        }
    }
    store_partials<3>(v, A.partials + CG_MAX_BLOCKS);
}

int
cg_solve_launch(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info)
{
    CgArgs A;
    A.H9 = ctx->H9;
    A.Pinv = ctx->Pinv;
    A.g = ctx->g;
    A.active = ctx->cg_use_active ? ctx->active : nullptr;
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
    A.num_nodes = ctx->num_nodes;
    A.stride = ctx->node_stride;
    A.k = 0;
    A.max_iterations = max_iterations;
    A.q_tolerance = q_tolerance;
    A.fixed_tolerance = error_tolerance;

    size_t const items = (size_t)ctx->num_nodes * 4;
    int nb = (int)((items + CG_THREADS - 1) / CG_THREADS);
    if (nb > CG_MAX_BLOCKS)
        nb = CG_MAX_BLOCKS;
    {
        ScopedKernelTimer timer(ctx, SMVS_K_CG_INIT);
        hipLaunchKernelGGL(cg_init_kernel, dim3(nb), dim3(CG_THREADS), 0,
            ctx->stream, A);
        hipLaunchKernelGGL(cg_init_finalize_kernel, dim3(1), dim3(CG_THREADS),
            0, ctx->stream, A, nb);
    }
    SMVS_HIP_CHECK(hipGetLastError());

    // A_k for k = 1 .. max_iterations (A_max only finishes iteration max-1),
    // B_k for k = 1 .. max_iterations - 1.
    // The host only learns the iteration count after the fact; kernels that
    // run after convergence are no-ops.  The first chunk is sized by the
    // previous solve of this context (iteration counts change slowly from one
    // Newton step to the next), later chunks are short.
    int chunk = ctx->last_cg_iterations > 0 ? ctx->last_cg_iterations + 1 : 8;
    int k = 1;
    bool done = false;
    while (!done) {
        for (int c = 0; c < chunk && k <= max_iterations; ++c, ++k) {
            A.k = k;
            {
                ScopedKernelTimer timer(ctx, SMVS_K_CG_SPMV);
                hipLaunchKernelGGL(cg_spmv_kernel, dim3(nb), dim3(CG_THREADS),
                    0, ctx->stream, A, nb);
            }
            if (k < max_iterations) {
                ScopedKernelTimer timer(ctx, SMVS_K_CG_UPDATE);
                hipLaunchKernelGGL(cg_update_kernel, dim3(nb),
                    dim3(CG_THREADS), 0, ctx->stream, A, nb);
            }
        }
        SMVS_HIP_CHECK(hipGetLastError());
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host, ctx->status,
            sizeof(int) * I_NUM, hipMemcpyDeviceToHost, ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        done = ctx->status_host[I_DONE] != 0 || k > max_iterations;
        chunk = 4;
    }
    if (ctx->status_host[I_DONE] == 0) {
        set_error("cg_solve_launch: solver did not report completion");
        return SMVS_ERR_STATE;
    }
    ctx->last_cg_iterations = ctx->status_host[I_ITER];
    if (num_iterations != nullptr)
        *num_iterations = ctx->status_host[I_ITER];
    if (info != nullptr)
        *info = ctx->status_host[I_INFO];
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
    SMVS_HIP_CHECK(hipSetDevice(ctx->device));
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
    SMVS_HIP_CHECK(hipSetDevice(ctx->device));
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
    SMVS_HIP_CHECK(hipSetDevice(ctx->device));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->x, x,
        (size_t)ctx->num_nodes * 4 * sizeof(double), hipMemcpyHostToDevice,
        ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}
