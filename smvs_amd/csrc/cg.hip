// Preconditioned conjugate gradient on the block-stencil matrix.
//
// Replaces ConjugateGradient::solve (reference: lib/conjugate_gradient.h:72-202),
// BlockSparseMatrix<4>::multiply (lib/block_sparse_matrix.h:276-298) and the
// SSEVector kernels (lib/sse_vector.cc).  The node grid is regular, so the
// block-CSC matrix of the reference becomes a 9-point stencil of 4x4 blocks,
// stored slot-major H9[slot][node][16]: no index arrays, fully coalesced.
// All scalars (alpha, beta, residual norms, the quadratic-model test) live on
// the device; the host only polls a "done" word once per chunk of iterations.
#include "common.h"

#include <cmath>

namespace smvs_hip {

constexpr int CG_THREADS = 256;

// ---- deterministic grid reductions --------------------------------------
// Every block stores its partial sums; the last block to arrive (ticket)
// adds the partials in block order, so results do not depend on scheduling.
template <int NV>
__device__ __forceinline__ bool
block_reduce_and_ticket(double (&v)[NV], double *partials, int max_blocks,
    int *ticket)
{
    __shared__ double red[NV][CG_THREADS / 64];
    __shared__ bool is_last;
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
            partials[(size_t)k * max_blocks + blockIdx.x] = s;
        }
        __threadfence();  // release the partials before taking the ticket
        int const t = atomicAdd(ticket, 1);
        is_last = (t == (int)gridDim.x - 1);
        if (is_last)
            *ticket = 0;
    }
    __syncthreads();
    if (!is_last)
        return false;
    __threadfence();      // acquire the other blocks' partials
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += CG_THREADS)
            s += __hip_atomic_load(&partials[(size_t)k * max_blocks + i],
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            v[k] = s;
        }
    }
    return threadIdx.x == 0;
}

struct CgArgs {
    const double *H9;
    const double *Pinv;
    const double *g;
    double *x, *r, *z, *Ad, *d, *b;
    double *partials;
    double *scalars;
    int *status;
    int num_nodes, stride, max_blocks;
    double q_tolerance;
    double fixed_tolerance;  // < 0: 0.01 * ||g||
};

// z = P r for one (node, row): block_sparse_matrix.h:289-295 accumulation order
__device__ __forceinline__ double
precond_row(const double *Pinv, const double *r, int n, int row)
{
#pragma clang fp contract(off)
    const double *v = Pinv + (size_t)n * 16 + row * 4;
    const double *rn = r + (size_t)n * 4;
    double s = 0.0;
    s += v[0] * rn[0];
    s += v[1] * rn[1];
    s += v[2] * rn[2];
    s += v[3] * rn[3];
    return s;
}

// b = -g, x = 0, r = b, z = P r, d = z; rr = z.r; tol from ||g||
__global__ void __launch_bounds__(CG_THREADS)
cg_init_kernel(CgArgs A)
{
    int const gid = blockIdx.x * blockDim.x + threadIdx.x;
    int const n = gid >> 2, row = gid & 3;
    double v[2] = { 0.0, 0.0 };
    if (n < A.num_nodes) {
        double const gi = A.g[gid];
        double const bi = -gi;
        A.b[gid] = bi;
        A.x[gid] = 0.0;
        A.r[gid] = bi;
        // z needs the whole r of the node: read g directly
        const double *gn = A.g + (size_t)n * 4;
        const double *P = A.Pinv + (size_t)n * 16 + row * 4;
        double zi;
        {
#pragma clang fp contract(off)
            zi = 0.0;
            zi += P[0] * (-gn[0]);
            zi += P[1] * (-gn[1]);
            zi += P[2] * (-gn[2]);
            zi += P[3] * (-gn[3]);
        }
        A.z[gid] = zi;
        A.d[gid] = zi;
        v[0] = zi * bi;
        v[1] = gi * gi;
    }
    if (block_reduce_and_ticket<2>(v, A.partials, A.max_blocks,
            &A.status[I_TICKET0])) {
        A.scalars[S_RR] = v[0];
        double const gnorm = sqrt(v[1]);
        A.scalars[S_GNORM] = gnorm;
        A.scalars[S_TOL] = A.fixed_tolerance < 0.0 ? gnorm * 0.01
            : A.fixed_tolerance;
        A.scalars[S_Q0] = -0.0;
        A.status[I_DONE] = 0;
        A.status[I_INFO] = SMVS_CG_MAX_ITERATIONS;
        A.status[I_ITER] = 1;
        // loop condition `num_iterations < max_iterations` fails at once
        if (A.status[I_MAXITER] <= 1)
            A.status[I_DONE] = 1;
        __threadfence();
    }
}

// Ad = A d (9-point block stencil), dAd partial sums.  One thread per
// (node, row); a wave reads 16 consecutive nodes' rows of one slot as one
// contiguous 2 KiB segment.  Accumulation follows the reference's order
// (ascending block column, then column inside the block, separate mul/add)
// so the product is bit-identical to BlockSparseMatrix::multiply.
__global__ void __launch_bounds__(CG_THREADS)
cg_spmv_kernel(CgArgs A)
{
    if (A.status[I_DONE])
        return;
    int const gid = blockIdx.x * blockDim.x + threadIdx.x;
    int const n = gid >> 2, row = gid & 3;
    double v[1] = { 0.0 };
    if (n < A.num_nodes) {
        int const ix = n % A.stride, iy = n / A.stride;
        size_t const N = (size_t)A.num_nodes;
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            int const dx = s % 3 - 1, dy = s / 3 - 1;
            int const mx = ix + dx;
            int const m = n + dy * A.stride + dx;
            if (mx < 0 || mx >= A.stride || m < 0 || m >= A.num_nodes)
                continue;
            const double *blk = A.H9 + ((size_t)s * N + n) * 16 + row * 4;
            const double *dm = A.d + (size_t)m * 4;
            double const h0 = blk[0], h1 = blk[1], h2 = blk[2], h3 = blk[3];
            {
#pragma clang fp contract(off)
                acc += h0 * dm[0];
                acc += h1 * dm[1];
                acc += h2 * dm[2];
                acc += h3 * dm[3];
            }
        }
        A.Ad[gid] = acc;
        v[0] = A.d[gid] * acc;
    }
    if (block_reduce_and_ticket<1>(v, A.partials, A.max_blocks,
            &A.status[I_TICKET1])) {
        A.scalars[S_DAD] = v[0];
        __threadfence();
    }
}

// x += alpha d; r -= alpha Ad; z = P r; partial sums of r.r, x.(b + r), z.r;
// then the termination tests of conjugate_gradient.h:136-198.
__global__ void __launch_bounds__(CG_THREADS)
cg_update_kernel(CgArgs A)
{
    if (A.status[I_DONE])
        return;
    int const gid = blockIdx.x * blockDim.x + threadIdx.x;
    int const n = gid >> 2, row = gid & 3;
    double const alpha = A.scalars[S_RR] / A.scalars[S_DAD];
    double v[3] = { 0.0, 0.0, 0.0 };
    bool const in_range = n < A.num_nodes;
    double xi = 0.0, ri = 0.0;
    if (in_range) {
#pragma clang fp contract(off)
        xi = A.x[gid] + alpha * A.d[gid];
        ri = A.r[gid] - alpha * A.Ad[gid];
    }
    // z = P r needs the node's whole residual: the four row-lanes of a node
    // are neighbours in the wave.
    int const base_lane = (threadIdx.x & 63) & ~3;
    double rn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        rn[k] = __shfl(ri, base_lane + k);
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
        v[0] = ri * ri;
        v[1] = xi * (A.b[gid] + ri);
        v[2] = zi * ri;
    }
    if (block_reduce_and_ticket<3>(v, A.partials, A.max_blocks,
            &A.status[I_TICKET2])) {
        double const new_rr = v[0];
        double const Q1 = -1.0 * v[1];
        int const it = A.status[I_ITER];
        bool done = false;
        if (new_rr < A.scalars[S_TOL]) {
            A.status[I_INFO] = SMVS_CG_CONVERGENCE;
            done = true;
        } else {
            double const Q0 = A.scalars[S_Q0];
            double const zeta = it * (Q1 - Q0) / Q1;
            if (zeta < A.q_tolerance) {
                A.status[I_INFO] = SMVS_CG_CONVERGENCE;
                done = true;
            }
        }
        A.scalars[S_RR_NEW] = new_rr;
        A.scalars[S_Q1] = Q1;
        if (!done) {
            A.scalars[S_Q0] = Q1;
            A.scalars[S_BETA] = v[2] / A.scalars[S_RR];
            A.scalars[S_RR] = v[2];
            A.status[I_ITER] = it + 1;
            if (it + 1 >= A.status[I_MAXITER]) {
                A.status[I_INFO] = SMVS_CG_MAX_ITERATIONS;
                done = true;
            }
        }
        if (done)
            A.status[I_DONE] = 1;
        __threadfence();
    }
}

// d = z + beta d
__global__ void __launch_bounds__(CG_THREADS)
cg_direction_kernel(CgArgs A)
{
    if (A.status[I_DONE])
        return;
    int const gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= A.num_nodes * 4)
        return;
    double const beta = A.scalars[S_BETA];
    {
#pragma clang fp contract(off)
        A.d[gid] = A.z[gid] + beta * A.d[gid];
    }
}

int
cg_solve_launch(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info)
{
    CgArgs A;
    A.H9 = ctx->H9;
    A.Pinv = ctx->Pinv;
    A.g = ctx->g;
    A.x = ctx->x;
    A.r = ctx->r;
    A.z = ctx->z;
    A.Ad = ctx->Ad;
    A.d = ctx->d;
    A.b = ctx->b;
    A.partials = ctx->partials;
    A.scalars = ctx->scalars;
    A.status = ctx->status;
    A.num_nodes = ctx->num_nodes;
    A.stride = ctx->node_stride;
    A.max_blocks = ctx->max_blocks;
    A.q_tolerance = q_tolerance;
    A.fixed_tolerance = error_tolerance;

    unsigned const blocks =
        (unsigned)(((size_t)ctx->num_nodes * 4 + CG_THREADS - 1) / CG_THREADS);
    if ((int)blocks > ctx->max_blocks) {
        set_error("cg_solve_launch: reduction buffer too small");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status + I_MAXITER, &max_iterations,
        sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    {
        ScopedKernelTimer timer(ctx, SMVS_K_CG_INIT);
        hipLaunchKernelGGL(cg_init_kernel, dim3(blocks), dim3(CG_THREADS), 0,
            ctx->stream, A);
    }
    SMVS_HIP_CHECK(hipGetLastError());

    int const chunk = 8;
    int issued = 1;  // iteration counter of the next iteration to enqueue
    bool done = max_iterations <= 1;
    while (!done) {
        for (int k = 0; k < chunk && issued < max_iterations; ++k, ++issued) {
            {
                ScopedKernelTimer timer(ctx, SMVS_K_CG_SPMV);
                hipLaunchKernelGGL(cg_spmv_kernel, dim3(blocks),
                    dim3(CG_THREADS), 0, ctx->stream, A);
            }
            {
                ScopedKernelTimer timer(ctx, SMVS_K_CG_UPDATE);
                hipLaunchKernelGGL(cg_update_kernel, dim3(blocks),
                    dim3(CG_THREADS), 0, ctx->stream, A);
            }
            {
                ScopedKernelTimer timer(ctx, SMVS_K_CG_DIR);
                hipLaunchKernelGGL(cg_direction_kernel, dim3(blocks),
                    dim3(CG_THREADS), 0, ctx->stream, A);
            }
        }
        SMVS_HIP_CHECK(hipGetLastError());
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host, ctx->status,
            sizeof(int) * I_NUM, hipMemcpyDeviceToHost, ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        done = ctx->status_host[I_DONE] != 0 || issued >= max_iterations;
    }
    if (max_iterations <= 1) {
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host, ctx->status,
            sizeof(int) * I_NUM, hipMemcpyDeviceToHost, ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
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
