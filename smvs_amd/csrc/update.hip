// Node update + re-activation, surface outputs, lighting normal equations and
// the fused Newton loop.
//
// Replaces DepthOptimizer::fill_node_reprojections (reference:
// lib/depth_optimizer.cc:647-677), Surface::update_nodes (lib/surface.cc:957-981),
// the active-set test (lib/depth_optimizer.cc:275-303), Surface::get_depth_map /
// get_normal_map (lib/surface.cc:155-183) and the accumulation loop of
// LightOptimizer::fit_lighting_to_image (lib/light_optimizer.cc:32-49).
#include "common.h"

#include <chrono>
#include <cmath>

namespace smvs_hip {

// bicubic value / first derivatives at pixel (ci, cj) of a patch from the
// 1-D Hermite table rows (global memory, [ps][12]).
__device__ __forceinline__ void
eval_patch(const double *tab, int ci, int cj, double const theta[16],
    double *w, double *wx, double *wy)
{
    const double *X = tab + (size_t)ci * 12;
    const double *Y = tab + (size_t)cj * 12;
    double v = 0.0, vx = 0.0, vy = 0.0;
#pragma unroll
    for (int ey = 0; ey < 4; ++ey) {
        double g0 = 0.0, g1 = 0.0;
#pragma unroll
        for (int ex = 0; ex < 4; ++ex) {
            int const a = ex & 1, ix = ex >> 1, b = ey & 1, iy = ey >> 1;
            double const c = theta[4 * (2 * b + a) + ix + 2 * iy];
            g0 = __builtin_fma(c, X[ex * 3 + 0], g0);
            g1 = __builtin_fma(c, X[ex * 3 + 1], g1);
        }
        v = __builtin_fma(g0, Y[ey * 3 + 0], v);
        vx = __builtin_fma(g1, Y[ey * 3 + 0], vx);
        vy = __builtin_fma(g0, Y[ey * 3 + 1], vy);
    }
    *w = v;
    *wx = vx;
    *wy = vy;
}

struct ReactivateArgs {
    const double *nodes;
    const double *x;            // delta
    const uint8_t *patch_valid;
    const uint32_t *patch_vis;
    const uint8_t *active;
    uint8_t *active_next;
    const double *hermite_tab;
    const DeviceCameras *cams;
    double *partials;
    double *scalars;
    int *status;
    int npx, stride, ps, start_x, start_y, n_subs, num_patches;
    int ps_log2;            // ps = 1 << ps_log2 (Surface: patchsize = 2^scale)
    double threshold;
    int full_optimization;
    const int *live_list;   // the step's live patches, or nullptr: all patches
    int live_count;
    int check_stop;
};

// One thread per (patch, full-resolution pixel): project with the old and
// the updated nodes into every visible neighbour (pixel coordinates WITHOUT
// the +0.5 convention, depth_optimizer.cc:669-672).
// (Round 6 also measured a variant that stages the 32 doubles of a workgroup's
// <= 64 patches in LDS instead of loading them per pixel thread: 23.5 / 27.9 us
// against 26.4 us on three boxes, nothing outside the box-to-box spread.  Not
// kept.)
// FULL: DepthOptimizer::Options::full_optimization (the mean shift instead of
// the active set, depth_optimizer.cc:277-288).  A template argument since round
// 6: as a run-time flag the compiler evaluated BOTH variants of the test for
// every neighbour and selected (38 FP64 instructions per neighbour where the
// active-set test needs 20).
template <bool FULL>
__global__ void __launch_bounds__(256)
reactivate_kernel(ReactivateArgs A)
{
    long long const gid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // (the wave-uniform words every thread needs first, asked for together: as
    // three tests in a row they were three scalar round trips before a wave's
    // first vector load)
    int const stop_words = A.status[I_STOP] | A.status[I_STEP_ABORT];
    int const live_on_device = A.status[I_LIVE_PATCHES];
    double const x0 = A.x[0];
    // the pipelined Newton loop: the loop has already ended / the solver of
    // this step gave up (finish_step_kernel reports it)
    if (A.check_stop && stop_words != 0)
        return;
    // the list length is read on the device when the launch was sized before
    // it was known (live_count < 0)
    int const live_count = A.live_list == nullptr ? A.num_patches
        : (A.live_count >= 0 ? A.live_count : live_on_device);
    if (((long long)live_count << (2 * A.ps_log2))
        > (long long)gridDim.x * blockDim.x) {
        // (cannot happen behind a patch kernel of the same step, which is
        // sized for the same length and would have abandoned the step)
        if (gid0 == 0)
            A.status[I_STEP_ABORT] = ABORT_GRID;
        return;
    }
    // NaN guard of the reference on delta[0] (depth_optimizer.cc:267)
    if (isnan(x0))
        return;
    // (the patch size is a power of two: shifts instead of 64-bit divisions,
    // which cost more than the arithmetic of a pixel)
    // (measured: one thread per 16-pixel chunk, loading the nodes once per
    // chunk, is slower -- 40 instead of 25 us: too few waves to hide the
    // latency of its serial pixels)
    int const slot = (int)(gid0 >> (2 * A.ps_log2));
    int const pid = (int)(gid0 & (long long)((1 << (2 * A.ps_log2)) - 1));
    // Round 6: ONE level of dependent loads behind the list entry (before: list
    // entry -> validity -> active flags -> nodes and deltas -> visibility mask):
    // everything that depends only on the patch id is asked for at once, whether
    // or not the patch turns out to need evaluation.
    // (What the kernel's 24 us are -- measured, profiles/r6_reactivate_counters.txt:
    // a wave lives 11,600 cycles at eight per SIMD, issuing 16 % of them, waiting
    // for memory 45 %, stalled at issue 39 % (109 FP64 instructions per wave at
    // four cycles each, eight waves taking turns).  Round 6 halved its vector
    // instructions (the template argument), flattened this chain and cut its
    // stores to one lane per patch, each for nothing measurable: the three
    // effects overlap, and what remains is ~27 waves per SIMD x 4.8 us / 8.)
    int patch = -1;
    if (slot < live_count)
        patch = A.live_list != nullptr ? A.live_list[slot] : slot;
    bool const in_range = patch >= 0 && patch < A.num_patches;
    int const pc = in_range ? patch : 0;
    int const ix = pc % A.npx, iy = pc / A.npx;
    int const n00 = iy * A.stride + ix;
    int const ids[4] = { n00, n00 + 1, n00 + A.stride, n00 + A.stride + 1 };
    uint8_t const valid = A.patch_valid[pc];
    uint8_t const act = A.active[ids[0]] | A.active[ids[1]] | A.active[ids[2]]
        | A.active[ids[3]];
    uint32_t const vis = A.patch_vis[pc];
    // w1 - w0 is the patch evaluated on the node deltas (the patch
    // is linear in its nodes)
    double th0[16], thd[16];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            th0[4 * n + k] = A.nodes[4 * (size_t)ids[n] + k];
            thd[4 * n + k] = A.x[4 * (size_t)ids[n] + k];
        }
    double sum = 0.0, cnt = 0.0;
    bool moved = false;
    if (in_range && valid && act != 0) {
        {
            double const th2 = A.threshold * A.threshold;
            int const ci = pid & (A.ps - 1), cj = pid >> A.ps_log2;
            double w0, dw, dum0, dum1;
            eval_patch(A.hermite_tab, ci, cj, th0, &w0, &dum0, &dum1);
            eval_patch(A.hermite_tab, ci, cj, thd, &dw, &dum0, &dum1);
            double const w1 = w0 + dw;
            double const u = (double)(A.start_x + ix * A.ps + ci);
            double const v = (double)(A.start_y + iy * A.ps + cj);
            for (int j = 0; j < A.n_subs; ++j) {
                if (!((vis >> j) & 1u))
                    continue;
                const double *M = A.cams->M[j];
                const double *t = A.cams->t[j];
                const double *sh = A.cams->shift[j];
                // (w0 p + t0)/(w0 r + t2) - (w1 p + t0)/(w1 r + t2)
                //   = (w0 - w1)(p t2 - r t0) / ((w0 r + t2)(w1 r + t2)):
                // no cancellation; the numerators are affine in the pixel
                // (depth_optimizer.cc:669-672, 277-303)
                double const r = M[6] * u + M[7] * v + M[8];
                double const nx = sh[0] * u + sh[1] * v + sh[2];
                double const ny = sh[3] * u + sh[4] * v + sh[5];
                double const den = (w0 * r + t[2]) * (w1 * r + t[2]);
                double const num2 = dw * dw * (nx * nx + ny * ny);
                cnt += 1.0;
                if (FULL) {
                    // the mean shift needs the quotient itself
                    double inv = __builtin_amdgcn_rcp(den);
                    inv = __builtin_fma(__builtin_fma(-den, inv, 1.0), inv, inv);
                    inv = __builtin_fma(__builtin_fma(-den, inv, 1.0), inv, inv);
                    double const d2 = num2 * inv * inv;
                    sum += sqrt(d2);
                    moved |= d2 > th2;
                } else {
                    // shift^2 > threshold^2 without the division
                    moved |= num2 > th2 * (den * den);
                }
            }
        }
    }
    if (!FULL) {
        // One lane per patch and wave raises the four flags (rounds 1-5: every
        // pixel that moved wrote them -- in a first step nearly all of 2 M
        // threads, four byte stores each to 128 k distinct bytes).  The lanes of
        // a patch are an aligned group of min(64, ps^2): they share the patch and
        // its node ids.
        int const lane = (int)(threadIdx.x & 63u);
        int const group = 2 * A.ps_log2 >= 6 ? 64 : 1 << (2 * A.ps_log2);
        unsigned long long const votes = __ballot(moved);
        unsigned long long const mine = group >= 64 ? ~0ull
            : ((1ull << group) - 1ull) << (lane & ~(group - 1));
        if ((votes & mine) != 0ull && (lane & (group - 1)) == 0) {
            A.active_next[ids[0]] = 1;
            A.active_next[ids[1]] = 1;
            A.active_next[ids[2]] = 1;
            A.active_next[ids[3]] = 1;
        }
    }
    if (!FULL)
        return;
    // mean reprojection delta (depth_optimizer.cc:277-282)
    __shared__ double red[2][4];
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        cnt += __shfl_xor(cnt, off);
    }
    if (lane == 0) {
        red[0][wave] = sum;
        red[1][wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0, c = 0.0;
        for (int wv = 0; wv < 4; ++wv) {
            s += red[0][wv];
            c += red[1][wv];
        }
        // the number of blocks can exceed the CG partial buffer: accumulate
        // with atomics (order-dependent in the last bits; only compared
        // against the 0.01 threshold).
        atomicAdd(&A.scalars[S_SUMDIFF], s);
        atomicAdd(&A.scalars[S_COUNT_DIFF], c);
    }
}

// delta[0] NaN guard (depth_optimizer.cc:267-268) and reset of the next set
__global__ void
prepare_update_kernel(uint8_t *active_next, int num_nodes, double *scalars,
    int *status)
{
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < num_nodes)
        active_next[i] = 0;
    if (i == 0) {
        status[I_NUM_ACTIVE] = 0;
        scalars[S_SUMDIFF] = 0.0;
        scalars[S_COUNT_DIFF] = 0.0;
    }
}

// nodes += delta for valid nodes (surface.cc:964-975); adopt the new active
// set and count it (depth_optimizer.cc:291-303).
__global__ void
apply_update_kernel(double *nodes, const double *x, const uint8_t *node_valid,
    uint8_t *active, const uint8_t *active_next, int num_nodes,
    int full_optimization, int *status)
{
    bool const skip = isnan(x[0]);
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0)
        status[I_NAN] = skip ? 1 : 0;
    bool on = false;
    if (!skip && i < num_nodes) {
        if (node_valid[i]) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                nodes[4 * (size_t)i + k] += x[4 * (size_t)i + k];
        }
        if (!full_optimization) {
            active[i] = active_next[i];
            on = active_next[i] == 1;
        } else {
            on = active[i] == 1;
        }
    }
    int const cnt = __syncthreads_count(on);
    if (threadIdx.x == 0 && cnt != 0)
        atomicAdd(&status[I_NUM_ACTIVE], cnt);
}

// End of a Newton step inside smvs_gn_run_loop, one launch instead of node
// update + list build + two device-to-host copies: thread c owns node c
// (nodes += delta, adopt the new active flag, count it, surface.cc:957-981,
// depth_optimizer.cc:291-303) and, where c is also the top-left node of a
// patch, that patch's entry in the next step's live list.  The last
// workgroup to finish publishes the step's result words in pinned host
// memory, which the host polls (no stream query, no copies).
struct FinishArgs {
    double *nodes;
    const double *x;
    const uint8_t *node_valid;
    const uint8_t *patch_valid;
    uint8_t *active;
    const uint8_t *active_next;
    int *list;
    int *status;
    unsigned long long *counter;   // bits 0-23 list length, 24-47 active nodes,
                                   // 48-63 workgroups arrived: ONE atomic per workgroup
                                   // (wide: [0] = list length | active nodes << 32,
                                   // [1] = workgroups arrived)
    int wide;              // surfaces of 2^24 nodes and more: two atomics per workgroup
    const double *scalars;
    int *host_words;       // pinned slot (STEP_SLOT_INTS ints): [0] sequence tag,
                           // [1] active nodes, [2] NaN, [3] active patches of the
                           // step, [4] next list length, [5] 0 ran / 1 skipped (the
                           // loop had ended) / 1 + ABORT_* (the step was abandoned),
                           // [6] CG iterations, [7] the loop ends after this step,
                           // [8..11] two doubles: sum of shifts, number of terms
    int npx, npy, stride, num_nodes;
    int full_optimization;
    int seq;
    int check_stop;        // launch-ahead mode: obey / maintain status[I_STOP]
    double full_opt_threshold;
    int begin;             // 1: the start of a loop instead of the end of a step --
                           // no node update; count the active set into
                           // status[I_NUM_INITIAL], build the first live list,
                           // clear the stop / abort words
    int reset_active;      // begin: active := valid nodes first (surface.cc:37-44)
};

constexpr int FINISH_THREADS = 1024;

__global__ void __launch_bounds__(FINISH_THREADS)
finish_step_kernel(FinishArgs A)
{
    if (A.check_stop && !A.begin) {
        // enqueued before the loop was known to have ended / the solver to
        // have given up: report it, touch nothing
        int const abort = A.status[I_STEP_ABORT];
        int const gate = A.status[I_STOP] != 0 ? 1 : (abort != 0 ? 1 + abort : 0);
        if (gate != 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                __hip_atomic_store(A.host_words + 5, gate, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(A.host_words + 0, A.seq, __ATOMIC_RELEASE,
                    __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
    }
    bool const begin = A.begin != 0;
    // depth_optimizer.cc:267: nothing is updated
    bool const skip = begin ? false : isnan(A.x[0]);
    bool const keep = begin ? A.reset_active == 0 : (skip || A.full_optimization != 0);
    // the flags the new active set comes from when it is not kept
    const uint8_t *incoming = begin ? A.node_valid : A.active_next;
    int const c = blockIdx.x * blockDim.x + threadIdx.x;
    bool on = false, live = false;
    int patch = 0;
    if (c < A.num_nodes) {
        if (!begin && !skip && A.node_valid[c]) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                A.nodes[4 * (size_t)c + k] += A.x[4 * (size_t)c + k];
        }
        // (other threads read `incoming`, never the flag adopted here)
        uint8_t const flag = keep ? A.active[c] : (incoming[c] != 0 ? 1 : 0);
        if (!keep)
            A.active[c] = flag;
        on = flag == 1;
        int const ix = c % A.stride, iy = c / A.stride;
        if (ix < A.npx && iy < A.npy) {
            patch = iy * A.npx + ix;
            const uint8_t *f = keep ? A.active : incoming;
            live = A.patch_valid[patch]
                && (flag | f[c + 1] | f[c + A.stride] | f[c + A.stride + 1]) != 0;
        }
    }
    // What the publishing thread reports besides the counts was written by
    // earlier launches of the stream: asked for here, all at once, instead of
    // one dependent round trip after the other behind the atomics (a launch of
    // ONE workgroup took 8.5 us, most of it that chain).
    int pre_active_patches = 0, pre_initial = 0, pre_iter = 0;
    double pre_sum_diff = 0.0, pre_count_diff = 0.0;
    if (threadIdx.x == 0) {
        pre_active_patches = A.status[I_ACTIVE_PATCHES];
        pre_initial = A.status[I_NUM_INITIAL];
        pre_iter = A.status[I_ITER];
        pre_sum_diff = A.scalars[S_SUMDIFF];
        pre_count_diff = A.scalars[S_COUNT_DIFF];
    }
    constexpr int WAVES = FINISH_THREADS / 64;
    __shared__ int wave_cnt[WAVES];
    __shared__ int base;
    __shared__ unsigned long long seen;
    __shared__ bool last_wide;
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long const ballot = __ballot(live);
    int const before = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0)
        wave_cnt[wave] = __popcll(ballot);
    int const cnt_on = __syncthreads_count(on);
    if (threadIdx.x == 0) {
        int total = 0;
        for (int wv = 0; wv < WAVES; ++wv)
            total += wave_cnt[wv];
        // one device-scope atomic per workgroup (atomics on one address are
        // served one after the other): list slots, active nodes and arrival
        if (!A.wide) {
            unsigned long long const add = (unsigned long long)total
                | ((unsigned long long)cnt_on << 24) | (1ull << 48);
            unsigned long long const old = atomicAdd(A.counter, add);
            base = (int)(old & 0xFFFFFFull);
            seen = old + add;
        } else {
            // the counts no longer fit beside the arrivals: the counts first,
            // then the arrival as a release / acquire (the counts of every
            // workgroup whose arrival the last one observes are visible to
            // its read-back); the last arriver reads the totals back
            unsigned long long const add = (unsigned long long)total
                | ((unsigned long long)cnt_on << 32);
            unsigned long long const old = atomicAdd(A.counter, add);
            base = (int)(old & 0xFFFFFFFFull);
            unsigned long long const arrived = __hip_atomic_fetch_add(A.counter + 1, 1ull,
                __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
            unsigned long long totals = 0ull;
            if (arrived == (unsigned long long)gridDim.x)
                totals = __hip_atomic_fetch_add(A.counter, 0ull, __ATOMIC_ACQUIRE,
                    __HIP_MEMORY_SCOPE_AGENT);
            // same layout as the packed word: length, active nodes, arrivals
            // (the packed fields are only decoded below)
            seen = arrived == (unsigned long long)gridDim.x ? totals : 0ull;
            last_wide = arrived == (unsigned long long)gridDim.x;
        }
    }
    __syncthreads();
    if (live) {
        int off = base + before;
        for (int wv = 0; wv < wave; ++wv)
            off += wave_cnt[wv];
        A.list[off] = patch;
    }
    // the workgroup that arrives last publishes the step (no fences: the
    // counters travelled in the atomic itself)
    bool const last = A.wide ? last_wide
        : (seen >> 48) == (unsigned long long)gridDim.x;
    if (threadIdx.x == 0 && last) {
        int const next_live = A.wide ? (int)(seen & 0xFFFFFFFFull)
            : (int)(seen & 0xFFFFFFull);
        int const num_active = A.wide ? (int)(seen >> 32)
            : (int)((seen >> 24) & 0xFFFFFFull);
        A.counter[0] = 0ull;   // (nobody else touches them any more) for the next launch
        A.counter[1] = 0ull;
        // (the publishing workgroup is whichever arrives last; thread 0 of
        // every workgroup has asked for the same words)
        int active_patches = pre_active_patches, initial = pre_initial;
        if (begin) {
            A.status[I_NUM_INITIAL] = num_active;
            A.status[I_STEP_ABORT] = 0;
            A.status[I_ACTIVE_PATCHES] = 0;
            active_patches = 0;
            initial = num_active;
        }
        A.status[I_NAN] = skip ? 1 : 0;
        A.status[I_NUM_ACTIVE] = num_active;
        A.status[I_LIVE_PATCHES] = next_live;
        __hip_atomic_store(A.host_words + 1, num_active, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.host_words + 2, skip ? 1 : 0, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.host_words + 3, active_patches,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.host_words + 4, next_live, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        double const sum_diff = pre_sum_diff;
        double const count_diff = pre_count_diff;
        // does the loop go on?  (depth_optimizer.cc:219-220, 267-268, 277-288;
        // the step limit is the host's business)
        bool stop = skip;
        if (A.full_optimization && !begin)
            stop = stop || sum_diff / count_diff < A.full_opt_threshold;
        else
            stop = stop || !(num_active > initial / 20);
        if (A.check_stop || begin)
            A.status[I_STOP] = stop ? 1 : 0;
        __hip_atomic_store(A.host_words + 5, 0, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.host_words + 6, begin ? 0 : pre_iter, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.host_words + 7, stop ? 1 : 0, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        double *host_scalars = reinterpret_cast<double *>(A.host_words + 8);
        host_scalars[0] = sum_diff;
        host_scalars[1] = count_diff;
        __hip_atomic_store(A.host_words + 0, A.seq, __ATOMIC_RELEASE,
            __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int loop_test_mode(void);   // SMVS_LOOP_TEST, below

static FinishArgs
finish_args(smvs_ctx *ctx, int seq)
{
    FinishArgs F;
    F.nodes = ctx->nodes;
    F.x = ctx->x;
    F.node_valid = ctx->node_valid;
    F.patch_valid = ctx->patch_valid;
    F.active = ctx->active;
    F.active_next = ctx->active_next;
    F.list = ctx->live_list;
    F.status = ctx->status;
    F.counter = ctx->step_counter;
    F.scalars = ctx->scalars;
    F.host_words = ctx->step_words + (seq & (STEP_SLOTS - 1)) * STEP_SLOT_INTS;
    F.npx = ctx->npx;
    F.npy = ctx->npy;
    F.stride = ctx->node_stride;
    F.num_nodes = ctx->num_nodes;
    F.wide = ctx->num_nodes >= (1 << 24) || loop_test_mode() == 4 ? 1 : 0;
    F.full_optimization = 0;
    F.seq = seq;
    F.check_stop = 0;
    F.full_opt_threshold = 0.0;
    F.begin = 0;
    F.reset_active = 0;
    return F;
}

// Start of a Newton loop (depth_optimizer.cc:214-220) without a host round
// trip: optionally active := valid nodes, count the active set into
// status[I_NUM_INITIAL], build the first live-patch list, clear the stop /
// abort words, publish the counts under `seq`.
static int
loop_begin_launch(smvs_ctx *ctx, bool reset_active, int seq)
{
    FinishArgs F = finish_args(ctx, seq);
    F.begin = 1;
    F.reset_active = reset_active ? 1 : 0;
    if (reset_active)
        ctx->cg_use_active = false;   // as smvs_ctx_set_active
    int const N = ctx->num_nodes;
    ScopedKernelTimer timer(ctx, SMVS_K_MISC);
    hipLaunchKernelGGL(finish_step_kernel,
        dim3((unsigned)((N + FINISH_THREADS - 1) / FINISH_THREADS)),
        dim3(FINISH_THREADS), 0, ctx->stream, F);
    SMVS_HIP_CHECK(hipGetLastError());
    return SMVS_OK;
}

int
reactivate_launch(smvs_ctx *ctx, double threshold, int full_optimization,
    bool build_live_list, int known_live, int publish_seq,
    const StepPipeline *pipe)
{
    if (pipe != nullptr) {
        publish_seq = pipe->seq;
        known_live = pipe->grid_live;
    }
    int const N = ctx->num_nodes;
    // (the assembly kernel of the same Newton step has already cleared the
    // flags and counters when the system came from smvs_gn_construct)
    if (!ctx->update_prepared) {
        ScopedKernelTimer timer(ctx, SMVS_K_MISC);
        hipLaunchKernelGGL(prepare_update_kernel, dim3((unsigned)((N + 255) / 256)),
            dim3(256), 0, ctx->stream, ctx->active_next, N, ctx->scalars,
            ctx->status);
    }
    ctx->update_prepared = false;
    SMVS_HIP_CHECK(hipGetLastError());
    ReactivateArgs A;
    A.nodes = ctx->nodes;
    A.x = ctx->x;
    A.patch_valid = ctx->patch_valid;
    A.patch_vis = ctx->patch_vis;
    A.active = ctx->active;
    A.active_next = ctx->active_next;
    A.hermite_tab = ctx->hermite_tab;
    A.cams = ctx->cams;
    A.partials = ctx->partials;
    A.scalars = ctx->scalars;
    A.status = ctx->status;
    A.npx = ctx->npx;
    A.stride = ctx->node_stride;
    A.ps = ctx->patchsize;
    A.ps_log2 = ctx->scale;
    A.start_x = ctx->start_x;
    A.start_y = ctx->start_y;
    A.n_subs = ctx->n_subs;
    A.num_patches = ctx->num_patches;
    A.threshold = threshold;
    A.full_optimization = full_optimization;
    // the live list of this step (built at the end of the previous one) when
    // the host knows its length
    A.live_list = known_live >= 0 ? ctx->live_list : nullptr;
    A.live_count = pipe != nullptr ? -1 : known_live;   // -1: read on the device
    A.check_stop = pipe != nullptr ? 1 : 0;
    long long const items = (long long)(known_live >= 0 ? known_live
        : ctx->num_patches) * ctx->patchsize * ctx->patchsize;
    {
        ScopedKernelTimer timer(ctx, SMVS_K_REACTIVATE);
        dim3 const grid((unsigned)((items + 255) / 256 > 0 ? (items + 255) / 256 : 1));
        if (full_optimization)
            hipLaunchKernelGGL(reactivate_kernel<true>, grid, dim3(256), 0, ctx->stream, A);
        else
            hipLaunchKernelGGL(reactivate_kernel<false>, grid, dim3(256), 0, ctx->stream, A);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    if (publish_seq != 0) {
        // the Newton loop: node update, next live list and the result words
        // in one launch
        FinishArgs F = finish_args(ctx, publish_seq);
        F.full_optimization = full_optimization;
        F.check_stop = pipe != nullptr ? 1 : 0;
        F.full_opt_threshold = pipe != nullptr ? pipe->full_opt_threshold : 0.0;
        ScopedKernelTimer timer(ctx, SMVS_K_MISC);
        hipLaunchKernelGGL(finish_step_kernel,
            dim3((unsigned)((N + FINISH_THREADS - 1) / FINISH_THREADS)),
            dim3(FINISH_THREADS), 0, ctx->stream, F);
        SMVS_HIP_CHECK(hipGetLastError());
        return SMVS_OK;
    }
    {
        ScopedKernelTimer timer(ctx, SMVS_K_MISC);
        hipLaunchKernelGGL(apply_update_kernel, dim3((unsigned)((N + 255) / 256)),
            dim3(256), 0, ctx->stream, ctx->nodes, ctx->x, ctx->node_valid,
            ctx->active, ctx->active_next, N, full_optimization, ctx->status);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    // the next construction's work list, from the new active set
    if (build_live_list)
        return live_patch_list_launch(ctx);
    return SMVS_OK;
}

// ---------------------------------------------------------------------------
// depth / normal maps
// ---------------------------------------------------------------------------
struct MapArgs {
    const double *nodes;
    const uint8_t *patch_valid;
    const double *hermite_tab;
    float *depth;     // [H][W] or nullptr
    float *normals;   // [H][W][3] or nullptr
    int W, H, npx, stride, ps, start_x, start_y, num_patches;
    double inv_flen;
    int to_mve;          // depth in MVE's ray-length convention
    float invproj[9];    // CameraInfo::fill_inverse_calibration (to_mve)
};

__global__ void __launch_bounds__(256)
surface_maps_kernel(MapArgs A)
{
    int const pp = A.ps * A.ps;
    long long const gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int const patch = (int)(gid / pp);
    int const pid = (int)(gid - (long long)patch * pp);
    if (patch >= A.num_patches || !A.patch_valid[patch])
        return;
    int const ix = patch % A.npx, iy = patch / A.npx;
    int const n00 = iy * A.stride + ix;
    int const ids[4] = { n00, n00 + 1, n00 + A.stride, n00 + A.stride + 1 };
    double th[16];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            th[4 * n + k] = A.nodes[4 * (size_t)ids[n] + k];
    int const ci = pid % A.ps, cj = pid / A.ps;
    double w, wx, wy;
    eval_patch(A.hermite_tab, ci, cj, th, &w, &wx, &wy);
    int const px = A.start_x + ix * A.ps + ci;
    int const py = A.start_y + iy * A.ps + cj;
    size_t const o = (size_t)py * A.W + px;
    if (A.depth != nullptr) {
        float d = (float)w;
        if (A.to_mve) {
#pragma clang fp contract(off)
            // StereoView::write_depth_to_view (stereo_view.h:100-119):
            // mve::image::depthmap_convert_conventions, z-depth -> ray length
            // (`double len = px.norm(); dm *= len`, tests/golden/README.md M10;
            // the float operations of host/stereo_view.cc, the inverse of
            // mesh.hip's mesh_prepare_kernel)
            float const fx = (float)px + 0.5f, fy = (float)py + 0.5f;
            float v[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                v[r] = A.invproj[3 * r] * fx + A.invproj[3 * r + 1] * fy
                    + A.invproj[3 * r + 2];
            float const len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            d = (float)((double)d * (double)len);
        }
        A.depth[o] = d;
    }
    if (A.normals != nullptr) {
        // surface_derivative.cc:17-28
        double const x = (double)px + 0.5 - (double)A.W / 2.0;
        double const y = (double)py + 0.5 - (double)A.H / 2.0;
        double const nz = (x * wx + y * wy + w) * A.inv_flen;
        double const len = sqrt(wx * wx + wy * wy + nz * nz);
        A.normals[3 * o + 0] = (float)(wx / len);
        A.normals[3 * o + 1] = (float)(-wy / len);
        A.normals[3 * o + 2] = (float)(nz / len);
    }
}

static int
launch_maps(smvs_ctx *ctx, float *depth_dev, float *normals_dev,
    const float *inv_calibration9 = nullptr)
{
    MapArgs A;
    A.to_mve = inv_calibration9 != nullptr ? 1 : 0;
    for (int i = 0; i < 9; ++i)
        A.invproj[i] = inv_calibration9 != nullptr ? inv_calibration9[i] : 0.0f;
    A.nodes = ctx->nodes;
    A.patch_valid = ctx->patch_valid;
    A.hermite_tab = ctx->hermite_tab;
    A.depth = depth_dev;
    A.normals = normals_dev;
    A.W = ctx->width;
    A.H = ctx->height;
    A.npx = ctx->npx;
    A.stride = ctx->node_stride;
    A.ps = ctx->patchsize;
    A.start_x = ctx->start_x;
    A.start_y = ctx->start_y;
    A.num_patches = ctx->num_patches;
    A.inv_flen = (double)ctx->inv_flen;
    long long const items = (long long)ctx->num_patches * ctx->patchsize
        * ctx->patchsize;
    ScopedKernelTimer timer(ctx, SMVS_K_MISC);
    hipLaunchKernelGGL(surface_maps_kernel, dim3((unsigned)((items + 255) / 256)),
        dim3(256), 0, ctx->stream, A);
    SMVS_HIP_CHECK(hipGetLastError());
    return SMVS_OK;
}

// ---------------------------------------------------------------------------
// lighting normal equations: A += sh sh^T, b += sh I over valid pixels
// ---------------------------------------------------------------------------
__device__ __forceinline__ void
sh_basis_4_band(const double n[3], double sh[16])
{
    double const nx = n[0], ny = n[1], nz = n[2];
    double const x2 = nx * nx, y2 = ny * ny, z2 = nz * nz;
    sh[0] = 1.0; sh[1] = ny; sh[2] = nz; sh[3] = nx;
    sh[4] = nx * ny; sh[5] = ny * nz; sh[6] = -x2 - y2 + 2.0 * z2;
    sh[7] = nx * nz; sh[8] = x2 - y2;
    sh[9] = (3.0 * x2 - y2) * ny; sh[10] = nx * ny * nz;
    sh[11] = (4.0 * z2 - x2 - y2) * ny;
    sh[12] = (2.0 * z2 - 3.0 * x2 - 3.0 * y2) * nz;
    sh[13] = (4.0 * z2 - x2 - y2) * nx;
    sh[14] = (x2 - y2) * nz; sh[15] = (x2 - 3.0 * y2) * nx;
}

constexpr int LIGHT_BLOCKS = 512;

// partial[block][152]: upper triangle of A (136) then b (16)
__global__ void __launch_bounds__(256)
light_accumulate_kernel(const float *__restrict__ normals,
    const float *__restrict__ image, size_t num_pixels,
    double *__restrict__ partial)
{
    double acc[152];
#pragma unroll
    for (int i = 0; i < 152; ++i)
        acc[i] = 0.0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
         p < num_pixels; p += (size_t)gridDim.x * blockDim.x) {
        double const n[3] = { (double)normals[3 * p], (double)normals[3 * p + 1],
            (double)normals[3 * p + 2] };
        double const len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        float const iv = image[p];
        if (fabs(len - 1.0) > 1e-6 || iv < 0.05f)
            continue;
        double sh[16];
        sh_basis_4_band(n, sh);
        int k = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = i; j < 16; ++j)
                acc[k++] += sh[i] * sh[j];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[136 + i] += sh[i] * (double)iv;
    }
    __shared__ double red[4][152];
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 152; ++i) {
        double s = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s += __shfl_xor(s, off);
        if (lane == 0)
            red[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 152; i += blockDim.x)
        partial[(size_t)blockIdx.x * 152 + i] = red[0][i] + red[1][i]
            + red[2][i] + red[3][i];
}

__global__ void
light_finalize_kernel(const double *__restrict__ partial, int blocks,
    double *__restrict__ Ab)
{
    int const i = threadIdx.x;
    if (i >= 152)
        return;
    double s = 0.0;
    for (int b = 0; b < blocks; ++b)
        s += partial[(size_t)b * 152 + i];
    if (i >= 136) {
        Ab[256 + (i - 136)] = s;
        return;
    }
    // unpack upper-triangle index
    int r = 0, k = i;
    while (k >= 16 - r) {
        k -= 16 - r;
        r += 1;
    }
    int const c = r + k;
    Ab[r * 16 + c] = s;
    Ab[c * 16 + r] = s;
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_update_and_reactivate(smvs_ctx *ctx, double threshold,
    int full_optimization, int *num_active, double *mean_delta, int *nan_flag)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_surface || !ctx->has_cameras) {
        set_error("smvs_update_and_reactivate: no surface / cameras");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    int rc = reactivate_launch(ctx, threshold, full_optimization);
    if (rc != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host, ctx->status,
        sizeof(int) * I_NUM, hipMemcpyDeviceToHost, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->scalars_host, ctx->scalars,
        sizeof(double) * S_NUM, hipMemcpyDeviceToHost, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (num_active != nullptr)
        *num_active = ctx->status_host[I_NUM_ACTIVE];
    if (nan_flag != nullptr)
        *nan_flag = ctx->status_host[I_NAN];
    // sum / count like depth_optimizer.cc:277-282: NaN when no patch is seen
    // by any neighbour (0 / 0), which the caller's `update < 0.01` rejects
    if (mean_delta != nullptr)
        *mean_delta = ctx->scalars_host[S_SUMDIFF]
            / ctx->scalars_host[S_COUNT_DIFF];
    return SMVS_OK;
}

// State of one smvs_gn_run_loop call
struct LoopState {
    int num_initial = 0;
    int num_active = 0;
    int newton_step = 0;
    int known_live = -1;   // length of the live-patch list once read back
    bool ended = false;    // a break of depth_optimizer.cc:267-268 / :284-288
    int begin_seq = 0;     // tag of the loop-begin words while they are unread
};

static bool
loop_goes_on(const smvs_gn_loop_params *prm, LoopState const &L)
{
    // depth_optimizer.cc:219-220
    return !L.ended && L.newton_step < prm->max_newton_steps
        && L.num_active > L.num_initial / 20;
}

// Spins on a result slot until finish_step_kernel has published `seq` there:
// the next launches follow the end of the step within microseconds (no stream
// query, no copies).
static int
wait_step_words(smvs_ctx *ctx, int seq, const int **slot_out)
{
    const int *words = ctx->step_words + (seq & (STEP_SLOTS - 1)) * STEP_SLOT_INTS;
    auto const t_start = std::chrono::steady_clock::now();
    long spins = 0;
    while (__atomic_load_n(&words[0], __ATOMIC_ACQUIRE) != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFFF) == 0) {
            hipError_t const q = hipStreamQuery(ctx->stream);
            if (q != hipSuccess && q != hipErrorNotReady)
                SMVS_HIP_CHECK(q);
            if (q == hipSuccess
                && __atomic_load_n(&words[0], __ATOMIC_ACQUIRE) != seq) {
                set_error("smvs_gn_run_loop: the step ended without "
                    "publishing its result");
                return SMVS_ERR_STATE;
            }
            if (std::chrono::steady_clock::now() - t_start
                > std::chrono::seconds(60)) {
                set_error("smvs_gn_run_loop: timed out waiting for the device");
                return SMVS_ERR_STATE;
            }
        }
    }
    *slot_out = words;
    return SMVS_OK;
}

// The counts loop_begin_launch published: the initial active set and the
// length of the first live list.
static int
read_loop_begin(smvs_ctx *ctx, LoopState &L)
{
    const int *words = nullptr;
    int const rc = wait_step_words(ctx, L.begin_seq, &words);
    if (rc != SMVS_OK)
        return rc;
    L.num_initial = L.num_active = words[1];
    L.known_live = words[4];
    L.begin_seq = 0;
    return SMVS_OK;
}

// The bookkeeping of depth_optimizer.cc:267-303 on a step's result words.
static void
account_step(const smvs_gn_loop_params *prm, smvs_gn_loop_stats *stats,
    LoopState &L, const int *words, int cg_iterations)
{
    L.newton_step += 1;
    stats->linear_iterations += cg_iterations;
    stats->active_patch_steps += words[3];
    L.known_live = words[4];
    if (words[2] != 0) {
        stats->nan_break = 1;
        L.ended = true;
        return;
    }
    if (prm->full_optimization) {
        // depth_optimizer.cc:277-288: sum_diff / size; with no reprojection
        // term at all this is 0 / 0 = NaN, the comparison is false and the
        // loop goes on to its step limit like the reference
        const double *sc = reinterpret_cast<const double *>(words + 8);
        if (sc[0] / sc[1] < prm->full_opt_threshold)
            L.ended = true;
        return;
    }
    L.num_active = words[1];
}

static int
next_step_seq(smvs_ctx *ctx)
{
    ctx->step_seq = (ctx->step_seq % 0x3FFFFFFF) + 1;
    return ctx->step_seq;
}

// SMVS_LOOP_TEST (tests/test_gpu_parity.py, tools/cg_trace.py): "undersize"
// sizes every launch-ahead step for half the list, "solver" makes the second
// solve of a loop report that it gave up, "unpipelined" runs the fused steps
// one at a time (the host waits for each solve: what SMVS_CG_TRACE needs),
// "wide" uses the end-of-step kernel's counters for surfaces of 2^24 nodes and
// more on any surface.
int
smvs_hip::loop_test_mode(void)
{
    static int const mode = [] {
        const char *e = std::getenv("SMVS_LOOP_TEST");
        if (e == nullptr)
            return 0;
        return std::strcmp(e, "undersize") == 0 ? 1
            : std::strcmp(e, "solver") == 0 ? 2
            : std::strcmp(e, "unpipelined") == 0 ? 3
            : std::strcmp(e, "wide") == 0 ? 4 : 0;
    }();
    return mode;
}

// One Newton step, the host waits for its result before it launches the next
// one: with the assembly kernel and the streaming solver, or (test mode
// "unpipelined") with the fused resident solver.
static int
run_step_streaming(smvs_ctx *ctx, const smvs_gn_loop_params *prm,
    smvs_gn_loop_stats *stats, LoopState &L)
{
    int rc;
    bool const fused = loop_test_mode() == 3
        && cg_resident_applies(ctx, prm->cg_max_iterations);
    if ((rc = gn_construct_launch(ctx, prm->regularization,
            prm->light_surf_regularization, prm->use_lighting != 0,
            L.known_live, fused)) != SMVS_OK)
        return rc;
    int iters = 0, info = 0;
    bool solved = false;
    if (fused && (rc = cg_resident_solve(ctx, prm->cg_max_iterations, -1.0,
            prm->cg_q_tolerance, &iters, &info, &solved, true)) != SMVS_OK)
        return rc;
    if (!solved) {
        if (fused && (rc = gn_assemble_launch(ctx)) != SMVS_OK)
            return rc;
        if ((rc = cg_solve_launch(ctx, prm->cg_max_iterations, -1.0,
                prm->cg_q_tolerance, &iters, &info)) != SMVS_OK)
            return rc;
    }
    int const seq = next_step_seq(ctx);
    if ((rc = reactivate_launch(ctx, prm->active_threshold,
            prm->full_optimization, true, L.known_live, seq)) != SMVS_OK)
        return rc;
    const int *words = nullptr;
    if ((rc = wait_step_words(ctx, seq, &words)) != SMVS_OK)
        return rc;
    account_step(prm, stats, L, words, iters);
    return SMVS_OK;
}

// The Newton loop with the fused resident solver, launch-ahead: step k + 1 is
// enqueued before the result of step k is known, so the GPU never waits for
// the host between steps.  The kernels size themselves from the device-side
// list length; finish_step_kernel evaluates the loop condition into
// status[I_STOP], and a step enqueued behind the end of the loop does nothing
// but report that it was skipped.  Returns with L.ended set, or -- when the
// solver gave up -- with the context's resident solver disabled and L at the
// last completed step (the caller goes on with run_step_streaming).
static int
run_steps_pipelined(smvs_ctx *ctx, const smvs_gn_loop_params *prm,
    smvs_gn_loop_stats *stats, LoopState &L)
{
    // the loop's share of the device's CUs (common.h, DeviceTileBudget): the
    // barrier kernels running side by side never ask for more workgroups than
    // the device can keep resident together
    ScopedTileBudget guard(ctx->device, cg_resident_tiles(ctx));
    static_assert(I_STEP_ABORT == I_STOP + 1, "cleared together");
    int const test_mode = loop_test_mode();
    int rc;
    int seqs[2] = { 0, 0 };      // tags of the steps in flight, oldest first
    int in_flight = 0;
    int enqueued = L.newton_step;   // steps enqueued so far (absolute number)
    auto enqueue = [&]() -> int {
        StepPipeline P;
        P.seq = next_step_seq(ctx);
        // The launches are sized for the newest list length the host has
        // seen -- one step old when the launch is ahead -- plus some slack
        // (the surplus workgroups leave at once; a list that still came out
        // longer makes the step abandon itself, below), for every patch
        // before the first count has come back.  Multiples of 32 patches:
        // the patch kernel's 8 bands of 4-patch workgroups.
        long long want = ctx->num_patches;
        if (L.known_live >= 0 && L.known_live + L.known_live / 16 + 64 < want)
            want = L.known_live + L.known_live / 16 + 64;
        // (test hooks, see loop_test_mode)
        if (test_mode == 1 && in_flight >= 1 && L.known_live >= 0)
            want = L.known_live / 2;
        P.grid_live = (int)((want + 31) / 32 * 32);
        P.full_opt_threshold = prm->full_opt_threshold;
        int r = gn_construct_launch(ctx, prm->regularization,
            prm->light_surf_regularization, prm->use_lighting != 0,
            P.grid_live, true, true);
        if (r == SMVS_OK)
            r = cg_resident_enqueue(ctx, prm->cg_max_iterations,
                prm->cg_q_tolerance, test_mode == 2 && enqueued == 1);
        if (r == SMVS_OK)
            r = reactivate_launch(ctx, prm->active_threshold,
                prm->full_optimization, true, -1, 0, &P);
        if (r == SMVS_OK) {
            seqs[in_flight++] = P.seq;
            enqueued += 1;
        }
        return r;
    };
    // every enqueued step is waited for before this function returns (the
    // result slots and the stop words are then quiet)
    auto drain = [&]() -> int {
        int r = SMVS_OK;
        while (in_flight > 0) {
            const int *words = nullptr;
            int const q = wait_step_words(ctx, seqs[0], &words);
            if (q != SMVS_OK && r == SMVS_OK)
                r = q;
            seqs[0] = seqs[1];
            in_flight -= 1;
            if (q != SMVS_OK)
                break;
        }
        return r;
    };
    // An error exit must not release the device's barrier-kernel mutex while
    // launches of this loop may still be in flight: wait for the stream (the
    // result words may never come, so no drain()).
    auto fail = [&](int r) -> int {
        (void)hipStreamSynchronize(ctx->stream);
        return r;
    };
    if ((rc = enqueue()) != SMVS_OK)
        return fail(rc);
    // A step enqueued behind the end of the loop costs four empty launches
    // (~20 us); a step that was not enqueued ahead costs the host's turn-around
    // (~25 us) once its predecessor has ended.  So the next step is enqueued
    // ahead only while the loop is likely to go on: the active set (one step
    // old) is still well above the level at which the loop stops, the mean
    // shift of full optimisation well above its threshold, and -- before the
    // first result -- the previous loop on this context ran that far.
    double last_update = -1.0;   // full optimisation: mean shift of the newest step
    auto likely_to_go_on = [&]() -> bool {
        if (test_mode == 1)
            return true;
        if (L.newton_step == 0)
            return ctx->last_loop_steps > enqueued;
        if (prm->full_optimization)
            return !(last_update < 4.0 * prm->full_opt_threshold);
        return L.num_active / 4 > L.num_initial / 20;
    };
    for (;;) {
        if (enqueued < prm->max_newton_steps
            && (in_flight == 0 || (in_flight < 2 && likely_to_go_on())))
            if ((rc = enqueue()) != SMVS_OK)
                return fail(rc);
        if (L.begin_seq != 0) {
            // (published long before the first step ends)
            if ((rc = read_loop_begin(ctx, L)) != SMVS_OK)
                return fail(rc);
            if (!loop_goes_on(prm, L)) {
                // no active node: the steps in flight report themselves skipped
                L.ended = true;
                return drain();
            }
        }
        const int *words = nullptr;
        if ((rc = wait_step_words(ctx, seqs[0], &words)) != SMVS_OK)
            return fail(rc);
        seqs[0] = seqs[1];
        in_flight -= 1;
        if (words[5] == 1 + ABORT_SOLVER || words[5] == 1 + ABORT_GRID) {
            // this step and the one behind it did nothing
            bool const solver = words[5] == 1 + ABORT_SOLVER;
            if ((rc = drain()) != SMVS_OK)
                return fail(rc);
            SMVS_HIP_CHECK(hipMemsetAsync(ctx->status + I_STOP, 0,
                2 * sizeof(int), ctx->stream));
            enqueued = L.newton_step;
            if (solver) {
                // its workgroups were not all resident: never try again on
                // this context, the caller goes on with the streaming solver
                return cg_resident_gave_up(ctx);
            }
            // the live list outgrew the launch: again, sized for the list
            // length that has come back in the meantime
            if ((rc = enqueue()) != SMVS_OK)
                return fail(rc);
            continue;
        }
        if (words[5] != 0) {
            set_error("smvs_gn_run_loop: a step was skipped before the loop ended");
            return fail(SMVS_ERR_STATE);
        }
        ctx->last_cg_iterations = words[6];
        account_step(prm, stats, L, words, words[6]);
        if (prm->full_optimization) {
            const double *sc = reinterpret_cast<const double *>(words + 8);
            last_update = sc[0] / sc[1];
        }
        if (!L.ended && !(L.num_active > L.num_initial / 20))
            L.ended = true;
        if (L.ended != (words[7] != 0)) {
            set_error("smvs_gn_run_loop: host and device disagree on the end "
                "of the loop");
            return fail(SMVS_ERR_STATE);
        }
        if (L.ended || L.newton_step >= prm->max_newton_steps)
            break;
    }
    // (a step enqueued behind the end reports itself skipped)
    rc = drain();
    L.ended = true;
    ctx->last_loop_steps = L.newton_step;
    return rc;
}

extern "C" int
smvs_gn_run_loop(smvs_ctx *ctx, const smvs_gn_loop_params *prm,
    smvs_gn_loop_stats *stats)
{
    SMVS_REQUIRE(ctx && prm && stats, "null argument");
    if (!ctx->has_surface || !ctx->has_cameras) {
        set_error("smvs_gn_run_loop: no surface / cameras");
        return SMVS_ERR_STATE;
    }
    if (prm->use_lighting && !ctx->has_shading) {
        set_error("smvs_gn_run_loop: lighting requested without shading planes");
        return SMVS_ERR_STATE;
    }
    SMVS_REQUIRE(prm->max_newton_steps >= 0, "max_newton_steps is negative");
    SMVS_REQUIRE(prm->active_threshold >= 0.0 && prm->full_opt_threshold >= 0.0
        && prm->cg_q_tolerance >= 0.0, "negative threshold");
    // (the iteration number travels in the low 16 bits of the solvers' tags,
    // as in cg_solve_launch; the neighbour planes are checked by
    // gn_construct_launch, shared with the stand-alone entry point)
    SMVS_REQUIRE(prm->cg_max_iterations >= 0 && prm->cg_max_iterations <= 0xFFFF,
        "cg_max_iterations out of range [0, 65535]");
    // (finish_step_kernel packs the list length and the active-node count
    // into 24 bits each of one atomic; larger surfaces use two atomics per
    // workgroup, FinishArgs::wide)
    SMVS_HIP_CHECK(set_device(ctx->device));
    memset(stats, 0, sizeof(*stats));
    int rc;
    if (prm->use_lighting)
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->lighting, prm->lighting,
            16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));

    LoopState L;
    if (prm->max_newton_steps == 0) {
        // nothing to run: only the count of the active set is reported
        int num_initial = 0;
        if (prm->reset_active
            && (rc = smvs_ctx_set_active(ctx, nullptr)) != SMVS_OK)
            return rc;
        if ((rc = smvs_get_active(ctx, nullptr, &num_initial)) != SMVS_OK)
            return rc;
        L.num_initial = L.num_active = num_initial;
    } else {
        // the start of the loop happens on the device (no round trip before
        // the first step): active set, its size, the first live list
        L.begin_seq = next_step_seq(ctx);
        if ((rc = loop_begin_launch(ctx, prm->reset_active != 0, L.begin_seq))
                != SMVS_OK)
            return rc;
    }
    while (L.begin_seq != 0 || loop_goes_on(prm, L)) {
        // With the resident solver the assembly happens inside the solve (H,
        // g and P never travel through HBM) and the steps are enqueued ahead
        // of their predecessors' results; if it cannot run -- or gives up
        // because its workgroups were not all resident -- the assembly kernel
        // and the streaming solver take over, one step at a time.
        if (loop_test_mode() != 3
            && cg_resident_applies(ctx, prm->cg_max_iterations)) {
            if ((rc = run_steps_pipelined(ctx, prm, stats, L)) != SMVS_OK)
                return rc;
            continue;
        }
        if (L.begin_seq != 0) {
            if ((rc = read_loop_begin(ctx, L)) != SMVS_OK)
                return rc;
            continue;
        }
        if ((rc = run_step_streaming(ctx, prm, stats, L)) != SMVS_OK)
            return rc;
    }
    int const newton_step = L.newton_step;
    int const num_active = L.num_active;
    stats->newton_steps = newton_step;
    stats->final_active_nodes = num_active;
    return SMVS_OK;
}

// W*H*3 floats of output scratch owned by the context (no allocation per call)
static int
ensure_map_scratch(smvs_ctx *ctx)
{
    size_t const want = (size_t)ctx->width * ctx->height * 3;
    if (ctx->map_scratch != nullptr && ctx->map_scratch_floats >= want)
        return SMVS_OK;
    int const rc = device_alloc(&ctx->map_scratch, want);
    ctx->map_scratch_floats = rc == SMVS_OK ? want : 0;
    return rc;
}

extern "C" int
smvs_get_depth_map(smvs_ctx *ctx, float *depth)
{
    SMVS_REQUIRE(ctx && depth, "null argument");
    if (!ctx->has_surface) {
        set_error("smvs_get_depth_map: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const npix = (size_t)ctx->width * ctx->height;
    int rc = ensure_map_scratch(ctx);
    if (rc != SMVS_OK)
        return rc;
    float *buf = ctx->map_scratch;
    hipError_t e = hipMemsetAsync(buf, 0, npix * sizeof(float), ctx->stream);
    if (e == hipSuccess)
        rc = launch_maps(ctx, buf, nullptr);
    if (e != hipSuccess) {
        set_error("smvs_get_depth_map: %s", hipGetErrorString(e));
        return SMVS_ERR_HIP;
    }
    if (rc != SMVS_OK)
        return rc;
    return ctx_download(ctx, depth, buf, npix * sizeof(float));
}

extern "C" int
smvs_get_normal_map(smvs_ctx *ctx, float *normals)
{
    SMVS_REQUIRE(ctx && normals, "null argument");
    if (!ctx->has_surface) {
        set_error("smvs_get_normal_map: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const n = (size_t)ctx->width * ctx->height * 3;
    int rc = ensure_map_scratch(ctx);
    if (rc != SMVS_OK)
        return rc;
    float *buf = ctx->map_scratch;
    hipError_t e = hipMemsetAsync(buf, 0, n * sizeof(float), ctx->stream);
    if (e == hipSuccess)
        rc = launch_maps(ctx, nullptr, buf);
    if (e != hipSuccess) {
        set_error("smvs_get_normal_map: %s", hipGetErrorString(e));
        return SMVS_ERR_HIP;
    }
    if (rc != SMVS_OK)
        return rc;
    return ctx_download(ctx, normals, buf, n * sizeof(float));
}

extern "C" int
smvs_get_maps(smvs_ctx *ctx, const float *inv_calibration9, float *depth, float *normals)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    SMVS_REQUIRE(depth != nullptr && normals != nullptr, "null output");
    if (!ctx->has_surface) {
        set_error("smvs_get_maps: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const npix = (size_t)ctx->width * ctx->height;
    // depth behind the normals in one scratch buffer of W*H*4 floats
    if (ctx->map_scratch_floats < npix * 4) {
        int const rc = device_alloc(&ctx->map_scratch, npix * 4);
        if (rc != SMVS_OK) {
            ctx->map_scratch_floats = 0;
            return rc;
        }
        ctx->map_scratch_floats = npix * 4;
    }
    float *nbuf = ctx->map_scratch, *dbuf = ctx->map_scratch + npix * 3;
    SMVS_HIP_CHECK(hipMemsetAsync(nbuf, 0, npix * 4 * sizeof(float), ctx->stream));
    int rc = launch_maps(ctx, dbuf, nbuf, inv_calibration9);
    if (rc != SMVS_OK)
        return rc;
    if (host_pointer_is_pinned(depth) && host_pointer_is_pinned(normals)) {
        SMVS_HIP_CHECK(hipMemcpyAsync(depth, dbuf, npix * sizeof(float),
            hipMemcpyDeviceToHost, ctx->stream));
        SMVS_HIP_CHECK(hipMemcpyAsync(normals, nbuf, npix * 3 * sizeof(float),
            hipMemcpyDeviceToHost, ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return SMVS_OK;
    }
    if ((rc = ctx_download(ctx, depth, dbuf, npix * sizeof(float))) != SMVS_OK)
        return rc;
    return ctx_download(ctx, normals, nbuf, npix * 3 * sizeof(float));
}

extern "C" int
smvs_light_accumulate_dev(smvs_ctx *ctx, double **Ab272_dev)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_surface || !ctx->has_shading) {
        set_error("smvs_light_accumulate: needs a surface and a shading image");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const npix = (size_t)ctx->width * ctx->height;
    int rc = ensure_map_scratch(ctx);
    if (rc != SMVS_OK)
        return rc;
    if (ctx->light_partial == nullptr
        && (rc = device_alloc(&ctx->light_partial,
                (size_t)LIGHT_BLOCKS * 152)) != SMVS_OK)
        return rc;
    float *normals = ctx->map_scratch;
    double *partial = ctx->light_partial;
    hipError_t e = hipMemsetAsync(normals, 0, npix * 3 * sizeof(float),
        ctx->stream);
    if (e == hipSuccess)
        rc = launch_maps(ctx, nullptr, normals);
    if (e == hipSuccess && rc == SMVS_OK) {
        ScopedKernelTimer timer(ctx, SMVS_K_MISC);
        hipLaunchKernelGGL(light_accumulate_kernel, dim3(LIGHT_BLOCKS),
            dim3(256), 0, ctx->stream, normals, ctx->main_shading, npix,
            partial);
        hipLaunchKernelGGL(light_finalize_kernel, dim3(1), dim3(256), 0,
            ctx->stream, partial, LIGHT_BLOCKS, ctx->lightAb);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("smvs_light_accumulate: %s", hipGetErrorString(e));
        return SMVS_ERR_HIP;
    }
    if (Ab272_dev != nullptr)
        *Ab272_dev = ctx->lightAb;
    return rc;
}

extern "C" int
smvs_light_accumulate(smvs_ctx *ctx, double *A256, double *b16)
{
    SMVS_REQUIRE(ctx && A256 && b16, "null argument");
    int rc = smvs_light_accumulate_dev(ctx, nullptr);
    if (rc != SMVS_OK)
        return rc;
    double host[272];
    SMVS_HIP_CHECK(hipMemcpy(host, ctx->lightAb, sizeof(host),
        hipMemcpyDeviceToHost));
    memcpy(A256, host, 256 * sizeof(double));
    memcpy(b16, host + 256, 16 * sizeof(double));
    return SMVS_OK;
}

extern "C" int
smvs_light_upload(smvs_ctx *ctx, const double *A256, const double *b16)
{
    SMVS_REQUIRE(ctx && A256 && b16, "null argument");
    SMVS_HIP_CHECK(set_device(ctx->device));
    double host[272];
    memcpy(host, A256, 256 * sizeof(double));
    memcpy(host + 256, b16, 16 * sizeof(double));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    SMVS_HIP_CHECK(hipMemcpy(ctx->lightAb, host, sizeof(host), hipMemcpyHostToDevice));
    return SMVS_OK;
}

extern "C" int
smvs_light_download(smvs_ctx *ctx, double *A256, double *b16)
{
    SMVS_REQUIRE(ctx && A256 && b16, "null argument");
    SMVS_HIP_CHECK(set_device(ctx->device));
    double host[272];
    SMVS_HIP_CHECK(hipMemcpy(host, ctx->lightAb, sizeof(host),
        hipMemcpyDeviceToHost));
    memcpy(A256, host, 256 * sizeof(double));
    memcpy(b16, host + 256, 16 * sizeof(double));
    return SMVS_OK;
}
