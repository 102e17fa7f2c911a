// Preconditioned conjugate gradient with the matrix resident on the chip.
//
// Same solver as cg.hip (ConjugateGradient::solve, lib/conjugate_gradient.h:72-202,
// on the block stencil of BlockSparseMatrix<4>), different data movement.  The
// streaming kernels of cg.hip read the matrix H (83 MB at 1920x1080, scale 2)
// once per iteration: 150 MB per iteration through L2 / Infinity Cache and two
// launches, ~33 us per iteration however small the active set is.  Here ONE
// launch runs the whole solve:
//
//   * the node grid is cut into <= 256 tiles of <= 512 nodes, one workgroup
//     (512 threads, one per node) per tile, one workgroup per CU;
//   * a thread keeps its node's five stored blocks (the diagonal one as its
//     10 unique entries + the four upper neighbours, 74 doubles = 148 VGPRs)
//     in REGISTERS for the whole solve: the 128 MB register file holds the
//     83 MB matrix, H is read from HBM once per solve instead of once per
//     iteration;
//   * the product uses the symmetry the storage already exploits: thread m
//     forms  B d  for its five blocks (its own rows) and  B^T d_m  for its four
//     upper blocks (the rows of the upper neighbours, handed over through
//     LDS); rows whose lower neighbour lives in another tile use a copy of
//     that neighbour's block kept in LDS (the tile's rim, 18 KB);
//   * what crosses workgroups is tagged data that is its own flag (16-byte
//     pairs {32 data bits, tag} x 2, write-through stores, L1-bypassing
//     loads, no fence): per iteration of the one-exchange solver the q of the
//     rim nodes for the neighbours' halo and eight partial sums per workgroup
//     for a two-level all-reduce; every workgroup sums in the same fixed order,
//     so all derive bit-identical alpha / beta and take the same branch;
//   * inside a workgroup the waves have roles during an exchange (wave 0 the
//     halo, waves 1.. the sums, wave 5 the stores of the sums), the cross-lane
//     sums run on the VALU (v_permlane*_swap, DPP) and the barriers order LDS
//     only -- each of the three for a measured reason, see there.
//
// Numerics: the same operations as the reference except the association of
// the sums (row sums collect the transposed contributions after the stored
// ones; dot products are tree sums, as in cg.hip).  Deterministic: the result
// does not depend on scheduling.  x, the iteration count and `info` are
// checked against the oracle by the same tests as the streaming solver.
//
// Every spin is bounded: if the workgroups are not all resident (another
// barrier kernel holds CUs, fewer CUs than tiles) the barrier times out, the
// solve reports failure and the caller falls back to the streaming kernels.
#include "common.h"

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <thread>

namespace smvs_hip {

constexpr int RES_THREADS = 512;
constexpr int RES_WAVES = RES_THREADS / 64;
constexpr int RES_MAX_BLOCKS = 256;

typedef double double4_r __attribute__((ext_vector_type(4)));

struct ResState {           // mirrors CgState of cg.hip
    double rr, q0, tol, gnorm;
    int iter, done, info, pad;
};

// Exchange area (zeroed before every launch).  A double travels as two
// 8-byte granules {tag = epoch, 32 data bits}: the data is its own flag
// (cdna_hip_programming.md Guideline 16, form R2), so one sweep over the
// granules of all workgroups is barrier and all-reduce at once.
constexpr int RES_KINDS = 8;                 // doubles per all-reduce, at most
constexpr int RES_GROUP = 16;                                  // workgroups per first-level group
constexpr int RES_MAX_GROUPS = RES_MAX_BLOCKS / RES_GROUP;
constexpr int RES_REPLICAS = 16;                               // copies of the group sums
struct ResExchange {
    // flat all-reduce (two-exchange solver): every workgroup sweeps all of these
    unsigned long long gran[2][2 * RES_KINDS][RES_MAX_BLOCKS];   // [parity][..][wg]
    // tree all-reduce (one-exchange solver): a double is the pair {lo, hi} of
    // adjacent granules; the first workgroup of a group of RES_GROUP sums its
    // group's partial sums into lvl2, every workgroup sums the groups
    unsigned long long lvl1[2][RES_MAX_BLOCKS][RES_KINDS][2];    // [parity][wg][kind]
    // (RES_REPLICAS copies of the group sums, 2 KB apart: all 256 workgroups
    // polling the same sixteen cache lines made those lines' memory channel the
    // clock of the second hop -- a poll round there took as long as the channel
    // needed for 4,096 line reads, and a group's store queued behind them)
    unsigned long long lvl2[2][RES_REPLICAS][RES_MAX_GROUPS][RES_KINDS][2];   // [parity][copy][group][kind]
    unsigned timeout;
    // compacted solve: solve tag | 1 (the tile has an active node) or | 2 (it has
    // none: its workgroup has left), written once per solve by every workgroup
    unsigned live[RES_MAX_BLOCKS];
};

// Who takes part in an exchange (the compacted solve, see the kernel): the
// first LIVE workgroup of a group sums the group, members and groups without an
// active node are not waited for -- their sums are exactly +0.0, so leaving
// them out of the tree changes no bit of any total.
struct LiveSet {
    bool leads;         // this workgroup sums its group (wave-uniform)
    unsigned bits;      // lane (kind, j): bit 0 member j of its group is live, bit 1 group j is
};

struct ResArgs {
    const double *H9;        // [5][N][16]
    const double *Pinv;      // [N][16]
    const double *g;         // [N][4]
    double *x, *b;           // [N][4], touched by the owning thread only
    unsigned long long *zg;  // two-exchange solver: [N][4][2] z of the rim nodes as tagged
                             // granules; one-exchange solver: [2][N][4][2] q of the rim
                             // nodes as tagged pairs, double-buffered by iteration parity
    ResExchange *ex;
    ResState *state;         // [2] (state[0] is written at the end)
    int *status;
    int *progress;           // pinned host words, see cg.hip
    int solve_tag;
    int num_nodes, stride, rows;
    int tw, th, tiles_x, num_tiles;
    int max_iterations;
    double q_tolerance, fixed_tolerance;
    long long *trace;        // debug: 100 MHz wall-clock stamps (or nullptr)
    int trace_wg;            // ... of this workgroup's iterations (SMVS_CG_TRACE_WG)
    int pipelined;           // bit 0: launch-ahead Newton loop (update.hip), bit 1:
                             // report a failure (test hook)
    // fused assembly (the Newton loop): per-patch systems instead of H / g / P
    const double *Hp;        // the packed per-patch systems, 36 quads per patch ...
    const double *gp;        // ... and gradients, 4 quads per patch, laid out as
    PatchLayout layout;      // this says (common.h)
    const uint8_t *patch_valid;
    const uint8_t *active;
    uint8_t *active_next;    // cleared for the node update of this step
    double *scalars;
    int npx, npy;
    const double *zeros;     // 16 doubles of +0.0 (a block that does not contribute)
    // polling cadence of the one-exchange solver (units of s_sleep(8) = 512
    // cycles ~ 0.25 us): before the halo's first poll, before a group
    // member's first poll of the group sums, between two polls
    int wait_halo, wait_member, wait_poll;
    int compact;             // leave out what has no active node (SMVS_CG_COMPACT, default 1)
};

// GaussNewtonStep::construct's scatter (gauss_newton_step.cc:88-142) in gather
// form for ONE node, as gn_assemble_kernel does it with four lanes: the <= 4
// incident patches in ascending patch id, local node order 0 (ix, iy),
// 1 (ix+1, iy), 2 (ix, iy+1), 3 (ix+1, iy+1); only stored slots (other node
// >= this node), blocks towards inactive nodes omitted (Q6).
struct NodeSystem {
    double hd[10];      // diagonal block, upper triangle (Q4)
    double hu[4][16];   // slots 5..8
    double g[4];
};

// `count` consecutive quads of patch p's record from element `e0` (a multiple
// of 4) on, or -- unconditionally, so that the loads of a thread stay
// independent of its flags -- the same number of loads from the block of zeros.
struct QuadSource {
    const double4_r *base;
    unsigned step;
    __device__ __forceinline__ double4_r operator[](int i) const { return base[(size_t)i * step]; }
};
__device__ __forceinline__ QuadSource
patch_quads(ResArgs const &A, int p, int e0, bool use)
{
    const double4_r *Hq = reinterpret_cast<const double4_r *>(A.Hp);
    QuadSource q;
    q.base = use ? Hq + ((size_t)(e0 >> 2) * A.layout.hq + (size_t)p * A.layout.hp)
        : reinterpret_cast<const double4_r *>(A.zeros);
    q.step = use ? A.layout.hq : 0u;
    return q;
}
__device__ __forceinline__ double4_r
patch_gradient(ResArgs const &A, int p, int ln, bool use)
{
    const double4_r *gq = reinterpret_cast<const double4_r *>(A.gp);
    return *(use ? gq + ((size_t)ln * A.layout.gq + (size_t)p * A.layout.gp)
        : reinterpret_cast<const double4_r *>(A.zeros));
}

// The loads below are unconditional: a block that does not contribute (patch
// outside the grid / invalid, other node inactive) is read from a block of
// zeros instead, so that all flag loads, then all block loads, are
// independent and in flight together -- with a branch per block every one of
// them cost a full memory latency, ~25 in a row at 2 waves per SIMD.  Adding
// +0.0 in place of an omitted term leaves every sum bit-identical (the sums
// start at +0.0 and can never be -0.0).
__device__ __forceinline__ void
assemble_node(ResArgs const &A, int ix, int iy, bool on, NodeSystem &S)
{
#pragma unroll
    for (int i = 0; i < 10; ++i)
        S.hd[i] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            S.hu[k][i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        S.g[i] = 0.0;
    if (!on)
        return;
    int const n = iy * A.stride + ix;
    int const rows = A.npy + 1;
    // flags of the 3 x 3 nodes around (ix, iy) and of the four incident patches
    bool act[3][3], pv[4];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int const jx = ix + dx, jy = iy + dy;
            bool const inside = jx >= 0 && jx < A.stride && jy >= 0 && jy < rows;
            uint8_t const f = A.active[inside ? jy * A.stride + jx : n];
            act[dy + 1][dx + 1] = inside && f != 0;
        }
    int pidx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int const pxq = ix - (1 - (q & 1));
        int const pyq = iy - (1 - (q >> 1));
        bool const inside = pxq >= 0 && pxq < A.npx && pyq >= 0 && pyq < A.npy;
        pidx[q] = inside ? pyq * A.npx + pxq : 0;
        uint8_t const f = A.patch_valid[pidx[q]];
        pv[q] = inside && f != 0;
    }
    if (!act[1][1])
        return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int const ln = 3 - q;   // local index of the node in that patch
#pragma unroll
        for (int lm = 0; lm < 4; ++lm) {
            if (lm < ln)
                continue;
            // the other node, relative to this one
            int const dx = (lm & 1) - (ln & 1), dy = (lm >> 1) - (ln >> 1);
            bool const use = pv[q] && act[dy + 1][dx + 1];
            if (lm == ln) {
                QuadSource const tri = patch_quads(A, pidx[q], patch_diag_offset(ln), use);
                double4_r const b0 = tri[0], b1 = tri[1], b2 = tri[2];
                S.hd[0] += b0.x; S.hd[1] += b0.y; S.hd[2] += b0.z; S.hd[3] += b0.w;
                S.hd[4] += b1.x; S.hd[5] += b1.y; S.hd[6] += b1.z;
                S.hd[7] += b1.w; S.hd[8] += b2.x;
                S.hd[9] += b2.y;
            } else {
                QuadSource const blk = patch_quads(A, pidx[q], patch_upper_offset(ln, lm), use);
                double4_r const b0 = blk[0], b1 = blk[1], b2 = blk[2], b3 = blk[3];
                int const k = (dy + 1) * 3 + dx + 1 - 5;
                S.hu[k][0] += b0.x; S.hu[k][1] += b0.y; S.hu[k][2] += b0.z; S.hu[k][3] += b0.w;
                S.hu[k][4] += b1.x; S.hu[k][5] += b1.y; S.hu[k][6] += b1.z; S.hu[k][7] += b1.w;
                S.hu[k][8] += b2.x; S.hu[k][9] += b2.y; S.hu[k][10] += b2.z; S.hu[k][11] += b2.w;
                S.hu[k][12] += b3.x; S.hu[k][13] += b3.y; S.hu[k][14] += b3.z; S.hu[k][15] += b3.w;
            }
        }
        double4_r const gv = patch_gradient(A, pidx[q], ln, pv[q]);
        S.g[0] += gv.x; S.g[1] += gv.y; S.g[2] += gv.z; S.g[3] += gv.w;
    }
}

// The diagonal block (upper triangle) and the gradient of node (ix, iy) alone,
// with assemble_node's sums in assemble_node's order: what the one-exchange
// solver needs of a HALO node to form that node's z = P r itself.
__device__ __forceinline__ void
assemble_diagonal(ResArgs const &A, int ix, int iy, double (&hd)[10], double (&g)[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i)
        hd[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        g[i] = 0.0;
    int const n = iy * A.stride + ix;
    uint8_t const fself = A.active[n];
    int pidx[4];
    bool pv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int const pxq = ix - (1 - (q & 1));
        int const pyq = iy - (1 - (q >> 1));
        bool const inside = pxq >= 0 && pxq < A.npx && pyq >= 0 && pyq < A.npy;
        pidx[q] = inside ? pyq * A.npx + pxq : 0;
        uint8_t const f = A.patch_valid[pidx[q]];
        pv[q] = inside && f != 0;
    }
    if (fself == 0)
        return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int const ln = 3 - q;   // local index of the node in that patch
        QuadSource const tri = patch_quads(A, pidx[q], patch_diag_offset(ln), pv[q]);
        double4_r const b0 = tri[0], b1 = tri[1], b2 = tri[2];
        hd[0] += b0.x; hd[1] += b0.y; hd[2] += b0.z; hd[3] += b0.w;
        hd[4] += b1.x; hd[5] += b1.y; hd[6] += b1.z;
        hd[7] += b1.w; hd[8] += b2.x;
        hd[9] += b2.y;
        double4_r const gv = patch_gradient(A, pidx[q], ln, pv[q]);
        g[0] += gv.x; g[1] += gv.y; g[2] += gv.z; g[3] += gv.w;
    }
}

// The four upper blocks (slots 5..8) of node (ix, iy) alone, with
// assemble_node's sums in assemble_node's order.
__device__ __forceinline__ void
assemble_upper(ResArgs const &A, int ix, int iy, bool on, double (&hu)[4][16])
{
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            hu[k][i] = 0.0;
    if (!on)
        return;
    int const n = iy * A.stride + ix;
    int const rows = A.npy + 1;
    bool act[2][3], pv[4];   // nodes (dx, dy) with dy in {0, 1}: the upper slots' ends
#pragma unroll
    for (int dy = 0; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int const jx = ix + dx, jy = iy + dy;
            bool const inside = jx >= 0 && jx < A.stride && jy >= 0 && jy < rows;
            uint8_t const f = A.active[inside ? jy * A.stride + jx : n];
            act[dy][dx + 1] = inside && f != 0;
        }
    int pidx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int const pxq = ix - (1 - (q & 1));
        int const pyq = iy - (1 - (q >> 1));
        bool const inside = pxq >= 0 && pxq < A.npx && pyq >= 0 && pyq < A.npy;
        pidx[q] = inside ? pyq * A.npx + pxq : 0;
        uint8_t const f = A.patch_valid[pidx[q]];
        pv[q] = inside && f != 0;
    }
    if (!act[0][1])
        return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int const ln = 3 - q;   // local index of the node in that patch
#pragma unroll
        for (int lm = 0; lm < 4; ++lm) {
            if (lm <= ln)
                continue;
            int const dx = (lm & 1) - (ln & 1), dy = (lm >> 1) - (ln >> 1);
            bool const use = pv[q] && act[dy][dx + 1];
            QuadSource const blk = patch_quads(A, pidx[q], patch_upper_offset(ln, lm), use);
            double4_r const b0 = blk[0], b1 = blk[1], b2 = blk[2], b3 = blk[3];
            int const k = (dy + 1) * 3 + dx + 1 - 5;
            hu[k][0] += b0.x; hu[k][1] += b0.y; hu[k][2] += b0.z; hu[k][3] += b0.w;
            hu[k][4] += b1.x; hu[k][5] += b1.y; hu[k][6] += b1.z; hu[k][7] += b1.w;
            hu[k][8] += b2.x; hu[k][9] += b2.y; hu[k][10] += b2.z; hu[k][11] += b2.w;
            hu[k][12] += b3.x; hu[k][13] += b3.y; hu[k][14] += b3.z; hu[k][15] += b3.w;
        }
    }
}

// One stored block of another node: row node (mx, my), its upper slot 5..8
// (a compile-time constant at every call: the loops fold to the <= 2 patches
// that hold both nodes).
__device__ __forceinline__ void
assemble_block(ResArgs const &A, int mx, int my, int slot, double *out16)
{
#pragma unroll
    for (int i = 0; i < 16; ++i)
        out16[i] = 0.0;
    int const mrow = my * A.stride + mx;
    bool const act_row = A.active[mrow] != 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int const ln = 3 - q;
#pragma unroll
        for (int lm = 0; lm < 4; ++lm) {
            if (lm <= ln)
                continue;
            int const dx = (lm & 1) - (ln & 1), dy = (lm >> 1) - (ln >> 1);
            if ((dy + 1) * 3 + dx + 1 != slot)
                continue;
            int const pxq = mx - (1 - (q & 1));
            int const pyq = my - (1 - (q >> 1));
            bool const inside = pxq >= 0 && pxq < A.npx && pyq >= 0 && pyq < A.npy;
            int const p = inside ? pyq * A.npx + pxq : 0;
            int const m = inside
                ? pyq * A.stride + pxq + (lm & 1) + (lm >> 1) * A.stride : mrow;
            uint8_t const fp = A.patch_valid[p];
            uint8_t const fm = A.active[m];
            bool const use = act_row && inside && fp != 0 && fm != 0;
            QuadSource const blk = patch_quads(A, p, patch_upper_offset(ln, lm), use);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double4_r const v = blk[i];
                out16[4 * i + 0] += v.x; out16[4 * i + 1] += v.y;
                out16[4 * i + 2] += v.z; out16[4 * i + 3] += v.w;
            }
        }
    }
}

// A rim block for the tile's LDS: the stored block of row node (mx, my)
// towards its neighbour of LOWER slot s seen from the tile node (i.e. the row
// node's upper slot 8 - s), s chosen at run time so that one thread per rim
// block can do the work.  The <= 2 patches holding both nodes, in ascending
// patch id as in assemble_block:
//   s = 0 (slot 8): q 3 (ln 0, lm 3)
//   s = 1 (slot 7): q 2 (ln 1, lm 3), q 3 (ln 0, lm 2)
//   s = 2 (slot 6): q 2 (ln 1, lm 2)
//   s = 3 (slot 5): q 1 (ln 2, lm 3), q 3 (ln 0, lm 1)
// Flags in one batch, the two blocks in one batch (a missing contribution is
// read from the block of zeros).
__device__ __forceinline__ void
assemble_rim_block(ResArgs const &A, int mx, int my, int s, double *dst16)
{
    int const q[2] = { s == 0 ? 3 : (s == 3 ? 1 : 2), 3 };
    int const ln[2] = { s == 0 ? 0 : (s == 3 ? 2 : 1), 0 };
    int const lm[2] = { s == 2 ? 2 : 3, s == 1 ? 2 : 1 };
    bool const two = s == 1 || s == 3;
    int const mrow = my * A.stride + mx;
    uint8_t const frow = A.active[mrow];
    QuadSource src[2];
    uint8_t fp[2], fm[2];
    bool inside[2];
    int p[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        int const pxq = mx - (1 - (q[e] & 1));
        int const pyq = my - (1 - (q[e] >> 1));
        inside[e] = (e == 0 || two) && pxq >= 0 && pxq < A.npx && pyq >= 0
            && pyq < A.npy;
        p[e] = inside[e] ? pyq * A.npx + pxq : 0;
        int const m = inside[e]
            ? pyq * A.stride + pxq + (lm[e] & 1) + (lm[e] >> 1) * A.stride : mrow;
        fp[e] = A.patch_valid[p[e]];
        fm[e] = A.active[m];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        bool const use = frow != 0 && inside[e] && fp[e] != 0 && fm[e] != 0;
        // (ln, lm depend on s at run time here: the offset is computed, not folded)
        src[e] = patch_quads(A, p[e], patch_upper_offset(ln[e], lm[e]), use);
    }
    double4_r v0[4], v1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v0[i] = src[0][i];
        v1[i] = src[1][i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (0 + first) + second, as assemble_block
        double4_r r;
        r.x = (0.0 + v0[i].x) + v1[i].x;
        r.y = (0.0 + v0[i].y) + v1[i].y;
        r.z = (0.0 + v0[i].z) + v1[i].z;
        r.w = (0.0 + v0[i].w) + v1[i].w;
        reinterpret_cast<double4_r *>(dst16)[i] = r;
    }
}

constexpr int TRACE_ITERS = 12, TRACE_POINTS = 12;
// after the iteration rows: four prologue stamps of every workgroup
constexpr int TRACE_BLOCK_BASE = (TRACE_ITERS + 1) * TRACE_POINTS;
// ... and four stamps of every workgroup in iteration TRACE_SKEW_ITER of the
// one-exchange solver (start, sums published, totals received, halo collected):
// how far apart the tiles run
constexpr int TRACE_SKEW_BASE = TRACE_BLOCK_BASE + 4 * RES_MAX_BLOCKS;
constexpr int TRACE_SKEW_ITER = 5;
// ... and eight stamps of every wave of workgroup SMVS_CG_TRACE_WG in the same iteration
constexpr int TRACE_WAVE_BASE = TRACE_SKEW_BASE + 4 * RES_MAX_BLOCKS;
constexpr int TRACE_TOTAL = TRACE_WAVE_BASE + 8 * 8;

__device__ __forceinline__ void
st_agent(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p),
        (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a
// release fence at workgroup scope: s_waitcnt vmcnt(0) in front of the
// s_barrier, so every wave that had published rim values sat at the next
// barrier until the fabric had acknowledged its write-through stores -- 2 to
// 2.8 us in EVERY iteration of the one-exchange solver (cg_trace.py, "sweep
// wave starts"; profiles/r4_cg_barrier.txt).  Nothing in this kernel passes
// data between the threads of a workgroup through global memory: what
// crosses workgroups carries its own tag, everything else is LDS.
__device__ __forceinline__ void
lds_barrier(void)
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A double that crosses workgroups travels as two 8-byte granules {tag, 32
// data bits} (Guideline 16, form R2): the reader polls until both tags carry
// the value it expects, no release / drain on the writer's side.
__device__ __forceinline__ void
st_granules(unsigned long long *g, unsigned tag, double v)
{
    unsigned long long const bits = (unsigned long long)__double_as_longlong(v);
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | (bits & 0xFFFFFFFFull),
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, ((unsigned long long)tag << 32) | (bits >> 32),
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double
ld_agent(const double *p)
{
    unsigned long long const v = __hip_atomic_load(
        reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}

// Cross-lane sums without the LDS pipe.  __shfl_xor of a double is two
// ds_bpermute_b32, and the CU has ONE LDS unit for its eight waves: the eight
// butterflies of an exchange (8 kinds x 6 steps x 2 words x 8 waves = 768
// bpermutes) kept it busy for ~2.5 us per iteration -- the largest single item
// of an iteration, found with the per-wave stamps of tools/cg_trace.py
// (profiles/r4_cg_waves.txt).  gfx950 has what is needed on the VALU:
// v_permlane32_swap / v_permlane16_swap exchange halves / rows between two
// registers, DPP row rotations cover the 16 lanes of a row.
typedef unsigned int uint2_r __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double
join_words(unsigned lo, unsigned hi)
{
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// HALF = 32: on return the lower 32 lanes hold x[l] + x[l + 32], the upper 32
// lanes y[l - 32] + y[l] -- one step of a reduce-scatter over two kinds (with
// y = x: a butterfly step).  HALF = 16: the same between the even and the odd
// rows of 16 lanes.  Every pair is summed as (lower lane) + (upper lane).
template <int HALF>
__device__ __forceinline__ double
swap_add(double x, double y)
{
    unsigned long long const xb = (unsigned long long)__double_as_longlong(x);
    unsigned long long const yb = (unsigned long long)__double_as_longlong(y);
    uint2_r lo, hi;
    if constexpr (HALF == 32) {
        lo = __builtin_amdgcn_permlane32_swap((unsigned)xb, (unsigned)yb, false, false);
        hi = __builtin_amdgcn_permlane32_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32),
            false, false);
    } else {
        lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
        hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32),
            false, false);
    }
    // .x: [x of the lower half | y of the lower half], .y: [x of the upper half |
    // y of the upper half]
    return join_words(lo.x, hi.x) + join_words(lo.y, hi.y);
}

template <int ROR>
__device__ __forceinline__ double
row_rotated(double v)
{
    unsigned long long const b = (unsigned long long)__double_as_longlong(v);
    int const lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x120 + ROR, 0xf, 0xf,
        false);
    int const hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x120 + ROR, 0xf,
        0xf, false);
    return join_words((unsigned)lo, (unsigned)hi);
}

// Sum over the 16 lanes of a row, every lane gets it (bit-identical in all of
// them: after the rotation by 8 the values have period 8, so the two lanes of
// every later pair add the same two numbers).
__device__ __forceinline__ double
row_sum(double v)
{
    v += row_rotated<8>(v);
    v += row_rotated<4>(v);
    v += row_rotated<2>(v);
    v += row_rotated<1>(v);
    return v;
}

// First half of a workgroup sum of K per-thread values: the per-wave sums go
// to red[K][RES_WAVES]; after the barrier inside, the sum of kind k is
// red[k][0] + ... + red[k][RES_WAVES - 1] in that order (block_total).  The
// wave sums are a reduce-scatter: across the halves of the wave a lane keeps
// half of its kinds, across the rows of a half a quarter; what is left (two
// kinds of eight) is summed over the row.  Fixed order, the same in every wave
// and workgroup.
template <int K>
__device__ __forceinline__ void
wave_partials(double const (&v)[K], double *red /*[K][RES_WAVES]*/)
{
    constexpr int P = K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : 8;   // kinds, padded
    static_assert(K <= 8, "block_partials: at most eight kinds");
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
        a[i] = i < K ? v[i] : 0.0;
    // halves of the wave
    constexpr int N1 = P > 1 ? P / 2 : 1;
#pragma unroll
    for (int i = 0; i < N1; ++i)
        a[i] = swap_add<32>(a[i], P > 1 ? a[i + N1] : a[i]);
    // rows of a half
    constexpr int N2 = N1 > 1 ? N1 / 2 : 1;
#pragma unroll
    for (int i = 0; i < N2; ++i)
        a[i] = swap_add<16>(a[i], N1 > 1 ? a[i + N2] : a[i]);
#pragma unroll
    for (int i = 0; i < N2; ++i)
        a[i] = row_sum(a[i]);
    // which kinds this lane's row holds: bit 5 of the lane chose among the
    // halves of a[0 .. P), bit 4 among the halves of what was left
    int const b5 = lane >> 5, b4 = (lane >> 4) & 1;
    int const first = (P > 1 ? b5 * N1 : 0) + (N1 > 1 ? b4 * N2 : 0);
    if ((lane & 15) == 0 && (P > 1 || b5 == 0) && (N1 > 1 || b4 == 0)) {
#pragma unroll
        for (int i = 0; i < N2; ++i)
            if (first + i < K)
                red[(first + i) * RES_WAVES + wave] = a[i];
    }
}

template <int K>
__device__ __forceinline__ void
block_partials(double const (&v)[K], double *red /*[K][RES_WAVES]*/)
{
    wave_partials<K>(v, red);
    lds_barrier();
    // (no second barrier: the partials are next written by the following
    // block_partials, and every thread passes the caller's barrier behind the
    // sweep first)
}

// The same without the workgroup barrier: every wave raises its own tag behind
// its partial sums (LDS serves a wave's requests in order), and only the waves
// that need the workgroup's sums wait for the eight tags -- the others go on
// to their stores and polls.  The slots are safe to reuse: whoever reads them
// does so before the barrier at the end of the exchange, and they are written
// again only behind it.
struct PartialTags {
    volatile unsigned *tag;     // [RES_WAVES]
    __device__ __forceinline__ void raise(unsigned t) const
    {
        if ((threadIdx.x & 63) == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tag[threadIdx.x >> 6] = t;
        }
    }
    // wave-uniform; false after a bounded wait
    __device__ __forceinline__ bool wait(unsigned t) const
    {
        for (unsigned spins = 0; !__all(tag[threadIdx.x & (RES_WAVES - 1)] == t); ++spins) {
            if (spins > (1u << 22))
                return false;
            __builtin_amdgcn_s_sleep(1);
        }
        return true;
    }
};

__device__ __forceinline__ double
block_total(const double *red, int k)
{
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < RES_WAVES; ++wv)
        s += red[k * RES_WAVES + wv];
    return s;
}

struct NoIdleWork {
    __device__ __forceinline__ void operator()() const {}
};

// All-reduce of K doubles over the workgroups of the grid, and the grid-wide
// synchronisation point of the phase: every workgroup publishes its K sums as
// tagged granules; K waves of every workgroup (kind k each, waves FIRST ..
// FIRST + K - 1) sweep the granules of all workgroups until every tag carries
// this epoch, then all sum them in the same fixed order.  The slots are
// double-buffered by epoch parity: a workgroup can publish epoch e + 2 only
// after everybody published e + 1, i.e. after everybody finished reading e.
// The waves that do not sweep run `idle` meanwhile (the halo of the
// one-exchange solver).  Returns false after a bounded wait (a workgroup is
// not resident / gave up).
template <int K, int FIRST, typename Idle>
__device__ __forceinline__ bool
grid_allreduce(ResExchange *ex, unsigned solve_tag, unsigned epoch, int nblocks,
    double (&v)[K], double *red, int *lds_flag, Idle idle)
{
    static_assert(K <= RES_KINDS && FIRST + K <= RES_WAVES, "sweeping waves");
    // tags carry the solve id: granules of earlier solves never match, so the
    // exchange area needs no clearing between solves
    unsigned const tag = solve_tag | epoch;
    double *res = red + RES_KINDS * RES_WAVES;   // [K] results, behind the partial sums
    block_partials<K>(v, red);
    if (nblocks == 1) {
        // a grid of one tile (the coarse scales, the tiny systems of the fuzz
        // sweep): nothing to exchange -- the workgroup's sums are the totals,
        // no granule leaves the CU (6 us per iteration on a loaded chip)
#pragma unroll
        for (int k = 0; k < K; ++k)
            v[k] = block_total(red, k);
        lds_barrier();   // (the partial sums are free for the next reduction)
        return true;
    }
    // (nothing to drain: everything that crosses workgroups is a granule, the
    // vectors of the solve live in registers and LDS)
    unsigned const par = epoch & 1u;
    if (threadIdx.x < 2 * K) {
        // (only the publishing threads need the workgroup's sums)
        int const k = threadIdx.x >> 1, half = threadIdx.x & 1;
        unsigned long long const bits = (unsigned long long)__double_as_longlong(
            block_total(red, k));
        unsigned const word = half ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(&ex->gran[par][threadIdx.x][blockIdx.x],
            ((unsigned long long)tag << 32) | word, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
    }
    int const wave = (int)(threadIdx.x >> 6) - FIRST;
    if (wave >= 0 && wave < K) {
        int const lane = threadIdx.x & 63;
        constexpr int PER_LANE = RES_MAX_BLOCKS / 64;
        unsigned lo[PER_LANE], hi[PER_LANE];
        bool ok = true;
        for (unsigned spins = 0;; ++spins) {
            bool all = true;
#pragma unroll
            for (int j = 0; j < PER_LANE; ++j) {
                int const blk = lane + 64 * j;
                unsigned long long g0 = (unsigned long long)tag << 32, g1 = g0;
                if (blk < nblocks) {
                    g0 = __hip_atomic_load(&ex->gran[par][2 * wave][blk],
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g1 = __hip_atomic_load(&ex->gran[par][2 * wave + 1][blk],
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                lo[j] = (unsigned)g0;
                hi[j] = (unsigned)g1;
                all &= (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag;
            }
            if (__all(all))
                break;
            if (spins > (1u << 18)
                || ((spins & 255u) == 255u
                    && __hip_atomic_load(&ex->timeout, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(&ex->timeout, 1u, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        // fixed order: workgroups lane, lane + 64, ... per lane, then the tree
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            unsigned long long const bits = ((unsigned long long)hi[j] << 32) | lo[j];
            sum += (lane + 64 * j) < nblocks
                ? __longlong_as_double((long long)bits) : 0.0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            sum += __shfl_xor(sum, off);
        if (lane == 0) {
            res[wave] = sum;
            // (a halo wait that gave up raises the same flag)
            if (wave == 0 && __hip_atomic_load(&ex->timeout, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT) != 0u)
                ok = false;
            lds_flag[wave] = ok ? 1 : 0;
        }
    } else {
        idle();
    }
    lds_barrier();
    // The results live apart from the partial sums, so two workgroup barriers
    // per all-reduce are enough (one inside block_partials, this one): results and
    // flags are next written behind the next all-reduce's first barrier, which
    // every thread reaches only after it has read these.
    bool ok = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = res[k];
        ok = ok && lds_flag[k] != 0;
    }
    return ok;
}

// A double as ONE 16-byte write-through store / ONE 16-byte L1-bypassing load
// of its two adjacent granules {data lo, tag}, {data hi, tag}.  Both halves
// carry the tag, so nothing depends on the 16 bytes arriving together; what
// the wide access buys is half the number of fabric transactions of the
// exchange (an 8-byte sc1 store is one fabric write per lane:
// MI355X_MICROARCH.md, "stores of each flavour").
typedef unsigned int uint4_r __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t
pair_buffer(void *base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void
st_pair16(__amdgpu_buffer_rsrc_t buf, unsigned byte_offset, unsigned tag, double v)
{
    unsigned long long const bits = (unsigned long long)__double_as_longlong(v);
    uint4_r const w = { (unsigned)bits, tag, (unsigned)(bits >> 32), tag };
    __builtin_amdgcn_raw_buffer_store_b128(w, buf, (int)byte_offset, 0, AUX_SC1);
}

__device__ __forceinline__ bool
ld_pair16(__amdgpu_buffer_rsrc_t buf, unsigned byte_offset, unsigned tag, double *v)
{
    uint4_r const w = __builtin_amdgcn_raw_buffer_load_b128(buf, (int)byte_offset, 0,
        AUX_SC1);
    *v = __longlong_as_double((long long)(((unsigned long long)w.z << 32) | w.x));
    return w.y == tag && w.w == tag;
}

// The four doubles of one node's exchanged vector (64 bytes): four 16-byte
// loads per poll until all carry `want`; false after a bounded wait.
__device__ __forceinline__ void
nap(int units)
{
    for (int i = 0; i < units; ++i)
        __builtin_amdgcn_s_sleep(8);
}

__device__ __forceinline__ bool
poll_node_pairs(__amdgpu_buffer_rsrc_t buf, unsigned byte_offset, unsigned want,
    ResExchange *ex, double (&out)[4], int gap = 0)
{
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ok &= ld_pair16(buf, byte_offset + (unsigned)q * 16u, want, &out[q]);
        if (ok)
            return true;
        if (spins > (1u << 18)
            || ((spins & 255u) == 255u
                && __hip_atomic_load(&ex->timeout, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(&ex->timeout, 1u, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
        nap(gap);
        asm volatile("" ::: "memory");   // (the loads are re-issued every round)
    }
}

// One double as a pair of adjacent tagged granules: every lane of the wave
// polls its own pair (one 16-byte load) until all lanes see their tag
// (inactive lanes take no part and get 0).  Wave-uniform result: false after a
// bounded wait.
__device__ __forceinline__ bool
poll_pairs(__amdgpu_buffer_rsrc_t buf, unsigned byte_offset, bool active, unsigned tag,
    ResExchange *ex, double *value, int gap = 0, unsigned *rounds = nullptr)
{
    double got = 0.0;
    bool mine_ok = !active;
    bool good = true;
    for (unsigned spins = 0;; ++spins) {
        if (!mine_ok)
            mine_ok = ld_pair16(buf, byte_offset, tag, &got);
        if (__all(mine_ok))
            break;
        if (rounds != nullptr)
            *rounds += 1;
        if (spins > (1u << 18)
            || ((spins & 255u) == 255u
                && __hip_atomic_load(&ex->timeout, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(&ex->timeout, 1u, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            good = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        nap(gap);
        asm volatile("" ::: "memory");
    }
    *value = active ? got : 0.0;
    if (rounds != nullptr)
        *rounds += 1;
    return good;
}

// Sum over the 16 / 32 lanes of an aligned segment, fixed order, every lane gets it.
__device__ __forceinline__ double
segment16_sum(double v)
{
    return row_sum(v);
}

__device__ __forceinline__ double
segment32_sum(double v)
{
    return row_sum(swap_add<16>(v, v));
}

// Who publishes the sums of an exchange.  On gfx9 loads and stores share ONE
// counter (vmcnt): a wave that has issued a write-through store cannot see the
// result of a later load before the fabric has acknowledged that store.  The
// sweeping waves live on their polls, so the workgroup's sums -- and, in a
// group's first workgroup, the group's sums, which the sweeping waves hand
// over through LDS -- are stored by the last wave, which never waits for a
// load.  (The rim's q is published by the owners of the nodes: one wave
// issuing all ~380 write-through stores of a tile was measured and is far
// slower, the issue rate of such stores is what counts there.)
// (the wave of the middle rows of a tile: the fewest rim nodes, so the fewest
// write-through stores of its own in front of the sums)
constexpr int RES_SUM_WAVE = 5;

// LDS mailbox between the sweeping waves and the publishing wave of a group's
// first workgroup: value first, then the tag (LDS serves a wave's requests in
// order), read in the opposite order.
struct GroupMailbox {
    volatile double *value;     // [RES_KINDS]
    volatile unsigned *tag;     // [RES_KINDS]
    __device__ __forceinline__ void put(int kind, unsigned t, double v) const
    {
        value[kind] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tag[kind] = t;
    }
    // (one lane per kind) false after a bounded wait
    __device__ __forceinline__ bool take(int kind, unsigned t, double *v) const
    {
        for (unsigned spins = 0; tag[kind] != t; ++spins) {
            if (spins > (1u << 22))
                return false;
            __builtin_amdgcn_s_sleep(1);
        }
        *v = value[kind];
        return true;
    }
};

__device__ __forceinline__ GroupMailbox
group_mailbox(double *red)
{
    double *base = red + RES_KINDS * RES_WAVES + RES_KINDS + (RES_KINDS + 1) / 2;
    return { base, reinterpret_cast<volatile unsigned *>(base + RES_KINDS) };
}

__device__ __forceinline__ PartialTags
partial_tags(double *red)
{
    double *base = red + RES_KINDS * RES_WAVES + RES_KINDS + (RES_KINDS + 1) / 2
        + RES_KINDS + (RES_KINDS + 1) / 2;
    return { reinterpret_cast<volatile unsigned *>(base) };
}

// All-reduce of K doubles over the workgroups in two levels.  The flat sweep
// above makes every workgroup read every workgroup's K sums: 256 x 256 x K
// granule pairs per exchange, all aimed at the same few KB -- measured, its
// time grows with K (4.4 us for one sum, 6.2 for three, 12 for seven).  Here
// the first workgroup of every group of RES_GROUP sums its group (16 x K
// pairs), publishes the group's sums, and every workgroup sums the <= 16
// groups: 2 x 16 x K pairs per workgroup and exchange instead of 256 x K, two
// hops instead of one.  Lane (kind, j) of the sweeping waves 1 .. (K + 3) / 4
// handles member / group j of one kind; the other waves run `others(wave)`.
// Same guarantees as the flat form: fixed summation order (a tree over the
// members of a group, then a tree over the groups), bit-identical results in
// every workgroup, slots double-buffered by epoch parity, bounded waits.
template <int K, typename Others, typename Mark>
__device__ __forceinline__ bool
grid_allreduce_tree(ResExchange *ex, unsigned solve_tag, unsigned epoch, int nblocks,
    LiveSet const live, double (&v)[K], double *red, int *lds_flag, Others others, Mark mark,
    int wait_member = 6, int wait_poll = 0)
{
    constexpr int SWEEPERS = (K + 3) / 4;
    static_assert(K <= RES_KINDS && 1 + SWEEPERS <= RES_SUM_WAVE, "sweeping waves");
    unsigned const tag = solve_tag | epoch;
    double *res = red + RES_KINDS * RES_WAVES;
    GroupMailbox const box = group_mailbox(red);
    PartialTags const partials = partial_tags(red);
    wave_partials<K>(v, red);
    partials.raise(tag);
    mark(20, -1);
    unsigned const par = epoch & 1u;
    int const b = (int)blockIdx.x;
    __amdgpu_buffer_rsrc_t const xbuf = pair_buffer(ex, sizeof(ResExchange));
    auto lvl1_at = [&](int wg, int kind) {
        return (unsigned)(offsetof(ResExchange, lvl1)
            + ((((size_t)par * RES_MAX_BLOCKS + (size_t)wg) * RES_KINDS + (size_t)kind) * 16));
    };
    auto lvl2_at = [&](int copy, int group, int kind) {
        return (unsigned)(offsetof(ResExchange, lvl2)
            + (((((size_t)par * RES_REPLICAS + (size_t)copy) * RES_MAX_GROUPS + (size_t)group)
                   * RES_KINDS + (size_t)kind) * 16));
    };
    int const ngroups = (nblocks + RES_GROUP - 1) / RES_GROUP;
    bool const leads = live.leads;
    bool const member_live = (live.bits & 1u) != 0u, group_live = (live.bits & 2u) != 0u;
    int const wave = (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave == RES_SUM_WAVE) {
        bool handed = partials.wait(tag);
        mark(21, -1);
        if (nblocks > 1 && lane < K)
            st_pair16(xbuf, lvl1_at(b, lane), tag, block_total(red, lane));
        if (leads) {
            // lane (copy, kind): eight copies per store instruction
            static_assert(K == 8, "eight kinds per copy");
            double part = 0.0;
            handed = box.take(lane & 7, tag, &part) && handed;
#pragma unroll
            for (int c = lane >> 3; c < RES_REPLICAS; c += 8)
                st_pair16(xbuf, lvl2_at(c, b / RES_GROUP, lane & 7), tag, part);
        }
        if (!__all(handed) && lane == 0)
            __hip_atomic_store(&ex->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        others(wave);
    } else if (wave >= 1 && wave <= SWEEPERS) {
        int const kind = 4 * (wave - 1) + (lane >> 4), j = lane & 15;
        bool const kind_ok = kind < K;
        bool ok = true, ok2 = true;
        double total;
        unsigned rounds1 = 0, rounds2 = 0;
        mark(8, -1);
        if (nblocks == 1) {
            // a grid of one tile (the coarse scales): nothing to exchange
            ok = partials.wait(tag);
            total = kind_ok ? block_total(red, kind) : 0.0;
        } else if (ngroups == 1) {
            // a single group: every workgroup sums its <= 16 members itself,
            // one hop instead of two
            ok = poll_pairs(xbuf, lvl1_at(j < nblocks ? j : 0, kind_ok ? kind : 0),
                kind_ok && j < nblocks && member_live, tag, ex, &total, wait_poll);
            total = segment16_sum(total);
        } else {
            if (leads) {
                // this workgroup sums its group
                int const member = b - b % RES_GROUP + j;
                double part;
                ok = poll_pairs(xbuf, lvl1_at(member < nblocks ? member : b,
                        kind_ok ? kind : 0), kind_ok && member < nblocks && member_live, tag,
                    ex, &part, wait_poll, &rounds1);
                part = segment16_sum(part);
                mark(5, -1);
                if (kind_ok && j == 0)
                    box.put(kind, tag, part);
            } else {
                // the group sums cannot be there yet (they are a hop behind)
                nap(wait_member);
            }
            mark(9, -1);
            ok2 = poll_pairs(xbuf, lvl2_at(b % RES_REPLICAS, j < ngroups ? j : 0,
                    kind_ok ? kind : 0),
                kind_ok && j < ngroups && group_live, tag, ex, &total, wait_poll, &rounds2);
            total = segment16_sum(total);
        }
        mark(6, -1);
        mark(10, (long long)(rounds1 * 1000u + rounds2));
        if (kind_ok && j == 0)
            res[kind] = total;
        if (lane == 0) {
            bool flag_ok = ok && ok2;
            // (a wait of another wave that gave up raises the same flag)
            if (wave == 1 && __hip_atomic_load(&ex->timeout, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT) != 0u)
                flag_ok = false;
            lds_flag[wave - 1] = flag_ok ? 1 : 0;
        }
    } else {
        others(wave);
    }
    lds_barrier();
    bool ok = true;
#pragma unroll
    for (int k = 0; k < K; ++k)
        v[k] = res[k];
#pragma unroll
    for (int w = 0; w < SWEEPERS; ++w)
        ok = ok && lds_flag[w] != 0;
    return ok;
}

// Position of a thread in its tile and the addressing that follows from it.
struct TileGeom {
    int tw, th, LW, lx, ly, li, lcore;
    // index of the copy (in the tile's rim array) of the block that the lower
    // neighbour of slot s = 0..3 <-> (dx, dy) = (-1,-1), (0,-1), (1,-1), (-1,0)
    // stores towards this node
    __device__ __forceinline__ int rim_index(int s) const
    {
        int const dx = s == 3 ? -1 : s - 1, dy = s == 3 ? 0 : -1;
        if (ly + dy < 0)
            return s * tw + lx;
        if (lx + dx < 0)
            return 3 * tw + (s == 0 ? 0 : th) + ly;
        return 3 * tw + 2 * th + ly;
    }
};

// acc = (H d)_node for the thread's node: its five stored blocks on the
// direction tile (own rows), the transposed products of its four upper blocks
// handed to the upper neighbours through `yl` one direction at a time (within
// a direction every row receives exactly one contribution: no atomics, fixed
// order), and the rows whose lower neighbour belongs to another tile from the
// copies of that neighbour's blocks in `fb`.  dself = d of the node.  Four
// workgroup barriers; the caller has synchronised the direction tile.
__device__ __forceinline__ void
tile_product(TileGeom const &G, const double *dtile, double *yl, const double *fb,
    double const (&hd)[10], double const (&hu)[4][16], unsigned low, unsigned up,
    bool mine, double (&acc)[4], double (&dself)[4])
{
    int const LW = G.LW, lcore = G.lcore;
    const double4_r *dt = reinterpret_cast<const double4_r *>(dtile);
    {
        double4_r const ds = dt[lcore];
        dself[0] = ds.x; dself[1] = ds.y; dself[2] = ds.z; dself[3] = ds.w;
        {
            // diagonal block from its upper triangle
            double const *d4 = dself;
            acc[0] = hd[0] * d4[0] + hd[1] * d4[1] + hd[2] * d4[2] + hd[3] * d4[3];
            acc[1] = hd[1] * d4[0] + hd[4] * d4[1] + hd[5] * d4[2] + hd[6] * d4[3];
            acc[2] = hd[2] * d4[0] + hd[5] * d4[1] + hd[7] * d4[2] + hd[8] * d4[3];
            acc[3] = hd[3] * d4[0] + hd[6] * d4[1] + hd[8] * d4[2] + hd[9] * d4[3];
        }
#pragma unroll
        for (int s = 5; s < 9; ++s) {
            int const dx = s % 3 - 1, dy = s / 3 - 1;
            double4_r const dm = dt[lcore + dy * LW + dx];
            double const dv[4] = { dm.x, dm.y, dm.z, dm.w };
#pragma unroll
            for (int row = 0; row < 4; ++row)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[row] = __builtin_fma(hu[s - 5][row * 4 + c], dv[c],
                        acc[row]);
        }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if ((up >> kk) & 1u) {
            int const s = 5 + kk;
            int const dx = s % 3 - 1, dy = s / 3 - 1;
            int const target = (G.ly + dy) * G.tw + G.lx + dx;
            double t[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int row = 0; row < 4; ++row)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    t[c] = __builtin_fma(hu[kk][row * 4 + c], dself[row],
                        t[c]);
            double4_r *dst = reinterpret_cast<double4_r *>(
                yl + (size_t)target * 4);
            double4_r const cur = *dst;
            *dst = (double4_r){ cur.x + t[0], cur.y + t[1], cur.z + t[2],
                cur.w + t[3] };
        }
        lds_barrier();
    }
    if (mine) {
        {
            double4_r const cv = *reinterpret_cast<const double4_r *>(
                yl + (size_t)G.li * 4);
            acc[0] += cv.x; acc[1] += cv.y; acc[2] += cv.z; acc[3] += cv.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            unsigned const kind = (low >> (2 * s)) & 3u;
            if (kind == 2u) {
                int const dx = s == 3 ? -1 : s - 1, dy = s == 3 ? 0 : -1;
                double4_r const dm = dt[lcore + dy * LW + dx];
                double const dv[4] = { dm.x, dm.y, dm.z, dm.w };
                const double4_r *blk = reinterpret_cast<const double4_r *>(
                    fb + (size_t)G.rim_index(s) * 16);
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    double4_r const br = blk[row];
                    acc[0] = __builtin_fma(br.x, dv[row], acc[0]);
                    acc[1] = __builtin_fma(br.y, dv[row], acc[1]);
                    acc[2] = __builtin_fma(br.z, dv[row], acc[2]);
                    acc[3] = __builtin_fma(br.w, dv[row], acc[3]);
                }
            }
        }
    }
}

// The symmetric 4 x 4 block of node `li` from its upper triangle in three
// planes (see resident_lds_layout).
__device__ __forceinline__ void
load_symmetric(const double *planes, int tile_nodes, int li, double (&P)[4][4])
{
    const double4_r *Pu = reinterpret_cast<const double4_r *>(planes);
    double4_r const a = Pu[li], b = Pu[tile_nodes + li], c = Pu[2 * tile_nodes + li];
    P[0][0] = a.x; P[0][1] = a.y; P[0][2] = a.z; P[0][3] = a.w;
    P[1][0] = a.y; P[1][1] = b.x; P[1][2] = b.y; P[1][3] = b.z;
    P[2][0] = a.z; P[2][1] = b.y; P[2][2] = b.w; P[2][3] = c.x;
    P[3][0] = a.w; P[3][1] = b.z; P[3][2] = c.x; P[3][3] = c.y;
}

// z = P r with the reference's operation order (block_sparse_matrix.h:276-298
// on a diagonal block: products added one by one, no contraction)
__device__ __forceinline__ void
precondition(double const (&P)[16], double const (&r)[4], double (&z)[4])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        double zi = 0.0;
        zi += P[row * 4 + 0] * r[0];
        zi += P[row * 4 + 1] * r[1];
        zi += P[row * 4 + 2] * r[2];
        zi += P[row * 4 + 3] * r[3];
        z[row] = zi;
    }
}

// Polls the eight granules of one node's exchanged vector until all carry
// `want`; false after a bounded wait (the exchange's timeout flag is raised).
__device__ __forceinline__ bool
poll_node_granules(const unsigned long long *src, unsigned want, ResExchange *ex,
    double (&out)[4])
{
    for (unsigned spins = 0;; ++spins) {
        unsigned long long g[8];
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            g[q] = __hip_atomic_load(src + q, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            ok &= (unsigned)(g[q] >> 32) == want;
        }
        if (ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[q] = __longlong_as_double((long long)(
                    (g[2 * q + 1] << 32) | (g[2 * q] & 0xFFFFFFFFull)));
            return true;
        }
        if (spins > (1u << 18)) {
            __hip_atomic_store(&ex->timeout, 1u, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// LDS carve (doubles) of the two solver variants; the host sizes the launch
// with the same function.
struct ResLds {
    size_t dtile, yl, Pl, rl, fb, xl, bl, rh, Ph, qh, hinfo, red, live, total;
};
__host__ __device__ __forceinline__ ResLds
resident_lds_layout(int tw, int th, bool one)
{
    size_t const tn = (size_t)tw * th;
    size_t const ring = (size_t)2 * (tw + 2) + 2 * th;
    ResLds L;
    size_t o = 0;
    L.dtile = o; o += (size_t)(tw + 2) * (th + 2) * 4;   // direction tile with halo
    L.yl = o; o += tn * 4;                               // row sums from below
    // P: four row planes (two-exchange solver), or its upper triangle in three
    // planes {p00 p01 p02 p03} {p11 p12 p13 p22} {p23 p33 - -} (one-exchange
    // solver: ldl_inverse4 forms P[c1][c2] and P[c2][c1] from the same products
    // in the same order, so P is symmetric to the bit) -- the 16 KB saved hold
    // r there, which frees eight registers in a kernel that spills
    L.Pl = o; o += (one ? 3 : 4) * tn * 4;
    L.rl = o; o += one ? tn * 4 : 0;                     // r (one-exchange solver)
    L.fb = o; o += (size_t)(3 * tw + 3 * th) * 16;       // rim blocks
    L.xl = o; o += tn * 4;                               // x
    L.bl = o; o += one ? 0 : tn * 4;                     // b (two-exchange solver only)
    L.rh = o; o += one ? ring * 4 : 0;                   // r of the halo nodes
    L.Ph = o; o += one ? ring * 16 : 0;                  // P of the halo nodes
    L.qh = o; o += one ? ring * 4 : 0;                   // q of the halo nodes (one iteration)
    L.hinfo = o; o += one ? (ring + 1) / 2 : 0;          // per halo slot: node id (ints)
    L.red = o; o += RES_KINDS * RES_WAVES + RES_KINDS;   // partial sums + results
    o += (RES_KINDS + 1) / 2;                            // flags (ints)
    o += RES_KINDS + (RES_KINDS + 1) / 2;                // group mailbox: values, tags (ints)
    o += (RES_WAVES + 1) / 2;                            // tags of the waves' partial sums (ints)
    L.live = o; o += RES_MAX_BLOCKS / 64;                // which tiles have an active node (bits)
    L.total = o;
    return L;
}

// ONE = false: the reference's operation order -- d.Ad is reduced first, then
// r.r, x.(b + r), z.r of the updated vectors: two grid-wide exchanges per
// iteration (conjugate_gradient.h:121-198 line by line).
// ONE = true: one exchange per iteration.  Before the step length is known
// the workgroups reduce eight dot products of the CURRENT vectors,
//   d.q, r.q, q.q, 2 w.r, w.q, d.r, r.r, z.r    (q = H d, w = P q),
// from which alpha and the three scalars the reference tests follow exactly
// (no approximation, only the association of the sums changes):
//   r'.r'      = r.r - 2 alpha r.q + alpha^2 q.q          (r' = r - alpha q)
//   z'.r'      = z.r - 2 alpha w.r + alpha^2 w.q          (z' = P r' = z - alpha w;
//                                                          z.q = w.r, P symmetric)
//   x'.(b+r')  = x.(b+r) + 2 alpha d.r - alpha^2 d.q      (x' = x + alpha d,
//                                                          b - r = H x)
// r.r and z.r are summed directly in every iteration and only extrapolated
// ONE step (carried by recurrence over a whole solve, z.r kept the absolute
// rounding errors of its large early values while it shrank by orders of
// magnitude: iteration counts on ill-conditioned systems moved by up to 10 %).
// The vectors themselves are updated with the reference's operations (z = P r
// is recomputed, not taken from the recurrence).  The second grid-wide wait of
// an iteration -- the z of the neighbouring tiles' rim nodes -- disappears as
// well: a tile keeps r and P of its halo nodes and receives their q (published
// BEFORE the all-reduce and collected by the wave that does not sweep, while
// the sweep runs), so it forms the halo's z and d itself, bit-identical with
// the owner's.
template <bool FUSED, bool ONE, bool TRACE>
__global__ void __launch_bounds__(RES_THREADS, 2)
cg_resident_kernel(ResArgs A)
{
    // (the stamps of tools/cg_trace.py cost registers in a kernel that has none
    // to spare: they are compiled into instantiations of their own)
    bool const tracing = TRACE && A.trace != nullptr;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // launch-ahead Newton loop: the loop ended (or a solve gave up) while this
    // launch was already enqueued -- every workgroup reads the same words
    if (A.pipelined && (A.status[I_STOP] | A.status[I_STEP_ABORT]) != 0)
        return;
    int const tw = A.tw, th = A.th;
    int const LW = tw + 2, LH = th + 2;
    int const tile_nodes = tw * th;
    ResLds const L = resident_lds_layout(tw, th, ONE);
    double *dtile = lds + L.dtile;                        // [LH*LW][4]
    double *yl = lds + L.yl;                              // [tile_nodes][4]
    double *Pl = lds + L.Pl;                              // [4 | 3][tile_nodes][4]
    double *rl = lds + L.rl;                              // [tile_nodes][4] (ONE)
    double *fb = lds + L.fb;                              // [3*tw + 3*th][16]
    double *xl = lds + L.xl;                              // [tile_nodes][4]
    double *bl = lds + L.bl;                              // [tile_nodes][4] (!ONE)
    double *rh = lds + L.rh;                              // [ring][4] (ONE)
    double *Ph = lds + L.Ph;                              // [ring][16] (ONE)
    double *qhl = lds + L.qh;                             // [ring][4] (ONE)
    int *hnode = reinterpret_cast<int *>(lds + L.hinfo);  // [ring] (ONE)
    double *red = lds + L.red;
    int *flag = reinterpret_cast<int *>(red + RES_KINDS * RES_WAVES + RES_KINDS);

    int const tid = threadIdx.x;
    int const nblocks = (int)gridDim.x;
    int const tile = (int)blockIdx.x;
    int const tiles_y = (A.rows + th - 1) / th;
    int const ty = tile / A.tiles_x;
    int const tx = tile - ty * A.tiles_x;
    auto tile_exists = [&](int x, int y) {
        return x >= 0 && x < A.tiles_x && y >= 0 && y < tiles_y;
    };
    int const lx = tid % tw, ly = tid / tw;
    int const gx = tx * tw + lx, gy = ty * th + ly;
    bool const mine = tid < tile_nodes && gx < A.stride && gy < A.rows;
    int const n = mine ? gy * A.stride + gx : 0;
    int const lcore = (ly + 1) * LW + lx + 1;
    int const li = ly * tw + lx;             // index in the tile
    TileGeom const G = { tw, th, LW, lx, ly, li, lcore };
    // the exchanged vector of a rim node is read by the neighbouring tiles
    // (a grid of one tile has no neighbours: nothing is published)
    // ---- the compacted solve (fused one-exchange kernel) ----
    // The reference's system holds the active nodes only
    // (gauss_newton_step.cc:73-79, 91-105).  Here the rows of an inactive node
    // are zero, and so are its r, z, d, q and its terms of every sum -- exactly:
    //   * an inactive rim node publishes nothing and nobody polls for it (the
    //     neighbour takes q = 0, what the owner would have sent);
    //   * a tile without an active node says so in `ex->live` and LEAVES before
    //     it assembles anything: its CU is free, the all-reduce neither waits for
    //     it nor reads its slots (the first live member of a group sums it).
    // Both leave every total, every iterate and the iteration count bit-identical
    // (SMVS_CG_COMPACT=0 runs the full grid; tests/test_gpu_parity.py compares).
    bool compact = FUSED && ONE && !TRACE && nblocks > 1 && A.compact != 0;
    unsigned long long *livemap = reinterpret_cast<unsigned long long *>(lds + L.live);
    unsigned const live_tag = (unsigned)A.solve_tag;
    // every thread of the first RES_MAX_BLOCKS polls one tile's word; false after
    // a bounded wait.  On return the map is in LDS.
    auto gather_live_map = [&]() -> bool {
        bool ok = true;
        if (tid < RES_MAX_BLOCKS) {
            bool const need = tid < nblocks;
            unsigned w = 0u;
            for (unsigned spins = 0; need; ++spins) {
                w = __hip_atomic_load(&A.ex->live[tid], __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT);
                if ((w & 0xFFFF0000u) == live_tag && (w & 3u) != 0u)
                    break;
                if (spins > (1u << 18)) {
                    __hip_atomic_store(&A.ex->timeout, 1u, __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    w = 0u;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            unsigned long long const m = __ballot(need && (w & 1u) != 0u);
            if ((tid & 63) == 0)
                livemap[tid >> 6] = m;
        }
        // (workgroup-wide AND through the flag words of the all-reduce, which
        // are free outside an exchange -- __syncthreads_and would bring static
        // LDS into a kernel whose dynamic LDS is sized to the CU's 160 KB)
        bool const wave_ok = __all(ok);
        if ((tid & 63) == 0)
            flag[tid >> 6] = wave_ok ? 1 : 0;
        lds_barrier();
        bool all_ok = true;
#pragma unroll
        for (int wv = 0; wv < RES_WAVES; ++wv)
            all_ok = all_ok && flag[wv] != 0;
        lds_barrier();
        return all_ok;
    };
    auto tile_live = [&](int t) {
        return ((livemap[t >> 6] >> (t & 63)) & 1ull) != 0ull;
    };
    bool node_on = true;        // FUSED: the thread's node is active
    if (compact) {
        node_on = mine && A.active[n] != 0;
        bool const wave_any = __any(node_on);
        if ((tid & 63) == 0)
            flag[tid >> 6] = wave_any ? 1 : 0;
        lds_barrier();
        bool any = false;
#pragma unroll
        for (int wv = 0; wv < RES_WAVES; ++wv)
            any = any || flag[wv] != 0;
        lds_barrier();
        if (tid == 0)
            __hip_atomic_store(&A.ex->live[tile], live_tag | (any ? 1u : 2u), __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
        if (!any) {
            // what the prologue and the end of the solve do for the tile's nodes:
            // clear the flags of the step's node update, delta = 0, b = -g
            if (mine) {
                A.active_next[n] = 0;
                *reinterpret_cast<double4_r *>(A.x + (size_t)n * 4)
                    = (double4_r){ 0.0, 0.0, 0.0, 0.0 };
                *reinterpret_cast<double4_r *>(A.b + (size_t)n * 4)
                    = (double4_r){ -0.0, -0.0, -0.0, -0.0 };
            }
            if (tile == 0 && tid == 0) {
                A.status[I_ACTIVE_PATCHES] = A.status[I_LIVE_PATCHES];
                A.status[I_NUM_ACTIVE] = 0;
                A.scalars[S_SUMDIFF] = 0.0;
                A.scalars[S_COUNT_DIFF] = 0.0;
            }
            bool const got = gather_live_map();
            bool any_live = false;
            for (int w = 0; w < RES_MAX_BLOCKS / 64; ++w)
                any_live = any_live || livemap[w] != 0ull;
            if (!got || any_live)
                return;
            // no tile has an active node: a system of zeros, which the full grid
            // runs the way it always has (every workgroup is here and stays)
            compact = false;
            node_on = true;
        }
    }
    bool const rim = nblocks > 1 && mine && (!compact || node_on)
        && (lx == 0 || lx == tw - 1 || ly == 0 || ly == th - 1);
    size_t const N = (size_t)A.num_nodes;

    // halo ring position served by this thread
    int const ring = 2 * LW + 2 * th;
    bool const has_halo = tid < ring;
    int hx = 0, hy = 0;
    if (tid < LW) {
        hx = tid; hy = 0;
    } else if (tid < 2 * LW) {
        hx = tid - LW; hy = LH - 1;
    } else if (tid < 2 * LW + th) {
        hx = 0; hy = tid - 2 * LW + 1;
    } else if (has_halo) {
        hx = LW - 1; hy = tid - 2 * LW - th + 1;
    }
    int halo_node = -1, halo_ix = 0, halo_iy = 0;
    if (has_halo && tile_exists(tx, ty)) {
        halo_ix = tx * tw + hx - 1;
        halo_iy = ty * th + hy - 1;
        if (halo_ix >= 0 && halo_ix < A.stride && halo_iy >= 0 && halo_iy < A.rows)
            halo_node = halo_iy * A.stride + halo_ix;
    }
    int const lhalo = hy * LW + hx;
    if (tracing && blockIdx.x == 0 && tid == 0)
        A.trace[0] = (long long)wall_clock64();
    if (tracing && tid == 0)
        A.trace[TRACE_BLOCK_BASE + 4 * blockIdx.x + 0] = (long long)wall_clock64();
    // tag of the exchanged vectors: the solve id and the iteration that reads them
    unsigned const ztag = (unsigned)A.solve_tag;
    // block-Jacobi preconditioner (block_sparse_matrix.h:300-316): the
    // inverted diagonal block, kept un-inverted on NaN / zero pivot (Q12)
    auto invert_diagonal = [](double const (&d10)[10], double (&Pfull)[16]) {
        double const full[16] = { d10[0], d10[1], d10[2], d10[3],
            d10[1], d10[4], d10[5], d10[6], d10[2], d10[5], d10[7], d10[8],
            d10[3], d10[6], d10[8], d10[9] };
#pragma unroll
        for (int i = 0; i < 16; ++i)
            Pfull[i] = full[i];
        ldl_inverse4(Pfull);
        bool nancheck = false;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            nancheck |= isnan(Pfull[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            Pfull[i] = nancheck ? full[i] : Pfull[i];
    };
    // ---- the matrix: five stored blocks in registers, the rim in LDS ----
    // (the diagonal block is symmetric, Q4: its upper triangle, 10 doubles)
    double hd[10];
    double hu[4][16];
    double gnode[4];          // gradient of the node
    double r[4], z[4];
    double v0[2] = { 0.0, 0.0 };   // z.r and g.g of this thread's node
    // x = 0, r = b = -g, z = P r of the thread's node (conjugate_gradient.h:86-118)
    // from hd / gnode; P goes to LDS
    auto init_own = [&]() {
        if (mine) {
            double const gg[4] = { gnode[0], gnode[1], gnode[2], gnode[3] };
            double Pfull[16];
            if (FUSED) {
                invert_diagonal(hd, Pfull);
            } else {
                const double4_r *P = reinterpret_cast<const double4_r *>(
                    A.Pinv + (size_t)n * 16);
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    double4_r const p = P[row];
                    Pfull[row * 4 + 0] = p.x; Pfull[row * 4 + 1] = p.y;
                    Pfull[row * 4 + 2] = p.z; Pfull[row * 4 + 3] = p.w;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r[k] = -gg[k];
            if (ONE) {
                double4_r *Pu = reinterpret_cast<double4_r *>(Pl);
                Pu[li] = (double4_r){ Pfull[0], Pfull[1], Pfull[2], Pfull[3] };
                Pu[tile_nodes + li] = (double4_r){ Pfull[5], Pfull[6], Pfull[7], Pfull[10] };
                Pu[2 * tile_nodes + li] = (double4_r){ Pfull[11], Pfull[15], 0.0, 0.0 };
                *reinterpret_cast<double4_r *>(rl + (size_t)li * 4)
                    = (double4_r){ r[0], r[1], r[2], r[3] };
            } else {
#pragma unroll
                for (int row = 0; row < 4; ++row)
                    *reinterpret_cast<double4_r *>(Pl + ((size_t)row * tile_nodes + li) * 4)
                        = (double4_r){ Pfull[row * 4 + 0], Pfull[row * 4 + 1],
                            Pfull[row * 4 + 2], Pfull[row * 4 + 3] };
            }
            precondition(Pfull, r, z);
            *reinterpret_cast<double4_r *>(xl + (size_t)li * 4)
                = (double4_r){ 0.0, 0.0, 0.0, 0.0 };
            if (ONE) {
                // b goes to HBM now (it is not needed again); d_1 = z_0
                *reinterpret_cast<double4_r *>(A.b + (size_t)n * 4)
                    = (double4_r){ r[0], r[1], r[2], r[3] };
                reinterpret_cast<double4_r *>(dtile)[lcore]
                    = (double4_r){ z[0], z[1], z[2], z[3] };
                *reinterpret_cast<double4_r *>(yl + (size_t)li * 4)
                    = (double4_r){ 0.0, 0.0, 0.0, 0.0 };
            } else {
                *reinterpret_cast<double4_r *>(bl + (size_t)li * 4)
                    = (double4_r){ r[0], r[1], r[2], r[3] };
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!ONE && rim)
                    st_granules(A.zg + ((size_t)n * 4 + k) * 2, ztag + 1u, z[k]);
                v0[0] += z[k] * r[k];
                v0[1] += gg[k] * gg[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r[k] = z[k] = 0.0;
        }
    };
    if constexpr (ONE) {
        // Order of the one-exchange prologue: halo, own diagonal block and P,
        // then the upper blocks -- the LDL^T inverses run before the 128
        // registers of the upper blocks are occupied.
        for (int i = tid; i < LH * LW * 4; i += RES_THREADS)
            dtile[i] = 0.0;   // (out-of-grid halo stays zero for the whole solve)
        // (LDS is not cleared between launches: a word a previous kernel left
        // where the exchange keeps its tags must not look like one of this
        // solve's -- tags are never 0)
        if (tid < RES_KINDS)
            group_mailbox(red).tag[tid] = 0u;
        if (tid < RES_WAVES)
            partial_tags(red).tag[tid] = 0u;
        lds_barrier();
        if (has_halo) {
            // (compacted: nobody publishes an inactive node's q -- it is zero)
            hnode[tid] = halo_node >= 0 && compact && A.active[halo_node] == 0 ? -1 : halo_node;
            // r and P of the halo node, with the operations of its owner
            double Ph16[16], gh[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int i = 0; i < 16; ++i)
                Ph16[i] = 0.0;
            if (halo_node >= 0) {
                if (FUSED) {
                    double hdh[10];
                    assemble_diagonal(A, halo_ix, halo_iy, hdh, gh);
                    invert_diagonal(hdh, Ph16);
                } else {
                    const double4_r *P = reinterpret_cast<const double4_r *>(
                        A.Pinv + (size_t)halo_node * 16);
#pragma unroll
                    for (int row = 0; row < 4; ++row) {
                        double4_r const p = P[row];
                        Ph16[row * 4 + 0] = p.x; Ph16[row * 4 + 1] = p.y;
                        Ph16[row * 4 + 2] = p.z; Ph16[row * 4 + 3] = p.w;
                    }
                    double4_r const gv = *reinterpret_cast<const double4_r *>(
                        A.g + (size_t)halo_node * 4);
                    gh[0] = gv.x; gh[1] = gv.y; gh[2] = gv.z; gh[3] = gv.w;
                }
            }
            double rhv[4], zh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rhv[k] = -gh[k];
            precondition(Ph16, rhv, zh);
            *reinterpret_cast<double4_r *>(rh + (size_t)tid * 4)
                = (double4_r){ rhv[0], rhv[1], rhv[2], rhv[3] };
#pragma unroll
            for (int row = 0; row < 4; ++row)
                *reinterpret_cast<double4_r *>(Ph + (size_t)tid * 16 + row * 4)
                    = (double4_r){ Ph16[row * 4 + 0], Ph16[row * 4 + 1],
                        Ph16[row * 4 + 2], Ph16[row * 4 + 3] };
            if (halo_node >= 0)
                reinterpret_cast<double4_r *>(dtile)[lhalo]
                    = (double4_r){ zh[0], zh[1], zh[2], zh[3] };
        }
        if (FUSED) {
            if (mine) {
                assemble_diagonal(A, gx, gy, hd, gnode);
                A.active_next[n] = 0;
            } else {
#pragma unroll
                for (int i = 0; i < 10; ++i)
                    hd[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    gnode[i] = 0.0;
            }
            if (blockIdx.x == 0 && tid == 0) {
                A.status[I_ACTIVE_PATCHES] = A.status[I_LIVE_PATCHES];
                A.status[I_NUM_ACTIVE] = 0;
                A.scalars[S_SUMDIFF] = 0.0;
                A.scalars[S_COUNT_DIFF] = 0.0;
            }
        } else {
            const double4_r *src = reinterpret_cast<const double4_r *>(
                A.H9 + (size_t)n * 16);
            double4_r const zero4 = { 0, 0, 0, 0 };
            double4_r const r0 = mine ? src[0] : zero4, r1 = mine ? src[1] : zero4,
                r2 = mine ? src[2] : zero4, r3 = mine ? src[3] : zero4;
            hd[0] = r0.x; hd[1] = r0.y; hd[2] = r0.z; hd[3] = r0.w;
            hd[4] = r1.y; hd[5] = r1.z; hd[6] = r1.w;
            hd[7] = r2.z; hd[8] = r2.w;
            hd[9] = r3.w;
            double4_r const gv = mine ? *reinterpret_cast<const double4_r *>(
                A.g + (size_t)n * 4) : zero4;
            gnode[0] = gv.x; gnode[1] = gv.y; gnode[2] = gv.z; gnode[3] = gv.w;
        }
        init_own();
        if (FUSED) {
            assemble_upper(A, gx, gy, mine, hu);
        } else {
#pragma unroll
            for (int s = 1; s < 5; ++s) {
                const double4_r *src = reinterpret_cast<const double4_r *>(
                    A.H9 + ((size_t)s * N + (size_t)n) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double4_r const v = mine ? src[q] : (double4_r){ 0, 0, 0, 0 };
                    hu[s - 1][4 * q + 0] = v.x; hu[s - 1][4 * q + 1] = v.y;
                    hu[s - 1][4 * q + 2] = v.z; hu[s - 1][4 * q + 3] = v.w;
                }
            }
        }
    } else if (FUSED) {
        // assembled here from the per-patch systems: H, g and P never go to
        // HBM.  Also what the assembly kernel does for the step's node update.
        NodeSystem S;
        assemble_node(A, gx, gy, mine, S);
#pragma unroll
        for (int i = 0; i < 10; ++i)
            hd[i] = S.hd[i];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                hu[k][i] = S.hu[k][i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gnode[i] = S.g[i];
        if (mine)
            A.active_next[n] = 0;
        if (blockIdx.x == 0 && tid == 0) {
            A.status[I_ACTIVE_PATCHES] = A.status[I_LIVE_PATCHES];
            A.status[I_NUM_ACTIVE] = 0;
            A.scalars[S_SUMDIFF] = 0.0;
            A.scalars[S_COUNT_DIFF] = 0.0;
        }
    } else {
        {
            const double4_r *src = reinterpret_cast<const double4_r *>(
                A.H9 + (size_t)n * 16);
            double4_r const zero4 = { 0, 0, 0, 0 };
            double4_r const r0 = mine ? src[0] : zero4, r1 = mine ? src[1] : zero4,
                r2 = mine ? src[2] : zero4, r3 = mine ? src[3] : zero4;
            hd[0] = r0.x; hd[1] = r0.y; hd[2] = r0.z; hd[3] = r0.w;
            hd[4] = r1.y; hd[5] = r1.z; hd[6] = r1.w;
            hd[7] = r2.z; hd[8] = r2.w;
            hd[9] = r3.w;
        }
#pragma unroll
        for (int s = 1; s < 5; ++s) {
            const double4_r *src = reinterpret_cast<const double4_r *>(
                A.H9 + ((size_t)s * N + (size_t)n) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double4_r const v = mine ? src[q] : (double4_r){ 0, 0, 0, 0 };
                hu[s - 1][4 * q + 0] = v.x; hu[s - 1][4 * q + 1] = v.y;
                hu[s - 1][4 * q + 2] = v.z; hu[s - 1][4 * q + 3] = v.w;
            }
        }
        double4_r const gv = mine ? *reinterpret_cast<const double4_r *>(
            A.g + (size_t)n * 4) : (double4_r){ 0, 0, 0, 0 };
        gnode[0] = gv.x; gnode[1] = gv.y; gnode[2] = gv.z; gnode[3] = gv.w;
    }
    if (tracing && blockIdx.x == 0 && tid == 0)
        A.trace[2] = (long long)wall_clock64();   // own blocks in registers
    if (tracing && tid == 0)
        A.trace[TRACE_BLOCK_BASE + 4 * blockIdx.x + 1] = (long long)wall_clock64();
    // `low` holds two bits per lower slot s = 0..3 <-> (dx, dy) = (-1,-1),
    // (0,-1), (1,-1), (-1,0): 0 neighbour outside the grid, 1 neighbour inside
    // the tile (its contribution arrives through `yl`), 2 the neighbour's
    // block sits in the rim; `up` one bit per upper slot: the neighbour's row
    // lives in this tile.
    unsigned low = 0u, up = 0u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int const dx = s == 3 ? -1 : s - 1, dy = s == 3 ? 0 : -1;
        int const nx = gx + dx, ny = gy + dy;
        if (mine && nx >= 0 && nx < A.stride && ny >= 0) {
            int const mlx = lx + dx, mly = ly + dy;
            bool const in_tile = mly >= 0 && mlx >= 0 && mlx < tw;
            low |= (in_tile ? 1u : 2u) << (2 * s);
            if (!in_tile) {
                // block (row m, col n) is stored at m under its upper slot 8 - s
                double *dst = fb + (size_t)G.rim_index(s) * 16;
                if (FUSED) {
                    // (assembled below, one thread per rim block)
                } else {
                    int const m = ny * A.stride + nx;
                    const double4_r *src = reinterpret_cast<const double4_r *>(
                        A.H9 + ((size_t)(8 - s - 4) * N + (size_t)m) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        reinterpret_cast<double4_r *>(dst)[e] = src[e];
                }
            }
        }
    }
    if (FUSED) {
        // The rim blocks, one thread each (with the tile node's own thread
        // doing up to three of them one after the other this phase was a
        // chain of ~8 memory latencies for a handful of threads).  The index
        // decodes as rim_index encodes: top row, left column (two slots),
        // right column.
        for (int r = tid; r < 3 * tw + 3 * th; r += RES_THREADS) {
            int rs, rlx, rly;
            if (r < 3 * tw) {
                rs = r / tw; rlx = r - rs * tw; rly = 0;
            } else if (r < 3 * tw + th) {
                rs = 0; rlx = 0; rly = r - 3 * tw;
            } else if (r < 3 * tw + 2 * th) {
                rs = 3; rlx = 0; rly = r - 3 * tw - th;
            } else {
                rs = 2; rlx = tw - 1; rly = r - 3 * tw - 2 * th;
            }
            int const dx = rs == 3 ? -1 : rs - 1, dy = rs == 3 ? 0 : -1;
            int const ngx = tx * tw + rlx, ngy = ty * th + rly;   // the tile node
            int const nx = ngx + dx, ny = ngy + dy;               // the row node
            bool const tile_node = ngx < A.stride && ngy < A.rows;
            bool const in_tile = rly + dy >= 0 && rlx + dx >= 0 && rlx + dx < tw;
            // (the column entries with rly = 0 belong to the top row's indices)
            bool const dup = r >= 3 * tw && r != 3 * tw + th && rly == 0;
            if (tile_node && !in_tile && !dup && nx >= 0 && nx < A.stride && ny >= 0)
                assemble_rim_block(A, nx, ny, rs, fb + (size_t)r * 16);
        }
    }
    if (tracing && blockIdx.x == 0 && tid == 0)
        A.trace[3] = (long long)wall_clock64();   // rim blocks in LDS
    if (tracing && tid == 0)
        A.trace[TRACE_BLOCK_BASE + 4 * blockIdx.x + 2] = (long long)wall_clock64();
    // upper slots 5..8 <-> (dx, dy) = (1,0), (-1,1), (0,1), (1,1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int const s = 5 + k;
        int const dx = s % 3 - 1, dy = s / 3 - 1;
        int const mlx = lx + dx, mly = ly + dy;
        bool const in_tile = mine && mlx >= 0 && mlx < tw && mly < th
            && gx + dx < A.stride && gy + dy < A.rows;
        up |= (in_tile ? 1u : 0u) << k;
    }
    unsigned epoch = 1;
    bool alive = true;
    int reporter = 0;       // the workgroup that writes the result (compacted: the first live one)
    bool report_all = false;
    ResState st;
    if constexpr (!ONE) {
        // zero the direction tile (out-of-grid halo stays zero for the whole solve)
        for (int i = tid; i < LH * LW * 4; i += RES_THREADS)
            dtile[i] = 0.0;
        init_own();
    }
    if (tracing && blockIdx.x == 0 && tid == 0)
        A.trace[4] = (long long)wall_clock64();   // P, r, z done
    if (tracing && tid == 0)
        A.trace[TRACE_BLOCK_BASE + 4 * blockIdx.x + 3] = (long long)wall_clock64();
    st.q0 = -0.0;
    st.iter = 1;
    st.done = 0;
    st.info = SMVS_CG_MAX_ITERATIONS;
    st.pad = 0;
    st.rr = st.tol = st.gnorm = 0.0;

    auto stamp = [&](int k, int point) {
        if (tracing && (int)blockIdx.x == A.trace_wg && tid == 0 && k <= TRACE_ITERS)
            A.trace[k * TRACE_POINTS + point] = (long long)wall_clock64();
    };
    if constexpr (!ONE) {
        alive = grid_allreduce<2, 3>(A.ex, ztag, epoch++, nblocks, v0, red, flag,
            NoIdleWork());
        st.rr = v0[0];
        st.gnorm = sqrt(v0[1]);
        st.tol = A.fixed_tolerance < 0.0 ? st.gnorm * 0.01 : A.fixed_tolerance;
        stamp(0, 1);
        // ---- iterations (conjugate_gradient.h:123-198) ----
        double beta = 0.0;
        for (int k = 1; alive && k < A.max_iterations; ++k) {
            stamp(k, 0);
            // d_k = z + beta d_{k-1}: own node and halo, in LDS
            {
#pragma clang fp contract(off)
                double4_r *dt = reinterpret_cast<double4_r *>(dtile);
                if (mine) {
                    double4_r const old = dt[lcore];
                    dt[lcore] = (double4_r){ z[0] + beta * old.x, z[1] + beta * old.y,
                        z[2] + beta * old.z, z[3] + beta * old.w };
                    *reinterpret_cast<double4_r *>(yl + (size_t)li * 4)
                        = (double4_r){ 0.0, 0.0, 0.0, 0.0 };
                }
                if (halo_node >= 0) {
                    // the neighbour's z of the previous iteration: poll its eight
                    // granules until all carry this iteration's tag
                    double zh[4] = { 0.0, 0.0, 0.0, 0.0 };
                    (void)poll_node_granules(A.zg + (size_t)halo_node * 8,
                        ztag + (unsigned)k, A.ex, zh);
                    double4_r const old = dt[lhalo];
                    dt[lhalo] = (double4_r){ zh[0] + beta * old.x, zh[1] + beta * old.y,
                        zh[2] + beta * old.z, zh[3] + beta * old.w };
                }
            }
            lds_barrier();
            stamp(k, 1);
            double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
            double dself[4];
            tile_product(G, dtile, yl, fb, hd, hu, low, up, mine, acc, dself);
            stamp(k, 2);
            double dad[1] = { 0.0 };
            if (mine)
                dad[0] = dself[0] * acc[0] + dself[1] * acc[1]
                    + dself[2] * acc[2] + dself[3] * acc[3];
            stamp(k, 3);
            if (!(alive = grid_allreduce<1, 3>(A.ex, ztag, epoch++, nblocks, dad, red,
                      flag, NoIdleWork())))
                break;
            stamp(k, 4);
            double const alpha = st.rr / dad[0];
            // x += alpha d, r -= alpha Ad, z = P r: x, b and P in LDS; only the rim
            // publishes its z (as granules: nothing to drain before the all-reduce)
            double v3[3] = { 0.0, 0.0, 0.0 };
            if (mine) {
#pragma clang fp contract(off)
                double4_r *xp = reinterpret_cast<double4_r *>(xl + (size_t)li * 4);
                double4_r const xv = *xp;
                double4_r const bv = *reinterpret_cast<const double4_r *>(
                    bl + (size_t)li * 4);
                double const xn[4] = { xv.x + alpha * dself[0], xv.y + alpha * dself[1],
                    xv.z + alpha * dself[2], xv.w + alpha * dself[3] };
                double const bo[4] = { bv.x, bv.y, bv.z, bv.w };
                *xp = (double4_r){ xn[0], xn[1], xn[2], xn[3] };
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r[q] = r[q] - alpha * acc[q];
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    double4_r const p = *reinterpret_cast<const double4_r *>(
                        Pl + ((size_t)row * tile_nodes + li) * 4);
                    double zi = 0.0;
                    zi += p.x * r[0];
                    zi += p.y * r[1];
                    zi += p.z * r[2];
                    zi += p.w * r[3];
                    z[row] = zi;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (rim)
                        st_granules(A.zg + ((size_t)n * 4 + q) * 2,
                            ztag + (unsigned)k + 1u, z[q]);
                    v3[0] += r[q] * r[q];
                    v3[1] += xn[q] * (bo[q] + r[q]);
                    v3[2] += z[q] * r[q];
                }
            }
            stamp(k, 5);
            if (!(alive = grid_allreduce<3, 3>(A.ex, ztag, epoch++, nblocks, v3, red,
                      flag, NoIdleWork())))
                break;
            stamp(k, 6);
            // termination tests of iteration k (conjugate_gradient.h:136-198)
            double const new_rr = v3[0];
            double const Q1 = -1.0 * v3[1];
            int done = 0, info = SMVS_CG_MAX_ITERATIONS, iter_out = k + 1;
            if (new_rr < st.tol) {
                done = 1; info = SMVS_CG_CONVERGENCE; iter_out = k;
            } else {
                double const zeta = k * (Q1 - st.q0) / Q1;
                if (zeta < A.q_tolerance) {
                    done = 1; info = SMVS_CG_CONVERGENCE; iter_out = k;
                } else if (k + 1 >= A.max_iterations) {
                    done = 1;
                }
            }
            beta = v3[2] / st.rr;
            st.rr = v3[2];
            st.q0 = Q1;
            st.iter = iter_out;
            st.done = done;
            st.info = info;
            if (done)
                break;
        }
    } else {
        // ---- one exchange per iteration ----
        lds_barrier();          // d_1 (own and halo) in the tile
        stamp(0, 1);
        // who takes part in the exchanges: everybody, or (compacted) the tiles
        // with an active node -- their words have been out since they started
        LiveSet live;
        int const ngroups_all = (nblocks + RES_GROUP - 1) / RES_GROUP;
        live.leads = ngroups_all > 1 && tile % RES_GROUP == 0;
        live.bits = 3u;
        if (compact) {
            // (a map that did not arrive: the solve is given up, and since nobody
            // knows who the first live tile is, every workgroup says so)
            if (!gather_live_map())
                alive = false, report_all = true;
            int const member0 = tile - tile % RES_GROUP, j = tid & 15;
            auto group_live = [&](int g) {
                return ((livemap[(g * RES_GROUP) >> 6] >> ((g * RES_GROUP) & 63)) & 0xFFFFull)
                    != 0ull;
            };
            unsigned const mine16 = (unsigned)((livemap[member0 >> 6] >> (member0 & 63))
                & 0xFFFFull);
            live.leads = ngroups_all > 1 && mine16 != 0u
                && (int)__builtin_ctz(mine16) == tile - member0;
            live.bits = (member0 + j < nblocks && tile_live(member0 + j) ? 1u : 0u)
                | (j < ngroups_all && group_live(j) ? 2u : 0u);
            // the first live tile reports the result
            reporter = 0;
            for (int w = RES_MAX_BLOCKS / 64 - 1; w >= 0; --w)
                if (livemap[w] != 0ull)
                    reporter = 64 * w + (int)__builtin_ctzll(livemap[w]);
        }
        double xbr = 0.0;       // x.(b + r) of the current vectors
        double zr_part = 0.0;   // this node's z.r, formed where z is (end of the last iteration)
        __amdgpu_buffer_rsrc_t const zbuf = pair_buffer(A.zg, (size_t)A.num_nodes * 128);
        // byte offset of a node's four pairs in `zg`: by node id
        auto zg_at = [&](unsigned par, unsigned node) {
            return ((par * (unsigned)A.num_nodes + node) * 4u) * 16u;
        };
        auto skew = [&](int k, int point, int who) {
            if (tracing && k == TRACE_SKEW_ITER && tid == who)
                A.trace[TRACE_SKEW_BASE + 4 * blockIdx.x + point] = (long long)wall_clock64();
        };
        for (int k = 1; alive && k < A.max_iterations; ++k) {
            stamp(k, 0);
            skew(k, 0, 0);
            auto sweep_mark = [&](int point, long long value) {
                if (point >= 16) {
                    // per-wave stamps of one iteration
                    if (tracing && (int)blockIdx.x == A.trace_wg
                        && (tid & 63) == 0 && k == TRACE_SKEW_ITER)
                        A.trace[TRACE_WAVE_BASE + 8 * (tid >> 6) + point - 16]
                            = (long long)wall_clock64();
                    return;
                }
                if (tracing && (int)blockIdx.x == A.trace_wg && tid == 64
                    && k <= TRACE_ITERS)
                    A.trace[k * TRACE_POINTS + point] = value >= 0 ? value
                        : (long long)wall_clock64();
            };
            sweep_mark(16, -1);
            double acc[4] = { 0.0, 0.0, 0.0, 0.0 };   // q = H d
            double dself[4];
            tile_product(G, dtile, yl, fb, hd, hu, low, up, mine, acc, dself);
            stamp(k, 1);
            sweep_mark(17, -1);
            // the sums (z.q + w.r is taken as 2 w.r: P is symmetric, z.q = r.(P q))
            double v8[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
            // The rim's q for the neighbouring tiles' halo.
            unsigned const hpar = (unsigned)k & 1u;
            auto publish_rim = [&]() {
                if (rim) {
                    unsigned const t = ztag + (unsigned)k;
                    unsigned const at = zg_at(hpar, (unsigned)n);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        st_pair16(zbuf, at + (unsigned)q * 16u, t, acc[q]);
                }
            };
            // (as soon as q is known, in front of the sums although the exchange
            // waits for those: the write-through stores of a CU drain slowly,
            // ~380 of them in 3-5 us, and issuing them takes a wave up to 1 us --
            // behind the sums, the halo arrived 3 us later; here the issue
            // overlaps the arithmetic of the sums.  Also measured and dropped:
            // the rim staged in LDS and stored behind the sums by two waves as
            // whole 64-byte records, four lanes per node -- fewer, fuller
            // writes, but the halo arrives late: 9.9 instead of 9.1 us.)
            publish_rim();
            sweep_mark(19, -1);
            if (mine) {
#pragma clang fp contract(off)
                double Pn[4][4];
                load_symmetric(Pl, tile_nodes, li, Pn);
                double4_r const rv = *reinterpret_cast<const double4_r *>(
                    rl + (size_t)li * 4);
                double const r[4] = { rv.x, rv.y, rv.z, rv.w };
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    double wi = 0.0;
                    wi += Pn[row][0] * acc[0];
                    wi += Pn[row][1] * acc[1];
                    wi += Pn[row][2] * acc[2];
                    wi += Pn[row][3] * acc[3];
                    v8[3] += 2.0 * (wi * r[row]);
                    v8[4] += wi * acc[row];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v8[0] += dself[q] * acc[q];
                    v8[1] += r[q] * acc[q];
                    v8[2] += acc[q] * acc[q];
                    v8[5] += dself[q] * r[q];
                    v8[6] += r[q] * r[q];
                }
                v8[7] = zr_part;
            }
            sweep_mark(18, -1);
            stamp(k, 2);
            skew(k, 1, 0);
            // The sweeping waves run the two-level all-reduce; wave 0 meanwhile
            // collects the q of the halo nodes into LDS.
            // (Measured and dropped: four adjacent lanes reading the four pairs of
            // one node, a quarter of the requests per poll round -- the extra
            // registers cost more than the requests: 174 instead of 176 M
            // patch-steps/s.)
            auto fetch_halo = [&]() {
                if (nblocks > 1)
                    nap(A.wait_halo);
                for (int hs = tid; hs < ring; hs += 64) {
                    int const node = hnode[hs];
                    double qv[4] = { 0.0, 0.0, 0.0, 0.0 };
                    if (node >= 0)
                        (void)poll_node_pairs(zbuf, zg_at(hpar, (unsigned)node),
                            ztag + (unsigned)k, A.ex, qv, A.wait_poll);
                    *reinterpret_cast<double4_r *>(qhl + (size_t)hs * 4)
                        = (double4_r){ qv[0], qv[1], qv[2], qv[3] };
                }
            };
            auto other_waves = [&](int wave) {
                if (wave == 0) {
                    fetch_halo();
                    stamp(k, 7);
                    skew(k, 3, 0);
                }
            };
            alive = grid_allreduce_tree<8>(A.ex, ztag, epoch++, nblocks, live, v8, red, flag,
                other_waves, sweep_mark, A.wait_member, A.wait_poll);
            if (!alive)
                break;
            stamp(k, 3);
            skew(k, 2, 0);
            sweep_mark(22, -1);      // per wave: the totals are in, the update starts
            double const dq = v8[0], rq = v8[1], qq = v8[2], s1 = v8[3], wq = v8[4],
                dr = v8[5], rr = v8[6];
            // z.r of the current vectors, summed directly like r.r (d_1 = z_0:
            // in the first iteration it is d.r); only the values one step ahead
            // come from the recurrences, so no rounding error is carried from
            // iteration to iteration
            double const zr = k == 1 ? dr : v8[7];
            if (k == 1) {
                // r_0 = -g
                st.gnorm = sqrt(rr);
                st.tol = A.fixed_tolerance < 0.0 ? st.gnorm * 0.01 : A.fixed_tolerance;
            }
            double const alpha = zr / dq;
            double const new_rr = __builtin_fma(alpha,
                __builtin_fma(alpha, qq, -2.0 * rq), rr);
            double const new_zr = __builtin_fma(alpha,
                __builtin_fma(alpha, wq, -s1), zr);
            double const new_xbr = __builtin_fma(alpha,
                __builtin_fma(-alpha, dq, 2.0 * dr), xbr);
            // termination tests of iteration k (conjugate_gradient.h:136-198)
            double const Q1 = -1.0 * new_xbr;
            int done = 0, info = SMVS_CG_MAX_ITERATIONS, iter_out = k + 1;
            if (new_rr < st.tol) {
                done = 1; info = SMVS_CG_CONVERGENCE; iter_out = k;
            } else {
                double const zeta = k * (Q1 - st.q0) / Q1;
                if (zeta < A.q_tolerance) {
                    done = 1; info = SMVS_CG_CONVERGENCE; iter_out = k;
                } else if (k + 1 >= A.max_iterations) {
                    done = 1;
                }
            }
            double const beta = new_zr / zr;
            // x += alpha d, r -= alpha q, z = P r, d = z + beta d: own node, and
            // r, z, d of the halo with the same operations
            {
#pragma clang fp contract(off)
                double4_r *dt = reinterpret_cast<double4_r *>(dtile);
                if (mine) {
                    double4_r *xp = reinterpret_cast<double4_r *>(xl + (size_t)li * 4);
                    double4_r const xv = *xp;
                    *xp = (double4_r){ xv.x + alpha * dself[0], xv.y + alpha * dself[1],
                        xv.z + alpha * dself[2], xv.w + alpha * dself[3] };
                    if (!done) {
                        double4_r *rp = reinterpret_cast<double4_r *>(rl + (size_t)li * 4);
                        double4_r const rv = *rp;
                        double const r[4] = { rv.x - alpha * acc[0], rv.y - alpha * acc[1],
                            rv.z - alpha * acc[2], rv.w - alpha * acc[3] };
                        *rp = (double4_r){ r[0], r[1], r[2], r[3] };
                        double Pn[4][4];
                        load_symmetric(Pl, tile_nodes, li, Pn);
                        double zn[4];
#pragma unroll
                        for (int row = 0; row < 4; ++row) {
                            double zi = 0.0;
                            zi += Pn[row][0] * r[0];
                            zi += Pn[row][1] * r[1];
                            zi += Pn[row][2] * r[2];
                            zi += Pn[row][3] * r[3];
                            zn[row] = zi;
                        }
                        {
                            double t = 0.0;
                            t += zn[0] * r[0];
                            t += zn[1] * r[1];
                            t += zn[2] * r[2];
                            t += zn[3] * r[3];
                            zr_part = t;
                        }
                        dt[lcore] = (double4_r){ zn[0] + beta * dself[0],
                            zn[1] + beta * dself[1], zn[2] + beta * dself[2],
                            zn[3] + beta * dself[3] };
                        *reinterpret_cast<double4_r *>(yl + (size_t)li * 4)
                            = (double4_r){ 0.0, 0.0, 0.0, 0.0 };
                    }
                }
                if (!done && halo_node >= 0) {
                    double4_r *rp = reinterpret_cast<double4_r *>(rh + (size_t)tid * 4);
                    double4_r const rv = *rp;
                    double4_r const qv = *reinterpret_cast<const double4_r *>(
                        qhl + (size_t)tid * 4);
                    double const rn[4] = { rv.x - alpha * qv.x, rv.y - alpha * qv.y,
                        rv.z - alpha * qv.z, rv.w - alpha * qv.w };
                    *rp = (double4_r){ rn[0], rn[1], rn[2], rn[3] };
                    double zh[4];
#pragma unroll
                    for (int row = 0; row < 4; ++row) {
                        double4_r const p = *reinterpret_cast<const double4_r *>(
                            Ph + (size_t)tid * 16 + row * 4);
                        double zi = 0.0;
                        zi += p.x * rn[0];
                        zi += p.y * rn[1];
                        zi += p.z * rn[2];
                        zi += p.w * rn[3];
                        zh[row] = zi;
                    }
                    double4_r const old = dt[lhalo];
                    dt[lhalo] = (double4_r){ zh[0] + beta * old.x, zh[1] + beta * old.y,
                        zh[2] + beta * old.z, zh[3] + beta * old.w };
                }
            }
            stamp(k, 4);
            sweep_mark(23, -1);      // per wave: own node (and, lanes < ring, halo node) updated
            xbr = new_xbr;
            st.rr = new_zr;
            st.q0 = Q1;
            st.iter = iter_out;
            st.done = done;
            st.info = info;
            if (done)
                break;
            lds_barrier();      // d_{k+1} in the tile
        }
    }

    // the solution (and b, which the streaming solver also leaves) go to HBM
    if (mine) {
        *reinterpret_cast<double4_r *>(A.x + (size_t)n * 4)
            = *reinterpret_cast<const double4_r *>(xl + (size_t)li * 4);
        if (!ONE)
            *reinterpret_cast<double4_r *>(A.b + (size_t)n * 4)
                = *reinterpret_cast<const double4_r *>(bl + (size_t)li * 4);
    }
    if ((tile == reporter || report_all) && tid == 0) {
        // (a halo wait of any workgroup that gave up raised the exchange's flag)
        bool const late_timeout = __hip_atomic_load(&A.ex->timeout, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT) != 0u;
        int const failed = alive && !late_timeout && !(A.pipelined & 2) ? 0 : 1;   // (bit 1: test hook)
        if (!st.done && !failed) {
            // max_iterations <= 1 never reaches here (handled by the host)
            st.done = 1;
        }
        A.state[0] = st;
        A.status[I_DONE] = failed ? 0 : 1;
        A.status[I_INFO] = st.info;
        A.status[I_ITER] = st.iter;
        if (failed && A.pipelined)
            A.status[I_STEP_ABORT] = ABORT_SOLVER;   // the enqueued steps do nothing
        __hip_atomic_store(A.progress + 2, st.info, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.progress + 3, st.iter, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.progress + 4, failed, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.progress + 1, A.solve_tag | 1, __ATOMIC_RELEASE,
            __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static size_t
resident_lds_bytes(int tw, int th, bool one)
{
    return resident_lds_layout(tw, th, one).total * sizeof(double);
}

// Tile shape: tw * th <= 512 nodes, at most max_tiles tiles, fits the LDS of
// the solver variant, smallest rim.
static bool
choose_tiling(int stride, int rows, int max_tiles, bool one, int *tw_out, int *th_out)
{
    long best = -1;
    for (int tw = 4; tw <= 128 && tw <= RES_THREADS; ++tw) {
        int const th = RES_THREADS / tw;
        if (th < 2)
            continue;
        for (int t2 = th; t2 >= 2 && t2 >= th - 8; --t2) {
            long const tiles = (long)((stride + tw - 1) / tw)
                * ((rows + t2 - 1) / t2);
            if (tiles > max_tiles)
                continue;
            // (long thin tiles have a long rim: the one-exchange solver keeps
            // r, P and q of the halo in LDS)
            if (resident_lds_bytes(tw, t2, one) > (size_t)160 * 1024)
                continue;
            // prefer few idle threads, then a short rim
            long const waste = tiles * (long)(tw * t2) - (long)stride * rows;
            long const score = waste * 4 + tiles * (tw + t2);
            if (best < 0 || score < best) {
                best = score;
                *tw_out = tw;
                *th_out = t2;
            }
        }
    }
    return best >= 0;
}

// Which of the two resident solvers a context runs (smvs_ctx_set_solver), and
// on which tiles.  The one-exchange recurrence exists to save a grid-wide
// exchange per iteration; on a grid of ONE tile nothing is exchanged, so AUTO
// runs the reference's operation order there (conjugate_gradient.h:121-198:
// d.Ad, then r.r, z.r, x.(b + r) of the updated vectors summed directly) --
// the tiny ill-conditioned systems of the fuzz sweep live on such grids.
// SMVS_REF_ORDER_TILES=n widens that to grids of <= n tiles (measurements).
struct ResidentPlan {
    int tw, th;
    bool one;
    int blocks;     // workgroups of the launch = tiles, row-major
};

static void
finish_plan(const smvs_ctx *ctx, int max_tiles, ResidentPlan *plan)
{
    int const stride = ctx->node_stride, rows = ctx->num_nodes / stride;
    int const tiles_x = (stride + plan->tw - 1) / plan->tw;
    int const tiles_y = (rows + plan->th - 1) / plan->th;
    plan->blocks = tiles_x * tiles_y;
}

static bool
compute_resident_plan(const smvs_ctx *ctx, ResidentPlan *plan);

// (once per grid: a Newton step asks for the plan, a loop three more times)
static bool
resident_plan(const smvs_ctx *ctx, ResidentPlan *plan)
{
    smvs_ctx::ResidentPlanMemo &m = ctx->res_plan;
    if (m.stride != ctx->node_stride || m.nodes != ctx->num_nodes
        || m.solver_mode != (int)ctx->solver_mode || m.cus != ctx->resident_cus) {
        ResidentPlan p = {};
        m.ok = compute_resident_plan(ctx, &p);
        m.stride = ctx->node_stride;
        m.nodes = ctx->num_nodes;
        m.solver_mode = (int)ctx->solver_mode;
        m.cus = ctx->resident_cus;
        m.tw = p.tw; m.th = p.th; m.one = p.one ? 1 : 0; m.blocks = p.blocks;
    }
    plan->tw = m.tw; plan->th = m.th; plan->one = m.one != 0; plan->blocks = m.blocks;
    return m.ok;
}

static bool
compute_resident_plan(const smvs_ctx *ctx, ResidentPlan *plan)
{
    int const stride = ctx->node_stride;
    int const rows = ctx->num_nodes / stride;
    int const max_tiles = ctx->resident_cus < RES_MAX_BLOCKS
        ? ctx->resident_cus : RES_MAX_BLOCKS;
    static int const ref_order_tiles = [] {
        const char *e = std::getenv("SMVS_REF_ORDER_TILES");
        int const n = e != nullptr ? std::atoi(e) : 1;
        return n < 0 ? 0 : n;
    }();
    int tw = 0, th = 0;
    bool const have_ref = choose_tiling(stride, rows, max_tiles, false, &tw, &th);
    if (have_ref) {
        long const tiles = (long)((stride + tw - 1) / tw) * ((rows + th - 1) / th);
        if (ctx->solver_mode == SMVS_SOLVER_RESIDENT_REF || tiles <= ref_order_tiles) {
            plan->tw = tw;
            plan->th = th;
            plan->one = false;
            finish_plan(ctx, max_tiles, plan);
            return true;
        }
    } else if (ctx->solver_mode == SMVS_SOLVER_RESIDENT_REF) {
        return false;
    }
    if (!choose_tiling(stride, rows, max_tiles, true, &tw, &th))
        return false;
    plan->tw = tw;
    plan->th = th;
    plan->one = true;
    finish_plan(ctx, max_tiles, plan);
    return true;
}

// Does the resident solver take this system?  (grid fits the chip's CUs and
// LDS, not disabled by SMVS_CG_RESIDENT=0 or by an earlier failure)
bool
cg_resident_applies(smvs_ctx *ctx, int max_iterations)
{
    if (ctx->resident_disabled || max_iterations <= 1 || !ctx->has_surface
        || ctx->solver_mode == SMVS_SOLVER_STREAMING)
        return false;
    // the exchange tags carry the epoch (<= 2 per iteration + 1) in their low
    // 16 bits beside the solve id: longer solves take the streaming kernels
    if (2 * max_iterations + 1 > 0xFFFF)
        return false;
    static int const env_off = [] {
        const char *e = std::getenv("SMVS_CG_RESIDENT");
        return e != nullptr && e[0] == '0' ? 1 : 0;
    }();
    if (env_off)
        return false;
    if (ctx->resident_cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, physical_device(ctx->device)) != hipSuccess)
            return false;
        ctx->resident_cus = prop.multiProcessorCount;
    }
    ResidentPlan plan;
    if (!resident_plan(ctx, &plan))
        return false;
    if ((size_t)ctx->num_nodes * 5 * 16 >= (size_t)1 << 32)
        return false;
    return true;
}

static DeviceTileBudget g_resident_budget[16];

void
DeviceTileBudget::bind(int device)
{
    // (caller holds the mutex)
    if (bound)
        return;
    bound = true;
    hipDeviceProp_t prop;
    capacity = RES_MAX_BLOCKS;
    // (`device` is the PHYSICAL device here: logical devices mapped onto one
    // GPU share its CUs, cg_resident_budget)
    if (hipGetDeviceProperties(&prop, device) == hipSuccess
        && prop.multiProcessorCount < capacity)
        capacity = prop.multiProcessorCount;
    // the same GPU may have different indices in different processes
    // (HIP_VISIBLE_DEVICES): name the file after the PCI bus id
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess)
        std::snprintf(bus, sizeof(bus), "index%d", device);
    for (char *c = bus; *c != 0; ++c)
        if (*c == ':' || *c == '/')
            *c = '_';
    // SMVS_LOCK_DIR, else the user's runtime directory, else /tmp.  The file
    // is never followed through a symlink; when another user created it
    // (O_RDWR refused) a read-only descriptor serves flock() just as well.
    const char *dir = std::getenv("SMVS_LOCK_DIR");
    if (dir == nullptr || dir[0] == 0)
        dir = std::getenv("XDG_RUNTIME_DIR");
    if (dir == nullptr || dir[0] == 0)
        dir = "/tmp";
    std::string const path = std::string(dir) + "/smvs_hip_barrier_" + bus + ".lock";
    fd = ::open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
    if (fd < 0)
        fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC | O_NOFOLLOW);
    if (fd < 0)
        std::fprintf(stderr, "[smvs_hip] no lock file %s (%s): the resident solver is "
            "serialised inside this process only; a second process on the same GPU may "
            "push it into the streaming kernels\n", path.c_str(), std::strerror(errno));
}

// The advisory file lock that makes PROCESSES sharing a GPU take turns with
// their barrier kernels.  Taken by the head of this process's line WITHOUT the
// budget's mutex held (it may sleep for as long as another process's loops
// run); kept while loops of this process follow each other, but for at most
// FILE_HOLD at a stretch: after that the next acquirer lets the loops in flight
// drain, returns the lock -- a process waiting on it gets its turn -- and takes
// it again (a drain every 100 ms costs the single process ~1 %).  A lock somebody holds for 20 s is not one of ours (or is stuck):
// go on without it for a while -- the worst case is a resident solve that times
// out into the streaming kernels, not a hang.
static constexpr auto FILE_HOLD = std::chrono::milliseconds(100);
static constexpr auto FILE_GIVE_UP = std::chrono::seconds(20);
static constexpr auto FILE_RETRY_AFTER = std::chrono::seconds(60);

bool
DeviceTileBudget::take_file_lock(void)
{
    // (no mutex held: the caller is the only thread of this process in here)
    auto const t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        if (::flock(fd, LOCK_EX | LOCK_NB) == 0)
            return true;
        if (errno != EWOULDBLOCK && errno != EINTR)
            return false;
        if (std::chrono::steady_clock::now() - t0 > FILE_GIVE_UP) {
            static std::atomic<bool> warned{false};
            if (!warned.exchange(true))
                std::fprintf(stderr, "[smvs_hip] barrier lock file busy for 20 s: "
                    "continuing without it\n");
            return false;
        }
        if (spin < 64)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

void
DeviceTileBudget::unlock_file(void)
{
    if (fd >= 0 && file_locked)
        (void)::flock(fd, LOCK_UN);
    file_locked = false;
}

void
DeviceTileBudget::acquire(int device, int tiles)
{
    std::unique_lock<std::mutex> guard(mutex);
    bind(device);
    if (tiles > capacity)
        tiles = capacity;
    unsigned long long const ticket = next_ticket++;
    // in arrival order; the head of the line waits for its tiles, the others
    // wait for the head
    turn.wait(guard, [&] { return serving == ticket && used + tiles <= capacity; });
    auto const now = std::chrono::steady_clock::now();
    bool handed_back = false;
    if (fd >= 0 && now >= no_file_until) {
        if (file_locked && now - file_since > FILE_HOLD) {
            // this process has had the GPU's barrier kernels to itself long
            // enough: let its loops in flight end (nobody passes the head of the
            // line meanwhile) and hand the lock back before taking it again
            turn.wait(guard, [&] { return holders == 0; });
            unlock_file();
            handed_back = true;
        }
        if (!file_locked) {
            guard.unlock();
            // (flock is not FIFO and a waiting process polls every 50 us: taking
            // the lock again at once would win it back nearly every time, and
            // the waiter would starve into its 20 s give-up.  Four of its poll
            // periods are its turn.)
            if (handed_back)
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            bool const got = take_file_lock();
            guard.lock();
            file_locked = got;
            file_since = std::chrono::steady_clock::now();
            if (!got)
                no_file_until = file_since + FILE_RETRY_AFTER;
        }
    }
    used += tiles;
    serving += 1;
    holders += 1;
    guard.unlock();
    turn.notify_all();   // (the next in line may fit beside this one)
}

void
DeviceTileBudget::release(int tiles)
{
    {
        std::lock_guard<std::mutex> guard(mutex);
        if (tiles > capacity)
            tiles = capacity;
        used -= tiles;
        if (--holders == 0)
            unlock_file();      // (non-blocking)
    }
    turn.notify_all();
}

DeviceTileBudget &
cg_resident_budget(int device)
{
    return g_resident_budget[physical_device(device) & 15];
}

int
cg_resident_tiles(smvs_ctx *ctx)
{
    ResidentPlan plan;
    if (ctx->resident_cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, physical_device(ctx->device)) != hipSuccess)
            return 0;
        ctx->resident_cus = prop.multiProcessorCount;
    }
    if (!resident_plan(ctx, &plan))
        return 0;
    return plan.blocks;
}

// Launches the solver (the caller holds its tiles of the budget and has checked
// cg_resident_applies).  pipelined: the Newton loop's launch-ahead mode -- the
// kernel leaves at once when status[I_STOP] / [I_STEP_ABORT] is set, raises
// I_STEP_ABORT itself when it gives up, and nobody waits on the progress words.
// *solve_tag_out: the tag the progress words will carry.
static int
resident_enqueue(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, bool fused, bool pipelined, long long *trace_dev,
    int *solve_tag_out, int *num_tiles_out, bool test_give_up = false)
{
    int const stride = ctx->node_stride;
    int const rows = ctx->num_nodes / stride;
    ResidentPlan plan;
    if (!resident_plan(ctx, &plan)) {
        set_error("resident_enqueue: the grid does not fit the resident solver");
        return SMVS_ERR_STATE;
    }
    int const tw = plan.tw, th = plan.th;
    int const tiles_x = (stride + tw - 1) / tw;
    int const num_tiles = plan.blocks;   // workgroups of the launch
    bool const one = plan.one;
    size_t const lds_bytes = resident_lds_bytes(tw, th, one);

    int rc;
    if (ctx->res_work == nullptr) {
        if ((rc = device_alloc(&ctx->res_work,
                 sizeof(ResExchange) / sizeof(double) + 8)) != SMVS_OK)
            return rc;
        SMVS_HIP_CHECK(hipMemsetAsync(ctx->res_work, 0, sizeof(ResExchange),
            ctx->stream));
    }
    // (the solver takes grids of at most RES_MAX_BLOCKS tiles of RES_THREADS nodes)
    size_t const zx_nodes = std::min(ctx->cap_nodes, (size_t)RES_MAX_BLOCKS * RES_THREADS);
    if (ctx->res_zx_cap < (size_t)ctx->num_nodes) {
        // [2][zx_nodes][4][2] words of `zg`
        if ((rc = device_alloc(&ctx->res_zx, zx_nodes * 16)) != SMVS_OK) {
            ctx->res_zx_cap = 0;
            return rc;
        }
        SMVS_HIP_CHECK(hipMemsetAsync(ctx->res_zx, 0, zx_nodes * 16 * sizeof(double),
            ctx->stream));
        ctx->res_zx_cap = zx_nodes;
    }
    {
        const void *kernels[5] = {
            reinterpret_cast<const void *>(cg_resident_kernel<false, false, false>),
            reinterpret_cast<const void *>(cg_resident_kernel<true, false, false>),
            reinterpret_cast<const void *>(cg_resident_kernel<false, true, false>),
            reinterpret_cast<const void *>(cg_resident_kernel<true, true, false>),
            reinterpret_cast<const void *>(cg_resident_kernel<true, true, true>) };
        for (const void *k : kernels)
            if ((rc = allow_dynamic_lds(ctx->device, k, lds_bytes)) != SMVS_OK)
                return rc;
    }

    ResArgs A;
    A.H9 = ctx->H9;
    A.Pinv = ctx->Pinv;
    A.g = ctx->g;
    A.x = ctx->x;
    A.b = ctx->b;
    A.zg = reinterpret_cast<unsigned long long *>(ctx->res_zx);
    A.ex = reinterpret_cast<ResExchange *>(ctx->res_work);
    A.state = reinterpret_cast<ResState *>(ctx->cg_state);
    A.status = ctx->status;
    A.progress = ctx->cg_progress;
    ctx->cg_solve_id = (ctx->cg_solve_id + 1) & 0x7FFF;
    if (ctx->cg_solve_id == 0) {
        // The 15-bit solve id starts over: a granule that no solve has
        // overwritten since the id was last used (a slot of an epoch only a
        // longer solve reaches) would carry a matching tag.  Clear both
        // exchange areas, once per 32,767 solves.
        ctx->cg_solve_id = 1;
        SMVS_HIP_CHECK(hipMemsetAsync(ctx->res_work, 0, sizeof(ResExchange),
            ctx->stream));
        SMVS_HIP_CHECK(hipMemsetAsync(ctx->res_zx, 0,
            ctx->res_zx_cap * 16 * sizeof(double), ctx->stream));
    }
    A.solve_tag = ctx->cg_solve_id << 16;
    A.num_nodes = ctx->num_nodes;
    A.stride = stride;
    A.rows = rows;
    A.tw = tw;
    A.th = th;
    A.tiles_x = tiles_x;
    A.num_tiles = num_tiles;
    A.max_iterations = max_iterations;
    A.q_tolerance = q_tolerance;
    A.fixed_tolerance = error_tolerance;
    A.Hp = ctx->Hp;
    A.gp = ctx->gp;
    A.layout = patch_layout(ctx);
    A.patch_valid = ctx->patch_valid;
    A.active = ctx->active;
    A.active_next = ctx->active_next;
    A.scalars = ctx->scalars;
    A.npx = ctx->npx;
    A.npy = ctx->npy;
    A.zeros = ctx->zero_block;
    {
        // polling cadence (the measured defaults; SMVS_CG_WAIT="halo,member,poll"
        // overrides them for experiments): the exchange gains 1.5 us per iteration
        // when the halo's polls start 1 us late (they compete with the
        // write-through stores of the rim)
        struct Knobs { int halo = -1, member = 3, poll = 0; };
        static Knobs const knobs = [] {
            Knobs k;
            if (const char *e = std::getenv("SMVS_CG_WAIT"))
                (void)std::sscanf(e, "%d,%d,%d", &k.halo, &k.member, &k.poll);
            return k;
        }();
        // (the full grid only: at 64 tiles the wait changes nothing, 6.95 against
        // 6.86 us per iteration, at 16 tiles it costs 0.5 us)
        A.wait_halo = knobs.halo >= 0 ? knobs.halo : num_tiles <= 64 ? 0 : 4;
        A.wait_member = knobs.member;
        A.wait_poll = knobs.poll;
    }
    {
        static int const compact = [] {
            const char *e = std::getenv("SMVS_CG_COMPACT");
            return e != nullptr && e[0] == '0' ? 0 : 1;
        }();
        A.compact = compact;
    }
    A.trace = trace_dev;
    {
        static int const wg = [] {
            const char *e = std::getenv("SMVS_CG_TRACE_WG");
            return e != nullptr ? std::atoi(e) : 0;
        }();
        A.trace_wg = wg;
    }
    A.pipelined = pipelined ? (test_give_up ? 3 : 1) : 0;
    *solve_tag_out = A.solve_tag;
    *num_tiles_out = num_tiles;
    {
        ScopedKernelTimer timer(ctx, SMVS_K_CG_RESIDENT);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(num_tiles), dim3(RES_THREADS), lds_bytes,
                ctx->stream, A);
        };
        // (stamps: the fused one-exchange kernel only, what tools/cg_trace.py runs)
        if (fused && one && trace_dev != nullptr)
            launch(cg_resident_kernel<true, true, true>);
        else if (fused && one)
            launch(cg_resident_kernel<true, true, false>);
        else if (fused)
            launch(cg_resident_kernel<true, false, false>);
        else if (one)
            launch(cg_resident_kernel<false, true, false>);
        else
            launch(cg_resident_kernel<false, false, false>);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    return SMVS_OK;
}

// A resident solve gave up (the caller has waited for the kernel): its
// workgroups were not all resident.  Never try again on this context, the
// streaming kernels take over.
int
cg_resident_gave_up(smvs_ctx *ctx)
{
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->res_work, 0, sizeof(ResExchange), ctx->stream));
    ctx->resident_disabled = true;
    return SMVS_OK;
}

int
cg_resident_enqueue(smvs_ctx *ctx, int max_iterations, double q_tolerance,
    bool test_give_up)
{
    int tag = 0, tiles = 0;
    return resident_enqueue(ctx, max_iterations, -1.0, q_tolerance, true, true,
        nullptr, &tag, &tiles, test_give_up);
}

// Returns SMVS_OK with *ran = false when the resident solver does not apply
// (grid too large for the chip, disabled, or it failed to synchronise) -- the
// caller then runs the streaming kernels.
int
cg_resident_solve(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info, bool *ran, bool fused)
{
    *ran = false;
    if (!cg_resident_applies(ctx, max_iterations))
        return SMVS_OK;
    // debug aid (tools/cg_trace.py): cycle stamps of workgroup 0
    static const char *trace_path = std::getenv("SMVS_CG_TRACE");
    long long *trace_dev = nullptr;
    size_t const trace_n = (size_t)TRACE_TOTAL;
    if (trace_path != nullptr) {
        SMVS_HIP_CHECK(hipMalloc((void **)&trace_dev, trace_n * sizeof(long long)));
        SMVS_HIP_CHECK(hipMemsetAsync(trace_dev, 0, trace_n * sizeof(long long),
            ctx->stream));
    }
    ScopedTileBudget guard(ctx->device, cg_resident_tiles(ctx));
    int solve_tag = 0, num_tiles = 0;
    int const rc = resident_enqueue(ctx, max_iterations, error_tolerance,
        q_tolerance, fused, false, trace_dev, &solve_tag, &num_tiles);
    if (rc != SMVS_OK)
        return rc;

    volatile int *progress = ctx->cg_progress;
    auto const t_start = std::chrono::steady_clock::now();
    long spins = 0;
    while (__atomic_load_n(&progress[1], __ATOMIC_ACQUIRE) != (solve_tag | 1)) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFFF) == 0) {
            hipError_t const q = hipStreamQuery(ctx->stream);
            if (q != hipSuccess && q != hipErrorNotReady)
                SMVS_HIP_CHECK(q);
            if (q == hipSuccess
                && __atomic_load_n(&progress[1], __ATOMIC_ACQUIRE)
                    != (solve_tag | 1)) {
                set_error("cg_resident_solve: kernel ended without a result");
                return SMVS_ERR_STATE;
            }
            if (std::chrono::steady_clock::now() - t_start
                > std::chrono::seconds(60)) {
                set_error("cg_resident_solve: timed out waiting for the device");
                return SMVS_ERR_STATE;
            }
        }
    }
    if (trace_dev != nullptr) {
        std::vector<long long> tr(trace_n);
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        SMVS_HIP_CHECK(hipMemcpy(tr.data(), trace_dev, trace_n * sizeof(long long),
            hipMemcpyDeviceToHost));
        (void)hipFree(trace_dev);
        if (FILE *f = std::fopen(trace_path, "a")) {
            std::fprintf(f, "solve nodes=%d tiles=%d its=%d exchanges=%d\n",
                ctx->num_nodes, num_tiles, progress[3],
                [&] { ResidentPlan p; return resident_plan(ctx, &p) && p.one; }() ? 1 : 2);
            for (int k = 0; k <= TRACE_ITERS; ++k) {
                for (int q = 0; q < TRACE_POINTS; ++q)
                    std::fprintf(f, "%lld ", tr[(size_t)k * TRACE_POINTS + q]);
                std::fprintf(f, "\n");
            }
            for (int b = 0; b < num_tiles; ++b)
                std::fprintf(f, "block %d %lld %lld %lld %lld\n", b,
                    tr[TRACE_BLOCK_BASE + 4 * b], tr[TRACE_BLOCK_BASE + 4 * b + 1],
                    tr[TRACE_BLOCK_BASE + 4 * b + 2], tr[TRACE_BLOCK_BASE + 4 * b + 3]);
            for (int w = 0; w < 8; ++w) {
                std::fprintf(f, "wave %d", w);
                for (int q = 0; q < 8; ++q)
                    std::fprintf(f, " %lld", tr[TRACE_WAVE_BASE + 8 * w + q]);
                std::fprintf(f, "\n");
            }
            for (int b = 0; b < num_tiles; ++b)
                std::fprintf(f, "skew %d %lld %lld %lld %lld\n", b,
                    tr[TRACE_SKEW_BASE + 4 * b], tr[TRACE_SKEW_BASE + 4 * b + 1],
                    tr[TRACE_SKEW_BASE + 4 * b + 2], tr[TRACE_SKEW_BASE + 4 * b + 3]);
            std::fclose(f);
        }
    }
    // (every workgroup has passed its last barrier when the result appears:
    // the kernel is draining and cannot block a barrier kernel started now)
    if (progress[4] != 0)
        return cg_resident_gave_up(ctx);
    ctx->last_cg_iterations = progress[3];
    if (num_iterations != nullptr)
        *num_iterations = progress[3];
    if (info != nullptr)
        *info = progress[2];
    *ran = true;
    return SMVS_OK;
}

} // namespace smvs_hip
