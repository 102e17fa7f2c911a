// Per-node / per-patch arithmetic of the grid surgery of smvs::Surface
// (lib/surface.cc), shared by the C++ host mirror (g++, host/surface.cc) and
// the device kernels (hipcc, csrc/surface.hip): one source for both sides, so
// the CPU tests that compare the host mirror with the oracle also pin what the
// kernels compute per element.  Every function is a GATHER for one output
// element -- the reference's loops scatter (a patch writes its five split
// nodes, a later patch overwrites an earlier one); the gather forms below pick
// the same winner, which is what makes one thread per output possible.
// Plain IEEE operations in source order, no contraction.
#ifndef SMVS_SURFACE_MATH_H
#define SMVS_SURFACE_MATH_H

#include <cstddef>
#include <cstdint>

#include "topo_math.h"

namespace smvs_surf {

// ---------------------------------------------------------------- geometry
// Surface::create, lib/surface.cc:28-37
struct Grid
{
    int width, height;      // image
    int scale, ps;          // patch size = 2^scale
    int npx, npy;           // patches
    int start_x, start_y;   // pixel of node (0, 0)
};

SMVS_HD Grid
grid_for_scale(int width, int height, int scale)
{
    Grid g;
    g.width = width;
    g.height = height;
    g.scale = scale;
    g.ps = 1 << scale;
    g.npx = (width - 2) / g.ps - 1;
    g.npy = (height - 2) / g.ps - 1;
    g.start_x = (width - g.npx * g.ps) / 2;
    g.start_y = (height - g.npy * g.ps) / 2;
    return g;
}

// Surface::subdivide_patches, lib/surface.cc:983-1012: the grid one scale
// finer.  off_x / off_y = 1 when the finer grid gains a column / row of
// patches on either side (the old node (i, j) becomes (2 i + off_x, 2 j + off_y)).
SMVS_HD Grid
grid_subdivided(Grid const& old, int* off_x, int* off_y)
{
    Grid g = old;
    g.scale = old.scale - 1;
    g.ps = 1 << g.scale;
    int new_npx = (old.width - 2) / g.ps;
    int new_npy = (old.height - 2) / g.ps;
    *off_x = 0;
    *off_y = 0;
    if (new_npx - old.npx * 2 >= 2) {
        new_npx = old.npx * 2 + 2;
        g.start_x = (old.width - new_npx * g.ps) / 2;
        *off_x = 1;
    } else
        new_npx = old.npx * 2;
    if (new_npy - old.npy * 2 >= 2) {
        new_npy = old.npy * 2 + 2;
        g.start_y = (old.height - new_npy * g.ps) / 2;
        *off_y = 1;
    } else
        new_npy = old.npy * 2;
    g.npx = new_npx;
    g.npy = new_npy;
    return g;
}

// ------------------------------------------------------------- subdivision
// The node (X, Y) of the subdivided grid, lib/surface.cc:1014-1100.
//  * both old-grid coordinates even: the old node, derivatives rescaled to the
//    new patch size (:1082-1098);
//  * otherwise one of the five split points of a patch (:1024-1080): the edge
//    midpoints (parameter 1/2 on an edge) and the centre.  An edge midpoint
//    belongs to two patches; the reference loops over the patches in
//    ascending id and the later one overwrites, i.e. the patch BELOW a
//    horizontal edge / RIGHT of a vertical edge wins when it exists.
// old_*: the surface before the subdivision.  Returns false (node stays
// null) when nothing defines it; out = f, dx, dy, dxy.
SMVS_HD bool
subdivide_node(int old_npx, int old_npy, int off_x, int off_y,
    double const* old_nodes, uint8_t const* old_node_valid,
    uint8_t const* old_patch_valid, int X, int Y, double* out)
{
    SMVS_NO_CONTRACT
    int const u = X - off_x, v = Y - off_y;
    if (u < 0 || v < 0 || u > 2 * old_npx || v > 2 * old_npy)
        return false;
    int const old_stride = old_npx + 1;
    if ((u & 1) == 0 && (v & 1) == 0) {
        std::size_t const i = (std::size_t)(v >> 1) * old_stride + (u >> 1);
        if (!old_node_valid[i])
            return false;
        out[0] = old_nodes[4 * i + 0];
        out[1] = old_nodes[4 * i + 1] / 2;
        out[2] = old_nodes[4 * i + 2] / 2;
        out[3] = old_nodes[4 * i + 3] / 4;
        return true;
    }
    // candidate patches in DESCENDING id (the last writer wins) and the
    // parameter of the split point inside each, in halves
    int cand_px[2], cand_py[2], cand_iu[2], cand_iv[2];
    int n = 0;
    if ((u & 1) == 1 && (v & 1) == 1) {
        cand_px[n] = u >> 1; cand_py[n] = v >> 1; cand_iu[n] = 1; cand_iv[n] = 1; n += 1;
    } else if ((u & 1) == 1) {
        // midpoint of a horizontal edge: top edge of the patch below,
        // bottom edge of the patch above
        cand_px[n] = u >> 1; cand_py[n] = v >> 1; cand_iu[n] = 1; cand_iv[n] = 0; n += 1;
        cand_px[n] = u >> 1; cand_py[n] = (v >> 1) - 1; cand_iu[n] = 1; cand_iv[n] = 2; n += 1;
    } else {
        // midpoint of a vertical edge: left edge of the patch to the right,
        // right edge of the patch to the left
        cand_px[n] = u >> 1; cand_py[n] = v >> 1; cand_iu[n] = 0; cand_iv[n] = 1; n += 1;
        cand_px[n] = (u >> 1) - 1; cand_py[n] = v >> 1; cand_iu[n] = 2; cand_iv[n] = 1; n += 1;
    }
    for (int c = 0; c < n; ++c) {
        int const px = cand_px[c], py = cand_py[c];
        if (px < 0 || py < 0 || px >= old_npx || py >= old_npy)
            continue;
        if (!old_patch_valid[(std::size_t)py * old_npx + px])
            continue;
        std::size_t const n00 = (std::size_t)py * old_stride + px;
        std::size_t const ids[4] = { n00, n00 + 1, n00 + old_stride,
            n00 + old_stride + 1 };
        double n16[16];
        for (int k = 0; k < 4; ++k)
            for (int q = 0; q < 4; ++q)
                n16[4 * k + q] = old_nodes[4 * ids[k] + q];
        double const x = 0.5 * cand_iu[c], y = 0.5 * cand_iv[c];
        // (surface.cc:1036-1079: value and derivatives of the coarse patch,
        // derivatives in units of the new patch size)
        out[0] = smvs_topo::patch_eval(n16, x, y, 0, 0);
        out[1] = smvs_topo::patch_eval(n16, x, y, 1, 0) / 2.0;
        out[2] = smvs_topo::patch_eval(n16, x, y, 0, 1) / 2.0;
        out[3] = smvs_topo::patch_eval(n16, x, y, 1, 1) / 4.0;
        return true;
    }
    return false;
}

// ------------------------------------------------------------------ expand
// One node of one round of Surface::expand, lib/surface.cc:482-628: the
// candidates extrapolated from complete triples of neighbours, in the
// reference's order; a later candidate replaces an earlier one only when it
// is more than 1/0.9 deeper (check_swap_nodes, :472-480).  nodes / node_valid:
// the surface at the START of the round (the round's proposals are applied
// after every node has been visited).  proposal / proposed: the node's state
// from the previous round, updated in place.
SMVS_HD void
expand_node(int npx, int npy, double const* nodes, uint8_t const* node_valid,
    int idx, int idy, double* proposal, uint8_t* proposed)
{
    SMVS_NO_CONTRACT
    int const stride = npx + 1;
    // neighbours 0..7: (-1,-1) (0,-1) (1,-1) (-1,0) (1,0) (-1,1) (0,1) (1,1)
    int const off_x[8] = { -1, 0, 1, -1, 1, -1, 0, 1 };
    int const off_y[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
    bool have[8];
    double f[8], dx[8], dy[8];
    for (int k = 0; k < 8; ++k) {
        int const nx = idx + off_x[k], ny = idy + off_y[k];
        have[k] = nx >= 0 && ny >= 0 && nx <= npx && ny <= npy
            && node_valid[(std::size_t)ny * stride + nx] != 0;
        f[k] = dx[k] = dy[k] = 0.0;
        if (have[k]) {
            double const* nd = nodes + 4 * ((std::size_t)ny * stride + nx);
            f[k] = nd[0];
            dx[k] = nd[1];
            dy[k] = nd[2];
        }
    }
    // rule: the three neighbours it needs, then its terms as (neighbour,
    // axis 1 = dx / 2 = dy, sign)
    int const need[8][3] = { { 0, 1, 3 }, { 1, 2, 4 }, { 3, 5, 6 }, { 4, 6, 7 },
        { 0, 1, 2 }, { 0, 3, 5 }, { 5, 6, 7 }, { 2, 4, 7 } };
    int const nterms[8] = { 2, 2, 2, 2, 3, 3, 3, 3 };
    int const term_nb[8][3] = { { 3, 1, 0 }, { 4, 1, 0 }, { 3, 6, 0 }, { 4, 6, 0 },
        { 0, 1, 2 }, { 0, 3, 5 }, { 5, 6, 7 }, { 2, 4, 7 } };
    int const term_axis[8][3] = { { 1, 2, 0 }, { 1, 2, 0 }, { 1, 2, 0 }, { 1, 2, 0 },
        { 2, 2, 2 }, { 1, 1, 1 }, { 2, 2, 2 }, { 1, 1, 1 } };
    int const term_sign[8][3] = { { +1, +1, 0 }, { -1, +1, 0 }, { +1, -1, 0 },
        { -1, -1, 0 }, { +1, +1, +1 }, { +1, +1, +1 }, { -1, -1, -1 },
        { -1, -1, -1 } };
    double prop = *proposal;
    bool has = *proposed != 0;
    for (int r = 0; r < 8; ++r) {
        if (!have[need[r][0]] || !have[need[r][1]] || !have[need[r][2]])
            continue;
        double sum = 0.0;
        for (int t = 0; t < nterms[r]; ++t) {
            int const nb = term_nb[r][t];
            double const d = term_axis[r][t] == 1 ? dx[nb] : dy[nb];
            double const term = term_sign[r][t] > 0 ? f[nb] + d / 2.0 : f[nb] - d / 2.0;
            sum = t == 0 ? term : sum + term;
        }
        double const cand = sum / (double)nterms[r];
        if (!has || cand * 0.9 > prop) {
            prop = cand;
            has = true;
        }
    }
    *proposal = prop;
    *proposed = has ? 1 : 0;
}

// --------------------------------------------- node initialisation from depth
// Surface::initialize_node_from_depth, lib/surface.cc:667-760.  The window of
// node (idx, idy) is four quadrants of (ps/2)^2 pixels; element e of its
// ps^2 pixels (quadrant-major) -> quadrant and pixel; false when the pixel
// lies outside the image.
SMVS_HD bool
window_pixel(Grid const& g, int idx, int idy, int e, int* q, int* xx, int* yy)
{
    int const window = g.ps / 2;
    int const per_q = window * window;
    int const quad = e / per_q, r = e - quad * per_q;
    int const x = idx * g.ps + g.start_x, y = idy * g.ps + g.start_y;
    int const i0 = (quad & 1) ? 0 : -window, j0 = (quad & 2) ? 0 : -window;
    *q = quad;
    *xx = x + i0 + r % window;
    *yy = y + j0 + r / window;
    return *xx >= 0 && *xx < g.width && *yy >= 0 && *yy < g.height;
}

// The node from the statistics of its window (:703-758): median = the
// element of rank count / 2 of the positive depths (std::nth_element),
// lowest[q] = smallest positive depth of quadrant q (0 without one).  false:
// the node stays null (:700-701).
SMVS_HD bool
node_from_window(double median, double const* lowest, int quadrants,
    std::size_t count, double* node)
{
    SMVS_NO_CONTRACT
    if (quadrants == 0 || count < 2)
        return false;
    node[0] = median;
    node[1] = node[2] = node[3] = 0.0;
    double const* a = lowest;
    if (quadrants == 4) {
        node[1] = ((a[1] + a[3]) - (a[0] + a[2])) / 2.0;
        node[2] = ((a[2] + a[3]) - (a[0] + a[1])) / 2.0;
        node[3] = ((a[3] - a[2]) - (a[1] - a[0]));
    } else {
        if ((a[1] == 0 || a[0] == 0) && a[3] != 0 && a[2] != 0)
            node[1] = a[3] - a[2];
        else if ((a[2] == 0 || a[3] == 0) && a[1] != 0 && a[0] != 0)
            node[1] = a[1] - a[0];
        if ((a[0] == 0 || a[2] == 0) && a[3] != 0 && a[1] != 0)
            node[2] = a[3] - a[1];
        else if ((a[1] == 0 || a[2] == 0) && a[0] != 0 && a[2] != 0)
            node[2] = a[2] - a[0];
    }
    return true;
}

// The key of a positive float for the rank selection: positive IEEE floats
// order like their bit patterns.
SMVS_HD uint32_t
depth_key(float d)
{
    union { float f; uint32_t u; } c;
    c.f = d;
    return c.u;
}

// ---------------------------------------------------- isolated patches
// Surface::remove_isolated_patches, lib/surface.cc:887-927, deletes in place
// while it walks the grid column by column, so a deletion changes the counts
// of the patches visited after it.  Cell (x, y) reads the visited state of
// (x-1, y-1..y+1) and (x, y-1) and the unvisited state of the rest: a
// recurrence on a DAG, del(p) = valid(p) and (valid neighbours - deleted
// earlier neighbours < 3), whose unique solution a relaxation reaches from any
// start.  csrc/surface.hip's kernel relaxes all patches at once on bit columns
// (32 rows per word, the neighbour counts by a bit-sliced adder; pinned by
// tests/test_isolated_relaxation_cpu.py); the host mirror walks like the
// reference (csrc/host/surface.cc).

} // namespace smvs_surf

#endif
