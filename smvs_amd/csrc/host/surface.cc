// Host mirror of lib/surface.cc + the evaluation part of lib/bicubic_patch.cc
// and lib/surface_patch.cc.  Topology operations run on the host between
// Newton batches.
#include "surface.h"
#include "topo_math.h"
#include "surface_math.h"

#include <atomic>
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace smvs_amd {

unsigned long
Surface::next_revision(void)
{
    static std::atomic<unsigned long> counter{ 0 };
    return ++counter;
}


PatchEval::PatchEval(double const* nodes16)
{
    std::copy(nodes16, nodes16 + 16, n);
}

double
PatchEval::eval(double x, double y, int kx, int ky) const
{
    // shared with the device topology kernels (topo_math.h)
    return smvs_topo::patch_eval(n, x, y, kx, ky);
}

/* ---------------------------------------------------------------------- */

Surface::Ptr
Surface::create(Bundle::ConstPtr bundle, StereoView::Ptr main_view, int scale,
    FloatImage::ConstPtr init_depth)
{
    // lib/surface.cc:19-53
    Ptr s(new Surface());
    int const width = main_view->get_width(), height = main_view->get_height();
    s->pixel_width = width;
    s->pixel_height = height;
    // (the integer rules of :28-37, shared with the device: surface_math.h)
    smvs_surf::Grid const g = smvs_surf::grid_for_scale(width, height, scale);
    s->scale = g.scale;
    s->patchsize = g.ps;
    s->npx = g.npx;
    s->npy = g.npy;
    s->patch_valid.assign((size_t)s->npx * s->npy, 0);
    s->node_valid.assign((size_t)(s->npx + 1) * (s->npy + 1), 0);
    s->nodes.assign(s->node_valid.size() * 4, 0.0);
    s->start_x = g.start_x;
    s->start_y = g.start_y;
    if (init_depth == nullptr) {
        s->depth = FloatImage::create(width, height, 1);
        s->initialize_depth_from_bundle(bundle, main_view->get_camera(),
            main_view->get_view_id());
    } else {
        // lib/surface.cc:74-79: the positive depths, zero elsewhere
        s->depth = FloatImage::create_for_overwrite(width, height, 1);
        float const* src = init_depth->begin();
        float* dst = s->depth->begin();
        int64_t const n = s->depth->get_pixel_amount();
        for (int64_t p = 0; p < n; ++p)
            dst[p] = src[p] > 0.0f ? src[p] : 0.0f;
    }
    s->fill_patches_from_depth();
    return s;
}

void
Surface::project_bundle(Bundle::ConstPtr bundle, CameraInfo const& cam, int view_id,
    int width, int height, std::vector<int32_t>* pixels, std::vector<float>* depths)
{
    // lib/surface.cc:90-130
    pixels->clear();
    depths->clear();
    double const fwidth2 = (double)width / 2.0, fheight2 = (double)height / 2.0;
    double const fnorm = (double)std::max(width, height);
    for (auto const& feat : bundle->features)
        for (int vid : feat.view_ids)
            if (vid == view_id) {
                float proj[3];
                for (int r = 0; r < 3; ++r)
                    proj[r] = cam.rot[3 * r] * feat.pos[0]
                        + cam.rot[3 * r + 1] * feat.pos[1]
                        + cam.rot[3 * r + 2] * feat.pos[2] + cam.trans[r];
                float const d = proj[2];
                proj[0] = proj[0] * cam.flen / proj[2];
                proj[1] = proj[1] * cam.flen / proj[2];
                float const ix = (float)(proj[0] * fnorm + fwidth2);
                float const iy = (float)(proj[1] * fnorm + fheight2);
                int const x = (int)std::floor(ix), y = (int)std::floor(iy);
                if (x >= 0 && x < width && y >= 0 && y < height) {
                    pixels->push_back(y * width + x);
                    depths->push_back(d);
                }
                break;
            }
    // one entry per pixel, the last write wins: stable sort by pixel, keep the
    // last of every run
    std::vector<std::size_t> order(pixels->size());
    for (std::size_t i = 0; i < order.size(); ++i)
        order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) {
        return (*pixels)[a] < (*pixels)[b]; });
    std::vector<int32_t> up;
    std::vector<float> ud;
    for (std::size_t k = 0; k < order.size(); ++k) {
        if (k + 1 < order.size() && (*pixels)[order[k + 1]] == (*pixels)[order[k]])
            continue;
        up.push_back((*pixels)[order[k]]);
        ud.push_back((*depths)[order[k]]);
    }
    pixels->swap(up);
    depths->swap(ud);
}

void
Surface::initialize_depth_from_bundle(Bundle::ConstPtr bundle,
    CameraInfo const& cam, int view_id)
{
    // lib/surface.cc:90-130 (the list form is what the device path uploads)
    std::vector<int32_t> pixels;
    std::vector<float> depths;
    project_bundle(bundle, cam, view_id, pixel_width, pixel_height, &pixels, &depths);
    float* dst = depth->begin();
    for (std::size_t i = 0; i < pixels.size(); ++i)
        dst[pixels[i]] = depths[i];
}

Surface::Ptr
Surface::from_arrays(int pixel_width, int pixel_height, int scale, int npx, int npy,
    int start_x, int start_y, std::vector<double> const& nodes,
    std::vector<uint8_t> const& node_valid, std::vector<uint8_t> const& patch_valid)
{
    if (node_valid.size() != (std::size_t)(npx + 1) * (npy + 1)
        || nodes.size() != 4 * node_valid.size()
        || patch_valid.size() != (std::size_t)npx * npy)
        throw std::invalid_argument("Surface::from_arrays: array sizes");
    Ptr s(new Surface());
    s->pixel_width = pixel_width;
    s->pixel_height = pixel_height;
    s->scale = scale;
    s->patchsize = 1 << scale;
    s->npx = npx;
    s->npy = npy;
    s->start_x = start_x;
    s->start_y = start_y;
    s->nodes = nodes;
    s->node_valid = node_valid;
    s->patch_valid = patch_valid;
    return s;
}

bool
Surface::node_exists(int idx, int idy) const
{
    if (idx < 0 || idy < 0 || idx > npx || idy > npy)
        return false;
    return node_valid[(size_t)idy * (npx + 1) + idx] != 0;
}

bool
Surface::patch_exists(int idx, int idy) const
{
    if (idx < 0 || idy < 0 || idx >= npx || idy >= npy)
        return false;
    return patch_valid[(size_t)idy * npx + idx] != 0;
}

int
Surface::count_valid_patches(void) const
{
    int n = 0;
    for (uint8_t v : patch_valid)
        n += v ? 1 : 0;
    return n;
}

void
Surface::fill_node_ids_for_patch(std::size_t patch_id,
    std::size_t* node_ids) const
{
    std::size_t const idx = patch_id % npx, idy = patch_id / npx;
    std::size_t const stride = npx + 1;
    node_ids[0] = idy * stride + idx;
    node_ids[1] = node_ids[0] + 1;
    node_ids[2] = node_ids[0] + stride;
    node_ids[3] = node_ids[2] + 1;
}

void
Surface::fill_patch_nodes(std::size_t patch_id, double* nodes16) const
{
    std::size_t ids[4];
    fill_node_ids_for_patch(patch_id, ids);
    for (int k = 0; k < 4; ++k)
        std::copy(nodes.begin() + 4 * ids[k], nodes.begin() + 4 * ids[k] + 4,
            nodes16 + 4 * k);
}

void
Surface::patch_origin(std::size_t patch_id, int* px, int* py) const
{
    *px = start_x + (int)(patch_id % npx) * patchsize;
    *py = start_y + (int)(patch_id / npx) * patchsize;
}

int
Surface::fill_holes(void)
{
    touch();
    // lib/surface.cc:630-651
    // (a patch is filled from its four nodes only: any order gives the same
    // result; rows of the grid, the way the arrays lie in memory)
    int filled = 0;
    int const stride = npx + 1;
    for (int y = 0; y < npy; ++y) {
        uint8_t* pv = &patch_valid[(size_t)y * npx];
        uint8_t const* n0 = &node_valid[(size_t)y * stride];
        uint8_t const* n1 = n0 + stride;
        for (int x = 0; x < npx; ++x) {
            if (pv[x])
                continue;
            if (n0[x] && n0[x + 1] && n1[x] && n1[x + 1]) {
                pv[x] = 1;
                filled += 1;
            }
        }
    }
    return filled;
}

void
Surface::remove_nodes_without_patch(void)
{
    touch();
    // lib/surface.cc:762-869: a node lives while one incident patch lives
    int const stride = npx + 1;
    for (int idy = 0; idy <= npy; ++idy)
        for (int idx = 0; idx <= npx; ++idx) {
            std::size_t const i = (size_t)idy * stride + idx;
            if (!node_valid[i])
                continue;
            if (!patch_exists(idx - 1, idy - 1) && !patch_exists(idx, idy - 1)
                && !patch_exists(idx - 1, idy) && !patch_exists(idx, idy))
                node_valid[i] = 0;
        }
}

void
Surface::remove_isolated_patches(void)
{
    touch();
    // lib/surface.cc:887-927.  The reference walks the grid column by column
    // and deletes in place, so a deletion changes the counts of the patches
    // visited after it: the order is part of the result.  The walk runs on a
    // transposed copy with a one-cell border of zeros (columns contiguous),
    // same visits in the same order.
    int const th = npy + 2;   // cells per (padded) column
    std::vector<uint8_t> t((size_t)(npx + 2) * th, 0);
    for (int y = 0; y < npy; ++y)
        for (int x = 0; x < npx; ++x)
            t[(size_t)(x + 1) * th + y + 1] = patch_valid[(size_t)y * npx + x] ? 1 : 0;
    for (int x = 0; x < npx; ++x) {
        uint8_t* c0 = &t[(size_t)x * th];        // column x - 1
        uint8_t* c1 = c0 + th;                   // column x
        uint8_t* c2 = c1 + th;                   // column x + 1
        for (int y = 0; y < npy; ++y) {
            if (!c1[y + 1])
                continue;
            int const neighbours = c0[y] + c0[y + 1] + c0[y + 2] + c1[y] + c1[y + 2]
                + c2[y] + c2[y + 1] + c2[y + 2];
            if (neighbours < 3)
                c1[y + 1] = 0;
        }
    }
    for (int y = 0; y < npy; ++y)
        for (int x = 0; x < npx; ++x)
            patch_valid[(size_t)y * npx + x] = t[(size_t)(x + 1) * th + y + 1];
    remove_nodes_without_patch();
}

void
Surface::initialize_node_from_depth(int idx, int idy)
{
    touch();
    // lib/surface.cc:667-760
    int const stride = npx + 1;
    std::size_t const id = (size_t)idy * stride + idx;
    if (node_valid[id])
        return;
    int const x = idx * patchsize + start_x, y = idy * patchsize + start_y;
    int const window = patchsize / 2;
    // (the quadrant minima and the median do not depend on the order the
    // window is walked in: rows of the depth map, not columns, and one list
    // for all nodes instead of an allocation per node)
    std::vector<double>& all = window_depths;
    all.clear();
    double lowest[4] = { 0, 0, 0, 0 };
    int quadrants = 4;
    int const dw = depth->width(), dh = depth->height();
    float const* dmap = depth->begin();
    for (int q = 0; q < 4; ++q) {
        int const i0 = (q & 1) ? 0 : -window, j0 = (q & 2) ? 0 : -window;
        bool any = false;
        int const xa = std::max(x + i0, 0), xb = std::min(x + i0 + window, dw);
        for (int yy = std::max(y + j0, 0); yy < std::min(y + j0 + window, dh); ++yy) {
            float const* row = dmap + (size_t)yy * dw;
            for (int xx = xa; xx < xb; ++xx) {
                if (!(row[xx] > 0.0))
                    continue;
                double const d = row[xx];
                lowest[q] = any ? std::min(lowest[q], d) : d;
                any = true;
                all.push_back(d);
            }
        }
        if (!any)
            quadrants -= 1;
    }
    if (quadrants == 0 || all.size() < 2)
        return;
    std::nth_element(all.begin(), all.begin() + all.size() / 2, all.end());
    // (:703-758, shared with the device kernel)
    double node[4];
    if (!smvs_surf::node_from_window(all[all.size() / 2], lowest, quadrants,
            all.size(), node))
        return;
    std::copy(node, node + 4, &nodes[4 * id]);
    node_valid[id] = 1;
}

void
Surface::fill_patches_from_depth(void)
{
    touch();
    // lib/surface.cc:140-152
    for (int i = 0; i < npx + 1; ++i)
        for (int j = 0; j < npy + 1; ++j)
            initialize_node_from_depth(i, j);
    fill_holes();
    remove_nodes_without_patch();
}

FloatImage::Ptr
Surface::get_depth_map(void) const
{
    // lib/surface.cc:155-168 + lib/surface_patch.cc:15-28
    FloatImage::Ptr dmap = FloatImage::create(pixel_width, pixel_height, 1);
    for (std::size_t p = 0; p < patch_valid.size(); ++p) {
        if (!patch_valid[p])
            continue;
        double n16[16];
        fill_patch_nodes(p, n16);
        PatchEval pe(n16);
        int px, py;
        patch_origin(p, &px, &py);
        for (int j = 0; j < patchsize; ++j)
            for (int i = 0; i < patchsize; ++i)
                dmap->at(px + i, py + j, 0) = (float)pe.f((i + 0.5) / patchsize,
                    (j + 0.5) / patchsize);
    }
    return dmap;
}

FloatImage::Ptr
Surface::get_normal_map(float inv_flen) const
{
    // lib/surface.cc:170-183 + lib/surface_patch.cc:30-55
    FloatImage::Ptr normals = FloatImage::create(pixel_width, pixel_height, 3);
    for (std::size_t p = 0; p < patch_valid.size(); ++p) {
        if (!patch_valid[p])
            continue;
        double n16[16];
        fill_patch_nodes(p, n16);
        PatchEval pe(n16);
        int px, py;
        patch_origin(p, &px, &py);
        for (int j = 0; j < patchsize; ++j)
            for (int i = 0; i < patchsize; ++i) {
                double const u = (i + 0.5) / patchsize, v = (j + 0.5) / patchsize;
                double const w = pe.f(u, v), wx = pe.dx(u, v) / patchsize,
                    wy = pe.dy(u, v) / patchsize;
                double const x = px + i + 0.5 - (double)pixel_width / 2.0;
                double const y = py + j + 0.5 - (double)pixel_height / 2.0;
                double nz = (x * wx + y * wy + w) * (double)inv_flen;
                double const len = std::sqrt(wx * wx + wy * wy + nz * nz);
                normals->at(px + i, py + j, 0) = (float)(wx / len);
                normals->at(px + i, py + j, 1) = (float)(-wy / len);
                normals->at(px + i, py + j, 2) = (float)(nz / len);
            }
    }
    return normals;
}

void
Surface::update_nodes(std::vector<double> const& delta)
{
    touch();
    // lib/surface.cc:957-981
    for (std::size_t i = 0; i < node_valid.size(); ++i) {
        if (!node_valid[i])
            continue;
        for (int k = 0; k < 4; ++k)
            nodes[4 * i + k] += delta[4 * i + k];
    }
}

void
Surface::subdivide_patches(void)
{
    touch();
    // lib/surface.cc:983-1107.  Gather form, one new node at a time
    // (smvs_surf::subdivide_node: the arithmetic and the "later patch wins"
    // rule the device kernel runs, csrc/surface.hip).
    smvs_surf::Grid old;
    old.width = pixel_width; old.height = pixel_height;
    old.scale = scale; old.ps = patchsize;
    old.npx = npx; old.npy = npy;
    old.start_x = start_x; old.start_y = start_y;
    int off_x = 0, off_y = 0;
    smvs_surf::Grid const g = smvs_surf::grid_subdivided(old, &off_x, &off_y);
    int const new_stride = g.npx + 1;
    std::vector<double> new_nodes((size_t)new_stride * (g.npy + 1) * 4, 0.0);
    std::vector<uint8_t> new_valid((size_t)new_stride * (g.npy + 1), 0);
    for (int Y = 0; Y <= g.npy; ++Y)
        for (int X = 0; X <= g.npx; ++X) {
            std::size_t const id = (size_t)Y * new_stride + X;
            double out[4];
            if (smvs_surf::subdivide_node(old.npx, old.npy, off_x, off_y, nodes.data(),
                    node_valid.data(), patch_valid.data(), X, Y, out)) {
                std::copy(out, out + 4, &new_nodes[4 * id]);
                new_valid[id] = 1;
            }
        }
    scale = g.scale;
    patchsize = g.ps;
    start_x = g.start_x;
    start_y = g.start_y;
    npx = g.npx;
    npy = g.npy;
    nodes.swap(new_nodes);
    node_valid.swap(new_valid);
    patch_valid.assign((size_t)npx * npy, 0);
    fill_holes();
    remove_nodes_without_patch();
}

int
Surface::expand(void)
{
    touch();
    // lib/surface.cc:482-628: two rounds of extrapolating new nodes from
    // complete triples of neighbours (smvs_surf::expand_node: the rules, their
    // order and check_swap_nodes :472-480, shared with the device kernel).
    // A round reads the surface as it was at its start; its proposals are
    // applied after every node has been visited.
    int const stride = npx + 1;
    std::size_t const count = node_valid.size();
    std::vector<double> proposal(count, 0.0);
    std::vector<uint8_t> proposed(count, 0);
    for (int round = 0; round < 2; ++round) {
        for (std::size_t id = 0; id < count; ++id) {
            if (node_valid[id] && !proposed[id])
                continue;
            smvs_surf::expand_node(npx, npy, nodes.data(), node_valid.data(),
                (int)(id % stride), (int)(id / stride), &proposal[id], &proposed[id]);
        }
        for (std::size_t id = 0; id < count; ++id)
            if (proposed[id]) {
                nodes[4 * id] = proposal[id];
                nodes[4 * id + 1] = nodes[4 * id + 2] = nodes[4 * id + 3] = 0.0;
                node_valid[id] = 1;
            }
    }
    int const filled = fill_holes();
    remove_nodes_without_patch();
    return filled;
}

} // namespace smvs_amd
