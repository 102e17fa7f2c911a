// lib/view_selection.cc:14-161 on flat records.  The reference keeps its
// candidates in a std::map keyed by distance and a std::multimap keyed by the
// number of shared features; here both are sorted vectors with the same
// tie rules (a later view at exactly the same distance replaces the earlier
// one; equal feature counts keep their insertion order).  Float arithmetic in
// the reference's order (MVE matrix / vector products accumulate left to
// right from zero, tests/golden/README.md M17).
#include "view_selection.h"

#include <algorithm>
#include <cmath>

namespace smvs_amd {

namespace {

// camera centre -R^T t (CameraInfo::fill_camera_pos, README M15)
void
camera_position(CameraInfo const& cam, float pos[3])
{
    for (int c = 0; c < 3; ++c) {
        float acc = 0.0f;
        for (int r = 0; r < 3; ++r)
            acc += cam.rot[3 * r + c] * cam.trans[r];
        pos[c] = -acc;
    }
}

// z of a world point in the camera frame: third row of [R | t] applied to
// (p, 1) (fill_world_to_cam + Matrix4f::mult(Vec3f, 1), README M17, M22)
float
camera_depth(CameraInfo const& cam, float const p[3])
{
    float acc = 0.0f;
    for (int c = 0; c < 3; ++c)
        acc += cam.rot[6 + c] * p[c];
    return acc + cam.trans[2] * 1.0f;
}

// first entry of the inverse calibration (README M13): 1 / (flen max(w, h))
float
inverse_focal(CameraInfo const& cam, int width, int height)
{
    float mat[9];
    cam.fill_inverse_calibration(mat, (float)width, (float)height);
    return mat[0];
}

bool
sees(Bundle::Feature3D const& f, int view_id)
{
    return std::find(f.view_ids.begin(), f.view_ids.end(), view_id)
        != f.view_ids.end();
}

float
dot3(float const a[3], float const b[3])
{
    float acc = 0.0f;
    for (int c = 0; c < 3; ++c)
        acc += a[c] * b[c];
    return acc;
}

} // namespace

std::vector<std::size_t>
ViewSelection::get_neighbors_for_view(std::size_t view) const
{
    // view_selection.cc:14-21
    return bundle != nullptr ? bundle_based_selection(view)
        : position_based_selection(view);
}

std::vector<std::size_t>
ViewSelection::get_sorted_neighbors(std::size_t view) const
{
    // view_selection.cc:134-159: every other view with a camera, by the
    // distance of its centre from the main view's
    float main_pos[3];
    camera_position(views[view].cam, main_pos);
    struct Candidate { float dist; std::size_t index; };
    std::vector<Candidate> found;
    for (std::size_t i = 0; i < views.size(); ++i) {
        if (!views[i].present || i == view || views[i].cam.flen == 0.0f)
            continue;
        float pos[3], d[3];
        camera_position(views[i].cam, pos);
        for (int c = 0; c < 3; ++c)
            d[c] = main_pos[c] - pos[c];
        found.push_back({ std::sqrt(dot3(d, d)), i });
    }
    // std::map<float, index>: ascending keys, assignment to an existing key
    // replaces its value -- of several views at one distance the last stays
    std::stable_sort(found.begin(), found.end(),
        [](Candidate const& a, Candidate const& b) { return a.dist < b.dist; });
    std::vector<std::size_t> order;
    for (std::size_t k = 0; k < found.size(); ++k) {
        if (k + 1 < found.size() && found[k + 1].dist == found[k].dist)
            continue;
        order.push_back(found[k].index);
    }
    return order;
}

std::vector<std::size_t>
ViewSelection::bundle_based_selection(std::size_t view) const
{
    // view_selection.cc:23-97
    std::vector<std::size_t> result;
    ViewInfo const& main_view = views[view];
    if (!main_view.has_image)
        return result;
    float const main_inv_f = inverse_focal(main_view.cam, main_view.width,
        main_view.height);

    // the main view's features and the size of a pixel at each of them
    std::vector<Bundle::Feature3D const*> seen;
    std::vector<float> footprint;
    for (Bundle::Feature3D const& f : bundle->features)
        if (sees(f, main_view.id)) {
            seen.push_back(&f);
            footprint.push_back(camera_depth(main_view.cam, f.pos) * main_inv_f);
        }

    // shared features with a similar footprint in the 50 nearest views
    struct Scored { std::size_t matches; int id; };
    std::vector<Scored> scored;
    std::vector<std::size_t> const nearest = get_sorted_neighbors(view);
    for (std::size_t i = 0; i < nearest.size() && i < 50; ++i) {
        ViewInfo const& v = views[nearest[i]];
        // (:62: the id is compared with the main view's INDEX)
        if ((std::size_t)v.id == view || v.cam.flen == 0.0f || !v.has_image)
            continue;
        float const inv_f = inverse_focal(v.cam, v.width, v.height);
        std::size_t matches = 0;
        for (std::size_t f = 0; f < seen.size(); ++f) {
            if (!sees(*seen[f], v.id))
                continue;
            float const theirs = camera_depth(v.cam, seen[f]->pos) * inv_f;
            if (std::min(theirs, footprint[f]) / std::max(theirs, footprint[f])
                > 0.6)
                matches += 1;
        }
        scored.push_back({ matches, v.id });
    }
    // std::multimap<count, id, greater>: descending counts, ties in
    // insertion order
    std::stable_sort(scored.begin(), scored.end(),
        [](Scored const& a, Scored const& b) { return a.matches > b.matches; });
    for (Scored const& s : scored) {
        if (s.matches > 10)
            result.push_back((std::size_t)s.id);   // (:91: views[id])
        if (result.size() >= opts.num_neighbors)
            break;
    }
    return result;
}

std::vector<std::size_t>
ViewSelection::position_based_selection(std::size_t view) const
{
    // view_selection.cc:99-132: nearest first, dropping views that look
    // elsewhere (viewing directions more than ~49 degrees apart) or are
    // upside down relative to the main view.  Viewing direction = third row
    // of R, "up" = third column (rot[2], rot[5], rot[8]) as the reference
    // reads it.  (README M23)
    CameraInfo const& main_cam = views[view].cam;
    float const main_dir[3] = { main_cam.rot[6], main_cam.rot[7], main_cam.rot[8] };
    float const main_up[3] = { main_cam.rot[2], main_cam.rot[5], main_cam.rot[8] };
    std::vector<std::size_t> result;
    for (std::size_t i : get_sorted_neighbors(view)) {
        CameraInfo const& cam = views[i].cam;
        float const dir[3] = { cam.rot[6], cam.rot[7], cam.rot[8] };
        float const up[3] = { cam.rot[2], cam.rot[5], cam.rot[8] };
        if (dot3(main_up, up) < 0.0f || dot3(main_dir, dir) < 0.65)
            continue;
        result.push_back(i);
    }
    return result;
}

} // namespace smvs_amd
