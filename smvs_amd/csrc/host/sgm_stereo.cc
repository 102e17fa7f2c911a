#include "sgm_stereo.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

#include "../../../include/smvs_hip.h"

namespace smvs_amd {

SGMStereo::SGMStereo(Options const& opts, StereoView::Ptr main,
    StereoView::Ptr neighbor)
    : opts(opts), main(main), neighbor(neighbor)
{
    // lib/sgm_stereo.cc:27-44
    this->main_image = main->get_byte_image();
    for (int i = 0; i < opts.scale; ++i)
        this->main_image = imgtools::rescale_half_size(this->main_image);
    this->neighbor_image = neighbor->get_byte_image();
    for (int i = 0; i < opts.scale; ++i)
        this->neighbor_image = imgtools::rescale_half_size(this->neighbor_image);
}

FloatImage::Ptr
SGMStereo::run_sgm(float min_depth, float max_depth)
{
    // lib/sgm_stereo.cc:98-124 on the device
    float M[9], t[3];
    this->main->get_camera().fill_reprojection(this->neighbor->get_camera(),
        (float)main_image->width(), (float)main_image->height(),
        (float)neighbor_image->width(), (float)neighbor_image->height(), M, t);
    FloatImage::Ptr depth = FloatImage::create(main_image->width(),
        main_image->height(), 1);
    int const rc = smvs_sgm_run(opts.device, main_image->begin(),
        main_image->width(), main_image->height(), neighbor_image->begin(),
        neighbor_image->width(), neighbor_image->height(), M, t, min_depth,
        max_depth, opts.num_steps, opts.penalty1, opts.penalty2,
        depth->begin(), nullptr, nullptr, nullptr);
    if (rc != SMVS_OK)
        throw std::runtime_error(std::string("smvs_sgm_run: ")
            + smvs_last_error());
    return depth;
}

void
SGMStereo::fill_depth_range_for_view(Bundle::ConstPtr bundle,
    StereoView::Ptr view, float* range)
{
    // lib/sgm_stereo.cc:669-720
    std::vector<float> depth_values;
    int const width = view->get_width(), height = view->get_height();
    CameraInfo const& cam = view->get_camera();
    double const fwidth2 = (double)width / 2.0, fheight2 = (double)height / 2.0;
    double const fnorm = (double)std::max(width, height);
    for (auto const& feat : bundle->features)
        for (int vid : feat.view_ids)
            if (vid == view->get_view_id()) {
                float proj[3];
                for (int r = 0; r < 3; ++r)
                    proj[r] = cam.rot[3 * r] * feat.pos[0]
                        + cam.rot[3 * r + 1] * feat.pos[1]
                        + cam.rot[3 * r + 2] * feat.pos[2] + cam.trans[r];
                float const depth = proj[2];
                proj[0] = proj[0] * cam.flen / proj[2];
                proj[1] = proj[1] * cam.flen / proj[2];
                float const ix = (float)(proj[0] * fnorm + fwidth2);
                float const iy = (float)(proj[1] * fnorm + fheight2);
                int const x = (int)std::floor(ix), y = (int)std::floor(iy);
                if (x >= 0 && x < width && y >= 0 && y < height)
                    depth_values.push_back(depth);
                break;
            }
    std::sort(depth_values.begin(), depth_values.end());
    if (depth_values.size() < 2) {
        range[0] = 0.3f;
        range[1] = 1.1f;
    } else {
        range[0] = depth_values.front() * 0.7f;
        range[1] = (float)(depth_values[(depth_values.size() * 99) / 100] * 5.0);
    }
}

FloatImage::Ptr
SGMStereo::reconstruct(Options sgm_opts, StereoView::Ptr main_view,
    StereoView::Ptr neighbor, Bundle::ConstPtr bundle)
{
    // lib/sgm_stereo.cc:46-96
    float depth_range[2] = { sgm_opts.min_depth, sgm_opts.max_depth };
    if (bundle != nullptr && sgm_opts.max_depth == 0.0)
        fill_depth_range_for_view(bundle, main_view, depth_range);
    SGMStereo sgm1(sgm_opts, main_view, neighbor);
    FloatImage::Ptr d_main = sgm1.run_sgm(depth_range[0], depth_range[1]);
    if (bundle != nullptr && sgm_opts.max_depth == 0.0)
        fill_depth_range_for_view(bundle, neighbor, depth_range);
    SGMStereo sgm2(sgm_opts, neighbor, main_view);
    FloatImage::Ptr d_neig = sgm2.run_sgm(depth_range[0], depth_range[1]);

    float Mf[9], tf[3];
    main_view->get_camera().fill_reprojection(neighbor->get_camera(),
        (float)d_main->width(), (float)d_main->height(), (float)d_neig->width(),
        (float)d_neig->height(), Mf, tf);
    double M[9], t[3];
    for (int i = 0; i < 9; ++i)
        M[i] = Mf[i];
    for (int i = 0; i < 3; ++i)
        t[i] = tf[i];
    // left / right consistency: integer pixel coordinates, truncating lookup
    int const cut = (int)(0.03 * std::max(d_neig->width(), d_neig->height()));
    for (int x = 0; x < d_main->width(); ++x)
        for (int y = 0; y < d_main->height(); ++y) {
            float& dm = d_main->at(x, y, 0);
            if (dm == 0)
                continue;
            double const w = dm;
            double const p = M[0] * x + M[1] * y + M[2];
            double const q = M[3] * x + M[4] * y + M[5];
            double const r = M[6] * x + M[7] * y + M[8];
            double const d = w * r + t[2];
            double const cx = (w * p + t[0]) / d, cy = (w * q + t[1]) / d;
            if (cx < cut || cx >= d_neig->width() - cut || cy < cut
                || cy >= d_neig->height() - cut) {
                dm = 0;
                continue;
            }
            float const cdepth = (float)d;
            float const ndepth = d_neig->at((int)cx, (int)cy, 0);
            float const ratio = std::min(cdepth, ndepth)
                / std::max(cdepth, ndepth);
            if (ndepth == 0 || ratio < 0.8)
                dm = 0;
        }
    return d_main;
}

FloatImage::Ptr
reconstruct_sgm_depth_for_view(SGMStereo::Options opts,
    StereoView::Ptr main_view, std::vector<StereoView::Ptr> const& neighbors,
    Bundle::ConstPtr bundle)
{
    FloatImage::Ptr d1 = SGMStereo::reconstruct(opts, main_view, neighbors[0],
        bundle);
    if (neighbors.size() > 1) {
        FloatImage::Ptr d2 = SGMStereo::reconstruct(opts, main_view,
            neighbors[1], bundle);
        for (int p = 0; p < d1->get_pixel_amount(); ++p) {
            if (d2->at(p) == 0.0f)
                continue;
            if (d1->at(p) == 0.0f) {
                d1->at(p) = d2->at(p);
                continue;
            }
            d1->at(p) = (d1->at(p) + d2->at(p)) * 0.5f;
        }
    }
    main_view->write_depth_to_view(d1, "smvs-sgm");
    return d1;
}

} // namespace smvs_amd
