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

namespace {

// One (main, neighbour) pair as the device entry wants it: the neighbour's raw
// bytes (the device desaturates and halves them: SGMStereo's constructor,
// lib/sgm_stereo.cc:27-39), both reprojections at the SGM-scale sizes and both
// depth ranges (lib/sgm_stereo.cc:46-62).
struct PairInputs {
    ByteImage::ConstPtr neighbor_bytes;
    int channels;
    smvs_sgm_neighbor dev;
};

void
sgm_scale_size(int scale, int* w, int* h)
{
    for (int i = 0; i < scale; ++i) {
        *w = (*w + 1) >> 1;
        *h = (*h + 1) >> 1;
    }
}

void
prepare_pair(SGMStereo::Options const& o, StereoView::Ptr main_view,
    StereoView::Ptr neighbor, Bundle::ConstPtr bundle, PairInputs* out)
{
    out->neighbor_bytes = neighbor->get_raw_bytes();
    out->channels = out->neighbor_bytes->channels();
    smvs_sgm_neighbor& d = out->dev;
    d.image = out->neighbor_bytes->begin();
    d.width = out->neighbor_bytes->width();      // full resolution
    d.height = out->neighbor_bytes->height();
    int mwi = main_view->get_width(), mhi = main_view->get_height();
    int nwi = d.width, nhi = d.height;
    sgm_scale_size(o.scale, &mwi, &mhi);
    sgm_scale_size(o.scale, &nwi, &nhi);
    float const mw = (float)mwi, mh = (float)mhi, nw = (float)nwi, nh = (float)nhi;
    main_view->get_camera().fill_reprojection(neighbor->get_camera(), mw, mh,
        nw, nh, d.M_fwd, d.t_fwd);
    neighbor->get_camera().fill_reprojection(main_view->get_camera(), nw, nh,
        mw, mh, d.M_bwd, d.t_bwd);
    d.range_main[0] = d.range_neighbor[0] = o.min_depth;
    d.range_main[1] = d.range_neighbor[1] = o.max_depth;
    if (bundle != nullptr && o.max_depth == 0.0) {
        SGMStereo::fill_depth_range_for_view(bundle, main_view, d.range_main);
        SGMStereo::fill_depth_range_for_view(bundle, neighbor, d.range_neighbor);
    }
}

FloatImage::Ptr
run_pairs(SGMStereo::Options const& o, StereoView::Ptr main_view,
    std::vector<PairInputs> const& pairs)
{
    std::vector<smvs_sgm_neighbor> dev;
    std::vector<int> channels;
    for (auto const& p : pairs) {
        dev.push_back(p.dev);
        channels.push_back(p.channels);
    }
    ByteImage::ConstPtr main_bytes = main_view->get_raw_bytes();
    int w = main_bytes->width(), h = main_bytes->height();
    sgm_scale_size(o.scale, &w, &h);
    FloatImage::Ptr depth = FloatImage::create_for_overwrite(w, h, 1);
    // desaturate + half-size on the device, then 4 x run_sgm, L/R check, merge
    int const rc = smvs_sgm_depth_for_view_raw(o.device, main_bytes->begin(),
        main_bytes->width(), main_bytes->height(), main_bytes->channels(), dev.data(),
        channels.data(), (int)dev.size(), o.scale, o.num_steps, o.penalty1, o.penalty2,
        depth->begin());
    if (rc != SMVS_OK)
        throw std::runtime_error(std::string("smvs_sgm_depth_for_view_raw: ")
            + smvs_last_error());
    return depth;
}

} // namespace

FloatImage::Ptr
SGMStereo::reconstruct(Options sgm_opts, StereoView::Ptr main_view,
    StereoView::Ptr neighbor, Bundle::ConstPtr bundle)
{
    // lib/sgm_stereo.cc:46-96: both run_sgm calls and the left / right
    // consistency check on the device
    std::vector<PairInputs> pairs(1);
    prepare_pair(sgm_opts, main_view, neighbor, bundle, &pairs[0]);
    return run_pairs(sgm_opts, main_view, pairs);
}

FloatImage::Ptr
reconstruct_sgm_depth_for_view(SGMStereo::Options opts,
    StereoView::Ptr main_view, std::vector<StereoView::Ptr> const& neighbors,
    Bundle::ConstPtr bundle)
{
    // app/smvsrecon.cc:346-384: SGMStereo::reconstruct against the first two
    // neighbours and the merge of the two checked maps, in one device call
    if (neighbors.empty())
        throw std::invalid_argument("reconstruct_sgm_depth_for_view: no neighbour");
    std::vector<PairInputs> pairs(std::min<std::size_t>(neighbors.size(), 2));
    for (std::size_t k = 0; k < pairs.size(); ++k)
        prepare_pair(opts, main_view, neighbors[k], bundle, &pairs[k]);
    FloatImage::Ptr d1 = run_pairs(opts, main_view, pairs);
    // (app/smvsrecon.cc:383: write_depth_to_view -- the conversion of the stored
    // copy to MVE's convention waits until the embedding is read; the optimizer
    // of this view takes the map as it is and converts on the device)
    main_view->write_depth_to_view_deferred(d1, "smvs-sgm");
    return d1;
}

} // namespace smvs_amd
