// Host mirror of smvs::SGMStereo (reference: lib/sgm_stereo.h:21-88).  The
// cost volumes, the 8-path aggregation, the WTA, the left/right check and the
// two-neighbour merge run on the device (smvs_sgm_depth_for_view); the host
// keeps the depth range from the bundle and the image half-sizing.
#pragma once

#include "image.h"
#include "stereo_view.h"

namespace smvs_amd {

class SGMStereo
{
public:
    struct Options  // lib/sgm_stereo.h:24-34
    {
        int debug_lvl = 0;
        int scale = 1;
        int num_steps = 128;
        float min_depth = 0.0f;
        float max_depth = 0.0f;
        uint16_t penalty1 = 6;
        uint16_t penalty2 = 96;
        int device = 0;      // HIP device (not in the reference)
    };

    SGMStereo(Options const& opts, StereoView::Ptr main,
        StereoView::Ptr neighbor);

    static FloatImage::Ptr reconstruct(Options sgm_opts,
        StereoView::Ptr main_view, StereoView::Ptr neighbor,
        Bundle::ConstPtr bundle = nullptr);

    FloatImage::Ptr run_sgm(float min_depth, float max_depth);

    static void fill_depth_range_for_view(Bundle::ConstPtr bundle,
        StereoView::Ptr view, float* range);

private:
    Options opts;
    StereoView::Ptr main;
    StereoView::Ptr neighbor;
    ByteImage::ConstPtr main_image;
    ByteImage::ConstPtr neighbor_image;
};

// app/smvsrecon.cc:346-384: SGM against the first two neighbours, averaged
FloatImage::Ptr reconstruct_sgm_depth_for_view(SGMStereo::Options opts,
    StereoView::Ptr main_view, std::vector<StereoView::Ptr> const& neighbors,
    Bundle::ConstPtr bundle = nullptr);

} // namespace smvs_amd
