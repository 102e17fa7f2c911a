// Host mirror of smvs::StereoView (reference: lib/stereo_view.h:21-79): the
// per-view image bundle consumed by DepthOptimizer / SGMStereo.  The MVE view
// + embedding name of the reference is replaced by an in-memory byte image
// and camera; named embeddings written by the optimizer are kept in a map.
#pragma once

#include <map>
#include <mutex>
#include <string>

#include "image.h"

namespace smvs_amd {

class StereoView
{
public:
    typedef std::shared_ptr<StereoView> Ptr;
    typedef std::shared_ptr<const StereoView> ConstPtr;

    // lib/stereo_view.h:28-30
    static Ptr create(int view_id, ByteImage::ConstPtr image,
        CameraInfo const& camera, bool initialize_linear = false,
        bool gamma_correction = false);

    void set_scale(int scale, bool debug = false);
    // planes of a scale computed elsewhere (smvs_ctx_set_scale on the device)
    void set_scale_planes(FloatImage::Ptr gradients, FloatImage::Ptr hessian);
    ByteImage::ConstPtr get_raw_bytes(void) const { return bytes; }

    int get_width(void) const { return bytes->width(); }
    int get_height(void) const { return bytes->height(); }
    int get_view_id(void) const { return view_id; }
    CameraInfo const& get_camera(void) const { return camera; }
    float get_flen(void) const;
    float get_inverse_flen(void) const;
    ByteImage::ConstPtr get_byte_image(void) const;
    // The float image (bytes / 255, lib/stereo_view.cc:16-22) is converted on
    // first use: the device pipeline reads the main view's only (bilateral
    // upsample), the neighbours' never -- 25 MB and 6 ms per 1080p neighbour.
    FloatImage::ConstPtr get_image(void) const;
    FloatImage::ConstPtr get_scaleimage(void) const { return scaleimage; }
    FloatImage::ConstPtr get_image_gradients(void) const { return image_grad; }
    FloatImage::ConstPtr get_image_hessian(void) const { return image_hessian; }
    FloatImage::ConstPtr get_shading_image(void) const { return shading; }
    FloatImage::ConstPtr get_shading_gradients(void) const { return shading_grad; }
    FloatImage::ConstPtr get_linear_image(void) const { return linear_image; }

    // "smvs-sgm" embedding: stored in MVE convention (ray length); the getter
    // converts to z-depth (lib/stereo_view.h:121-130)
    FloatImage::Ptr get_sgm_depth(void) const;
    bool has_embedding(std::string const& name) const;
    FloatImage::Ptr get_embedding(std::string const& name) const;
    std::map<std::string, FloatImage::Ptr> const& get_embeddings(void) const;

    void write_image_to_view(FloatImage::Ptr image, std::string const& name);
    void write_depth_to_view(FloatImage::Ptr depth, std::string const& name);
    // Not in the reference: write_depth_to_view whose conversion to MVE's
    // convention (0.5 M pixels on the host for an SGM map) happens when the
    // embedding is first asked for -- any getter sees exactly what
    // write_depth_to_view would have stored.  A consumer that can convert on the
    // device (DepthOptimizer::create_initial_surface) takes the z-depth map with
    // get_deferred_depth() and the embedding is never formed on the host unless
    // somebody reads or saves it.
    void write_depth_to_view_deferred(FloatImage::Ptr depth, std::string const& name);
    FloatImage::Ptr get_deferred_depth(std::string const& name) const;

private:
    StereoView(void) = default;
    void initialize_linear(bool gamma_correction);

private:
    int view_id = 0;
    CameraInfo camera;
    ByteImage::ConstPtr bytes;
    mutable FloatImage::ConstPtr image;   // lazily converted, see get_image()
    mutable std::once_flag image_once;
    FloatImage::Ptr scaleimage, image_grad, image_hessian;
    FloatImage::Ptr linear_image, shading, shading_grad;
    mutable std::map<std::string, FloatImage::Ptr> embeddings;
    // z-depth maps of write_depth_to_view_deferred not yet converted and stored
    mutable std::map<std::string, FloatImage::Ptr> deferred_depth;
    void store_deferred(std::string const& name) const;
};

// image helpers shared with SGMStereo / DepthOptimizer
namespace imgtools {
FloatImage::Ptr blur_gaussian(FloatImage::ConstPtr in, float sigma);
FloatImage::Ptr desaturate(FloatImage::ConstPtr in);
ByteImage::Ptr desaturate(ByteImage::ConstPtr in);
ByteImage::Ptr rescale_half_size(ByteImage::ConstPtr in);
void gradients_and_hessian(FloatImage::ConstPtr input, FloatImage::Ptr gradient,
    FloatImage::Ptr hessian);
// z-depth <-> ray length (mve::image::depthmap_convert_conventions)
void depthmap_convert_conventions(FloatImage::Ptr dm, float const* invproj,
    bool to_mve);
}

} // namespace smvs_amd
