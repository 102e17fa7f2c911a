// MVE scene I/O for the host mirror and the scene-level driver of smvsrecon
// (reference: app/smvsrecon.cc:388-752; mve::Scene / mve::View / mve::Bundle
// are MVE's, not in the reference tree: every format detail below is
// [MVE-unverified] and listed in tests/golden/README.md).
//
// What is read: views/view_XXXX.mve/meta.ini ([camera] focal_length,
// pixel_aspect, principal_point, rotation, translation; [view] id, name), the
// bundle synth_0.out ("drews 1.0"), raw .mvei images (u8 / float) and PNG
// byte images (csrc/host/png_io.cc, on zlib) -- what makescene leaves in a
// view directory ("undistorted.png").  JPEG embeddings are not decoded (no
// codec in this tree; a view whose only copy of the embedding is a .jpg is
// reported with that reason).  Results are written like StereoView::
// write_depth_to_view / write_image_to_view do (lib/stereo_view.h:100-130):
// <output>.mvei (depth, MVE ray-length convention), <output>N.mvei (normals),
// smvs-sgm.mvei.
#pragma once

#include <string>
#include <vector>

#include "image.h"
#include "view_selection.h"

namespace smvs_amd {

// \211MVE_IMAGE\n + int32 width, height, channels, type + raw data
ByteImage::Ptr load_mvei_u8(std::string const& path);
FloatImage::Ptr load_mvei_float(std::string const& path);
void save_mvei(std::string const& path, ByteImage::ConstPtr image);
void save_mvei(std::string const& path, FloatImage::ConstPtr image);
// width, height, channels, type of an .mvei file without loading it
bool mvei_header(std::string const& path, int* whct);

struct SceneView
{
    bool present = false;        // a null View::Ptr otherwise
    int id = 0;
    std::string name, directory;
    CameraInfo camera;
    bool is_camera_valid(void) const { return camera.flen > 0.0f; }
    // mve::View::has_image: <embedding>.mvei or <embedding>.png (a .jpg alone
    // does not count: it cannot be decoded here)
    bool has_image(std::string const& embedding) const;
    // the file of an embedding: the existing .mvei, else the existing .png,
    // else where a NEW embedding of that name is written (.mvei)
    std::string image_path(std::string const& embedding) const;
    // byte image of an embedding (mve::View::get_byte_image), any container
    ByteImage::Ptr load_byte_image(std::string const& embedding) const;
    // width, height, channels without decoding (mve::View::get_image_proxy)
    bool image_size(std::string const& embedding, int* whc) const;
};

// mve::image::rescale_half_size_gaussian<uint8_t> (sigma^2 = 0.75), the filter
// smvsrecon pre-scales its input embedding with (app/smvsrecon.cc:634-647)
// [MVE-unverified, tests/golden/README.md M29]
ByteImage::Ptr rescale_half_size_gaussian(ByteImage::ConstPtr image);

class Scene
{
public:
    typedef std::shared_ptr<Scene> Ptr;
    // mve::Scene::create(path): views/view_*.mve, list index = view id
    static Ptr create(std::string const& path);
    std::vector<SceneView>& get_views(void) { return views; }
    // mve::Scene::get_bundle(): <path>/synth_0.out; throws if unreadable
    Bundle::Ptr get_bundle(void) const;
    std::string const& get_path(void) const { return path; }
private:
    std::string path;
    std::vector<SceneView> views;
};

Bundle::Ptr load_mve_bundle(std::string const& path);

// AppSettings of app/smvsrecon.cc:37-71 (the part that reaches the per-view task)
struct ReconSettings
{
    std::vector<int> view_ids;              // empty: every view with a valid camera
    std::string image_embedding = "undistorted";
    float regularization = 1.0f;            // alpha
    int output_scale = 2;                   // -o
    int input_scale = -1;                   // -s; < 0: automatic (:477-500)
    std::size_t max_pixels = 1700000;       // --max-pixels (:48)
    bool use_shading = false;               // -S
    float light_surf_regularization = 0.0f;
    bool gamma_correction = false;
    bool use_sgm = true;
    bool force_recon = false, force_sgm = false, full_optimization = false;
    float sgm_min = 0.0f, sgm_max = 0.0f;
    int sgm_scale = 1;
    std::size_t num_neighbors = 6, min_neighbors = 3;
    int first_device = 0, num_devices = 1, views_in_flight = 2;
};

struct ReconReport
{
    std::vector<int> reconstructed, skipped_few_neighbors, already_done;
    double seconds = 0.0;
    std::string output_name;
    std::string input_name;      // the embedding the views were read from
    int input_scale = 0;         // as used (the automatic choice resolved)
};

// The body of smvsrecon's main between scene loading and mesh generation
// (app/smvsrecon.cc:400-745): view list, ViewSelection, one ViewQueue task per
// reference view (StereoViews, SGM front end, DepthOptimizer::optimize),
// embeddings saved into the view directories.
ReconReport reconstruct_scene(std::string const& scene_path,
    ReconSettings const& settings);

} // namespace smvs_amd
