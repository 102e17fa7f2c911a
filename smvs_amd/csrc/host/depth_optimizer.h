// Host mirror of smvs::DepthOptimizer (reference: lib/depth_optimizer.h:27-122).
// Same Options, constructor and optimize() / get_depth() / get_normals()
// surface; the Newton loop (lib/depth_optimizer.cc:219-304), the lighting
// accumulation and the bilateral upsample run on the GPU through the C ABI of
// include/smvs_hip.h, and so do the topology operations between the batches
// (csrc/topology.hip, csrc/surface.hip): the surface stays in the context.
#pragma once

#include <string>
#include <vector>

#include "image.h"
#include "stereo_view.h"
#include "surface.h"

#include "../../../include/smvs_hip.h"

namespace smvs_amd {

// Measurement hook (not in the reference): while a recording is on -- per host
// thread -- every Newton batch leaves a device-resident clone of what its loop
// reads (smvs_ctx_clone_loop_state) together with the parameters it was run
// with, taken just before smvs_gn_run_loop.  bench.py replays exactly these
// loops as its timed region (the Newton loops of optimize()).
struct RecordedLoop
{
    smvs_ctx* ctx;
    smvs_gn_loop_params params;
    int scale, iter;
};
std::vector<RecordedLoop>& recorded_loops(void);   // of this thread
bool& recording_loops(void);                       // of this thread

class DepthOptimizer
{
public:
    struct Options  // lib/depth_optimizer.h:30-42
    {
        double regularization = 0.001;
        double light_surf_regularization = 0.0;
        int num_iterations = 10;
        int min_scale = 1;
        int debug_lvl = 0;
        bool use_shading = false;
        bool use_sgm = false;
        bool full_optimization = false;
        std::string output_name = "smvs";
        int device = 0;  // HIP device (not in the reference)
        int solver = 0;  // smvs_solver_mode of include/smvs_hip.h (not in the reference)
    };

    struct IterationLog
    {
        int scale, iter, newton_steps, valid_patches, cg_iterations;
        long long active_patch_steps;   // sum over the steps of active patches
        double loop_seconds;            // wall time of the device Newton loop
    };

public:
    DepthOptimizer(StereoView::Ptr main_view,
        std::vector<StereoView::Ptr> const& sub_views,
        Bundle::ConstPtr bundle, Options const& options);
    // lib/depth_optimizer.h:53-56, 127-134: takes an existing surface and
    // leaves the bundle null.  As in the reference, optimize() then rebuilds
    // the surface (lib/depth_optimizer.cc:56 -> :35-51), which without a bundle
    // only works with use_sgm (the reference dereferences the null bundle
    // otherwise; here that case throws std::invalid_argument); get_depth() /
    // get_normals() evaluate the surface that was passed in.
    DepthOptimizer(StereoView::Ptr main_view,
        std::vector<StereoView::Ptr> const& sub_views,
        Surface::Ptr surface, Options const& options);
    ~DepthOptimizer(void);
    DepthOptimizer(DepthOptimizer const&) = delete;
    DepthOptimizer& operator=(DepthOptimizer const&) = delete;

    void optimize(void);

    FloatImage::Ptr get_depth(void);
    FloatImage::Ptr get_normals(void);

    std::vector<IterationLog> const& get_log(void) const { return log; }
    bool has_lighting(void) const { return lit; }
    double const* get_lighting(void) const { return lighting; }

    // The scale-space planes live on the device; this copies the current
    // scale's gradient / Hessian planes into the StereoViews (not in the
    // reference, where StereoView::set_scale fills them on the host).
    void download_scale_planes(void);

private:
    void prepare_correspondences(void);
    void create_initial_surface(void);
    void create_subview_surfaces(void);
    void run_newton_iterations(int num_iters);
    int cut_boundaries_until_stable(void);
    FloatImage::Ptr depthmap_bilateral_filter(FloatImage::ConstPtr dm,
        FloatImage::ConstPtr ci, float sigma = 5, int kernel_size = 5);
    void fit_lighting(void);

    // the surface lives in the device context between the batches
    void surface_on_device(void);
    int current_scale(void) const;
    Surface::Ptr download_surface(void) const;

    void set_scale_everywhere(int scale);
    void upload_images(void);
    void upload_surface(void);
    void check(int status, char const* what) const;
    void dump_state(int iter, char const* tag) const;
    // Options::debug_lvl >= 2 (lib/depth_optimizer.h:150-160, depth_optimizer.cc:
    // 44-45, 68-70, 119-127, 139-156): the intermediate embeddings
    void write_debug_depth(std::string const& postfix = "");
    void write_debug_shading(bool with_albedo);

private:
    Options const& opts;
    Bundle::ConstPtr bundle;
    StereoView::Ptr main_view;
    std::vector<StereoView::Ptr> const& sub_views;
    std::vector<double> Mi, ti;          // 9 / 3 per neighbour
    FloatImage::ConstPtr sgm_depth;
    // optimize() keeps the surface in the device context (device_surface):
    // `geom` mirrors its geometry, `valid_patches` its number of non-null
    // patches; `surface` is only set by the constructor that takes one (and by
    // the SMVS_HOST_SURGERY debugging path)
    Surface::Ptr surface;
    bool device_surface = false;
    bool host_surgery = false;
    smvs_surface_geometry geom = { 0, 0, 0, 0, 0, 0 };
    int valid_patches = 0;
    std::vector<uint32_t> subsurfaces;   // bit j: neighbour j sees the patch
    // what the device context holds (upload_surface skips a repeat)
    Surface const* uploaded_surface = nullptr;
    unsigned long uploaded_rev = 0, uploaded_subs_rev = 0, subs_rev = 1;
    bool lit = false;
    bool images_uploaded = false;
    double lighting[16];
    smvs_ctx* ctx = nullptr;
    std::vector<IterationLog> log;
};

} // namespace smvs_amd
