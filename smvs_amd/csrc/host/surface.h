// Host mirror of smvs::Surface (reference: lib/surface.h:25-121): the grid of
// bicubic Hermite patches.  Struct-of-arrays instead of the reference's
// shared_ptr graph: `nodes` is exactly the array the device context takes
// (smvs_ctx_set_surface), null nodes / patches are validity bytes.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "image.h"
#include "stereo_view.h"

namespace smvs_amd {

// Cubic Hermite evaluation of one patch (lib/bicubic_patch.cc:56-198,
// evaluated in the Hermite basis instead of monomial coefficients).
struct PatchEval
{
    // nodes16 = {n00, n10, n01, n11} x {f, dx, dy, dxy}
    explicit PatchEval(double const* nodes16);
    // kinds: f, dx, dy, dxy, dxx, dyy at (x, y) in [0,1]^2 (patch units)
    double f(double x, double y) const { return eval(x, y, 0, 0); }
    double dx(double x, double y) const { return eval(x, y, 1, 0); }
    double dy(double x, double y) const { return eval(x, y, 0, 1); }
    double dxy(double x, double y) const { return eval(x, y, 1, 1); }
    double dxx(double x, double y) const { return eval(x, y, 2, 0); }
    double dyy(double x, double y) const { return eval(x, y, 0, 2); }
    double eval(double x, double y, int kx, int ky) const;
    double n[16];
};

class Surface
{
public:
    typedef std::shared_ptr<Surface> Ptr;

    // lib/surface.h:36-37
    static Ptr create(Bundle::ConstPtr bundle, StereoView::Ptr main_view,
        int scale, FloatImage::ConstPtr init_depth = nullptr);

    // A surface from its flat arrays (what smvs_surface_download returns): the
    // host view of a surface that lives in a device context.
    static Ptr from_arrays(int pixel_width, int pixel_height, int scale, int npx,
        int npy, int start_x, int start_y, std::vector<double> const& nodes,
        std::vector<uint8_t> const& node_valid,
        std::vector<uint8_t> const& patch_valid);
    // initialize_depth_from_bundle (lib/surface.cc:90-130) as a list: the
    // features seen by view_id projected into the image -> pixel index
    // (y * width + x) and depth, at most one entry per pixel (where the
    // reference's loop writes a pixel twice, the later feature's depth).
    static void project_bundle(Bundle::ConstPtr bundle, CameraInfo const& cam,
        int view_id, int width, int height, std::vector<int32_t>* pixels,
        std::vector<float>* depths);

    FloatImage::Ptr get_depth_map(void) const;
    FloatImage::Ptr get_normal_map(float inv_flen) const;
    int get_scale(void) const { return scale; }
    int get_patchsize(void) const { return patchsize; }
    int get_num_nodes(void) const { return (int)node_valid.size(); }
    int get_num_patches(void) const { return (int)patch_valid.size(); }
    int get_num_patches_x(void) const { return npx; }
    int get_num_patches_y(void) const { return npy; }
    int get_node_stride(void) const { return npx + 1; }
    int get_pixel_start_x(void) const { return start_x; }
    int get_pixel_start_y(void) const { return start_y; }
    int count_valid_patches(void) const;

    /* global operations, lib/surface.h:50-57 */
    int expand(void);
    void subdivide_patches(void);
    void update_nodes(std::vector<double> const& delta);
    void fill_patches_from_depth(void);

    void fill_node_ids_for_patch(std::size_t patch_id,
        std::size_t* node_ids) const;
    void fill_patch_nodes(std::size_t patch_id, double* nodes16) const;
    void patch_origin(std::size_t patch_id, int* px, int* py) const;
    bool node_exists(int idx, int idy) const;
    bool patch_exists(int idx, int idy) const;

    void delete_patch(std::size_t patch_id) { patch_valid[patch_id] = 0; touch(); }
    void remove_isolated_patches(void);
    void remove_nodes_without_patch(void);

    // Counts the changes to nodes / validity: every mutating operation bumps
    // it (handing out the writable node array counts as one), so the device
    // mirror knows when its copy is stale (host/depth_optimizer.cc).
    unsigned long revision(void) const { return rev; }

    // raw arrays (what the device context consumes)
    std::vector<double>& node_values(void) { touch(); return nodes; }
    std::vector<double> const& node_values(void) const { return nodes; }
    std::vector<uint8_t> const& node_validity(void) const { return node_valid; }
    std::vector<uint8_t> const& patch_validity(void) const { return patch_valid; }

private:
    Surface(void) = default;
    void initialize_depth_from_bundle(Bundle::ConstPtr bundle,
        CameraInfo const& cam, int view_id);
    void initialize_node_from_depth(int idx, int idy);
    std::vector<double> window_depths;   // scratch of initialize_node_from_depth
    int fill_holes(void);

private:
    int pixel_width = 0, pixel_height = 0;
    int scale = 0, patchsize = 0, npx = 0, npy = 0, start_x = 0, start_y = 0;
    FloatImage::Ptr depth;
    std::vector<double> nodes;        // 4 per node
    std::vector<uint8_t> node_valid;
    std::vector<uint8_t> patch_valid;
    // (unique over all surfaces of the process: a new surface at a recycled
    // address never looks like the one the device holds)
    void touch(void) { rev = next_revision(); }
    static unsigned long next_revision(void);
    unsigned long rev = next_revision();
};

} // namespace smvs_amd
