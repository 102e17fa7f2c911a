// MOCK of the reference types the INTEGRATION.md binding touches -- test
// infrastructure only (tests/test_integration_stub.py compiles the code block
// of INTEGRATION.md section 2 against it).  Names and signatures follow
// flanggut/smvs: lib/depth_optimizer.h:27-122, lib/surface.h:24-120,
// lib/bicubic_patch.h:20-45, lib/stereo_view.h:23-60, lib/global_lighting.h:20-40
// and MVE's math::Vector / mve::Image; no behaviour, declarations only.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

namespace math {
template <typename T, int N> struct Vector {
    T v[N];
    T* begin(void) { return v; }
    T const* begin(void) const { return v; }
    T const* end(void) const { return v + N; }
    T& operator[](int i) { return v[i]; }
};
template <typename T, int N, int M> struct Matrix {
    T m[N * M];
    T* begin(void) { return m; }
};
typedef Vector<double, 3> Vec3d;
typedef Matrix<double, 3, 3> Matrix3d;
}
namespace mve {
struct FloatImage {
    typedef std::shared_ptr<FloatImage> Ptr;
    typedef std::shared_ptr<FloatImage const> ConstPtr;
    float const* begin(void) const;   // interleaved channels, row-major
    int width(void) const;
    int height(void) const;
};
}
namespace smvs {
struct BicubicPatch { struct Node { typedef std::shared_ptr<Node> Ptr; double f, dx, dy, dxy; }; };
struct SurfacePatch { typedef std::shared_ptr<SurfacePatch> Ptr; };
class Surface {
public:
    typedef std::shared_ptr<Surface> Ptr;
    typedef std::vector<BicubicPatch::Node::Ptr> NodeList;
    typedef std::vector<SurfacePatch::Ptr> PatchList;
    int get_scale(void) const;
    PatchList const& get_patches(void) const;
    NodeList const& get_nodes(void) const;
    void update_nodes(std::vector<double> const& delta, std::vector<double>* depth_updates);
    // four one-line getters a maintainer adds for the private grid members
    // (surface.h:106-110: num_patches_x / _y, pixel_start_x / _y)
    int get_num_patches_x(void) const;
    int get_num_patches_y(void) const;
    int get_pixel_start_x(void) const;
    int get_pixel_start_y(void) const;
};
class StereoView {
public:
    typedef std::shared_ptr<StereoView> Ptr;
    int get_width(void) const;
    int get_height(void) const;
    float get_flen(void) const;
    float get_inverse_flen(void) const;
    mve::FloatImage::ConstPtr get_image_gradients(void) const;
    mve::FloatImage::ConstPtr get_image_hessian(void) const;
    mve::FloatImage::ConstPtr get_shading_image(void) const;
    mve::FloatImage::ConstPtr get_shading_gradients(void) const;
};
class GlobalLighting {
public:
    typedef std::shared_ptr<GlobalLighting> Ptr;
    typedef math::Vector<double, 16> Params;
    Params const& get_parameters(void);
};
}
struct smvs_ctx;
namespace smvs {
class DepthOptimizer {
public:
    struct Options {
        double regularization, light_surf_regularization;
        int num_iterations, min_scale, debug_lvl;
        bool use_shading, use_sgm, full_optimization;
    };
    DepthOptimizer(StereoView::Ptr main_view, std::vector<StereoView::Ptr> const& sub_views,
        Surface::Ptr surface, Options const& options);
    // added by the binding:
    void setup_hip(int device);
    void upload_scale_planes_hip(void);
    void run_newton_loop_hip(void);
    void run_newton_iterations_hip(int num_iters);
private:
    Options const& opts;
    StereoView::Ptr main_view;
    std::vector<StereoView::Ptr> const& sub_views;
    std::vector<math::Matrix3d> Mi;
    std::vector<math::Vec3d> ti;
    Surface::Ptr surface;
    std::vector<std::vector<std::size_t>> subsurfaces;
    GlobalLighting::Ptr lighting;
    std::vector<double> depths;
    smvs_ctx* ctx;     // added by the binding
    int valid_patches_hip;   // added by the binding: non-null patches of the device surface
    float invproj_hip[9];    // added by the binding: CameraInfo::fill_inverse_calibration
};
}
