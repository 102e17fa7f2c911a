// Host mirror of smvs::GaussNewtonStep (reference: lib/gauss_newton_step.h:25-98)
// with the reference's constructor and construct() signature; the construction
// runs on the GPU through smvs_gn_construct / smvs_gn_download.
// Compatibility surface, see conjugate_gradient.h.
#pragma once

#include <memory>
#include <vector>

#include "conjugate_gradient.h"
#include "stereo_view.h"
#include "surface.h"

struct smvs_ctx;

namespace smvs_amd {

struct Matrix3d { double m[9]; };   // math::Matrix3d, row-major
struct Vec3d { double v[3]; };      // math::Vec3d

// lib/global_lighting.h:20-40: the 16 spherical-harmonics parameters
class GlobalLighting
{
public:
    typedef std::shared_ptr<GlobalLighting> Ptr;
    static Ptr create(double const* params16);
    double const* get_parameters(void) const { return params; }
private:
    double params[16];
};

class GaussNewtonStep
{
public:
    struct Options   // lib/gauss_newton_step.h:28-34
    {
        double regularization = 0.001;
        double light_surf_regularization = 0.0;
        double l1_min_factor = 1e-4;   // (R_FACTOR; fixed at 1e-4 in the kernels)
        int device = 0;                // HIP device (not in the reference)
    };

    typedef ConjugateGradient::Vector DenseVector;   // SSEVector
    typedef BlockStencilMatrix SparseMatrix;         // BlockSparseMatrix<4>

public:
    // lib/gauss_newton_step.h:40-44.  The planes of the views' current scale
    // (StereoView::set_scale) are read at every construct().
    GaussNewtonStep(Options const& opts, StereoView::ConstPtr main_view,
        std::vector<StereoView::Ptr> const& sub_views,
        std::vector<Matrix3d> const& Mi, std::vector<Vec3d> const& ti);
    ~GaussNewtonStep(void);
    GaussNewtonStep(GaussNewtonStep const&) = delete;
    GaussNewtonStep& operator=(GaussNewtonStep const&) = delete;

    // lib/gauss_newton_step.h:46-50; any output may be null.
    void construct(Surface::Ptr surface,
        std::vector<std::vector<std::size_t>> const& subsurfaces,
        std::vector<char> const& active_nodes, GlobalLighting::Ptr lighting,
        SparseMatrix* hessian, DenseVector* gradient, SparseMatrix* precond);

private:
    void upload_planes(bool with_shading);

private:
    // (copies: the reference keeps references, lib/gauss_newton_step.h:83-90,
    // which dangle when a caller passes temporaries)
    Options const opts;
    StereoView::ConstPtr main_view;
    std::vector<StereoView::Ptr> const sub_views;
    std::vector<Matrix3d> const Mi;
    std::vector<Vec3d> const ti;
    smvs_ctx* ctx = nullptr;
    // The planes the device holds, re-uploaded when a view's scale changed.
    // StereoView::set_scale / set_scale_planes install NEW image objects; the
    // shared pointers kept here keep the uploaded ones alive, so an address
    // cannot be recycled by a later image and compare equal by accident.
    // (Planes are immutable once a view hands them out; writing into one in
    // place is not detected.)
    std::vector<FloatImage::ConstPtr> uploaded;
};

} // namespace smvs_amd
