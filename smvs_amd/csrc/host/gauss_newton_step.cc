#include "gauss_newton_step.h"

#include <algorithm>
#include <stdexcept>
#include <string>

#include "../../../include/smvs_hip.h"

namespace smvs_amd {

GlobalLighting::Ptr
GlobalLighting::create(double const* params16)
{
    Ptr l(new GlobalLighting());
    std::copy(params16, params16 + 16, l->params);
    return l;
}

namespace {
void
check(int rc, char const* what)
{
    if (rc != SMVS_OK)
        throw std::runtime_error(std::string(what) + ": " + smvs_last_error());
}
}

GaussNewtonStep::GaussNewtonStep(Options const& opts,
    StereoView::ConstPtr main_view, std::vector<StereoView::Ptr> const& sub_views,
    std::vector<Matrix3d> const& Mi, std::vector<Vec3d> const& ti)
    : opts(opts), main_view(main_view), sub_views(sub_views), Mi(Mi), ti(ti)
{
    if (main_view == nullptr || sub_views.empty() || Mi.size() != sub_views.size()
        || ti.size() != sub_views.size())
        throw std::invalid_argument("GaussNewtonStep: views and reprojections");
    check(smvs_ctx_create(opts.device, main_view->get_width(),
        main_view->get_height(), (int)sub_views.size(), &ctx), "smvs_ctx_create");
    std::vector<double> M(9 * Mi.size()), t(3 * ti.size());
    for (std::size_t i = 0; i < Mi.size(); ++i) {
        std::copy(Mi[i].m, Mi[i].m + 9, M.begin() + 9 * i);
        std::copy(ti[i].v, ti[i].v + 3, t.begin() + 3 * i);
    }
    int const rc = smvs_ctx_set_cameras(ctx, M.data(), t.data(),
        main_view->get_flen(), main_view->get_inverse_flen());
    if (rc != SMVS_OK) {
        smvs_ctx_destroy(ctx);
        ctx = nullptr;
        check(rc, "smvs_ctx_set_cameras");
    }
    uploaded.assign(sub_views.size() + 2, nullptr);
}

GaussNewtonStep::~GaussNewtonStep(void)
{
    if (ctx != nullptr)
        smvs_ctx_destroy(ctx);
}

void
GaussNewtonStep::upload_planes(bool with_shading)
{
    // main_gradients / main_gradients_linear, lib/gauss_newton_step.cc:24-31;
    // sub gradients and Hessians, :176-180
    FloatImage::ConstPtr grad = main_view->get_image_gradients();
    if (grad == nullptr)
        throw std::invalid_argument("GaussNewtonStep: the main view has no "
            "gradient planes (StereoView::set_scale)");
    FloatImage::ConstPtr sh = with_shading ? main_view->get_shading_image() : nullptr;
    FloatImage::ConstPtr shg = with_shading ? main_view->get_shading_gradients() : nullptr;
    if (with_shading && (sh == nullptr || shg == nullptr))
        throw std::invalid_argument("GaussNewtonStep: lighting without a "
            "shading image (StereoView::create(..., initialize_linear))");
    if (uploaded[0] != grad || (with_shading && uploaded[1] != sh)) {
        check(smvs_ctx_upload_main(ctx, grad->begin(),
            with_shading ? sh->begin() : nullptr,
            with_shading ? shg->begin() : nullptr), "smvs_ctx_upload_main");
        uploaded[0] = grad;
        if (with_shading)
            uploaded[1] = sh;
    }
    for (std::size_t j = 0; j < sub_views.size(); ++j) {
        FloatImage::ConstPtr g = sub_views[j]->get_image_gradients();
        FloatImage::ConstPtr h = sub_views[j]->get_image_hessian();
        if (g == nullptr || h == nullptr)
            throw std::invalid_argument("GaussNewtonStep: a neighbour has no "
                "gradient / Hessian planes (StereoView::set_scale)");
        if (uploaded[2 + j] == g)
            continue;
        check(smvs_ctx_upload_sub(ctx, (int)j, g->width(), g->height(), g->begin(),
            h->begin()), "smvs_ctx_upload_sub");
        uploaded[2 + j] = g;
    }
}

void
GaussNewtonStep::construct(Surface::Ptr surface,
    std::vector<std::vector<std::size_t>> const& subsurfaces,
    std::vector<char> const& active_nodes, GlobalLighting::Ptr lighting,
    SparseMatrix* hessian, DenseVector* gradient, SparseMatrix* precond)
{
    if (surface == nullptr)
        throw std::invalid_argument("GaussNewtonStep::construct: null surface");
    Surface const& s = *surface;
    std::size_t const N = (std::size_t)s.get_num_nodes();
    std::size_t const P = (std::size_t)s.get_num_patches();
    if (subsurfaces.size() != P || active_nodes.size() != N)
        throw std::invalid_argument("GaussNewtonStep::construct: subsurfaces / "
            "active_nodes do not match the surface");
    this->upload_planes(lighting != nullptr);
    // subsurfaces[patch] -> bit mask (depth_optimizer.h:108)
    std::vector<uint32_t> vis(P, 0);
    for (std::size_t p = 0; p < P; ++p)
        for (std::size_t sub : subsurfaces[p]) {
            if (sub >= sub_views.size())
                throw std::invalid_argument("GaussNewtonStep::construct: "
                    "neighbour index out of range");
            vis[p] |= 1u << sub;
        }
    check(smvs_ctx_set_surface(ctx, s.get_scale(), s.get_num_patches_x(),
        s.get_num_patches_y(), s.get_pixel_start_x(), s.get_pixel_start_y(),
        s.node_values().data(), s.node_validity().data(),
        s.patch_validity().data(), vis.data()), "smvs_ctx_set_surface");
    std::vector<uint8_t> active(N);
    for (std::size_t n = 0; n < N; ++n)
        active[n] = active_nodes[n] ? 1 : 0;
    check(smvs_ctx_set_active(ctx, active.data()), "smvs_ctx_set_active");
    check(smvs_gn_construct(ctx, opts.regularization, opts.light_surf_regularization,
        lighting != nullptr ? lighting->get_parameters() : nullptr, nullptr),
        "smvs_gn_construct");
    std::vector<double> Pdiag;
    if (hessian != nullptr) {
        hessian->num_nodes = N;
        hessian->node_stride = (std::size_t)s.get_node_stride();
        hessian->blocks.assign(N * 9 * 16, 0.0);
    }
    if (gradient != nullptr)
        gradient->assign(4 * N, 0.0);
    if (precond != nullptr)
        Pdiag.assign(N * 16, 0.0);
    check(smvs_gn_download(ctx, hessian != nullptr ? hessian->blocks.data() : nullptr,
        gradient != nullptr ? gradient->data() : nullptr,
        precond != nullptr ? Pdiag.data() : nullptr), "smvs_gn_download");
    if (precond != nullptr) {
        precond->num_nodes = N;
        precond->node_stride = (std::size_t)s.get_node_stride();
        precond->blocks.assign(N * 9 * 16, 0.0);
        for (std::size_t n = 0; n < N; ++n)
            std::copy(Pdiag.begin() + n * 16, Pdiag.begin() + (n + 1) * 16,
                precond->blocks.begin() + (n * 9 + 4) * 16);
    }
}

} // namespace smvs_amd
