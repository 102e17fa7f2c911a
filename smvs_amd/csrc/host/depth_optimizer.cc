// Host mirror of lib/depth_optimizer.cc driving the HIP hot path.
#include "depth_optimizer.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

#include "../../../include/smvs_hip.h"

namespace smvs_amd {

namespace {

// Projection of a main-view pixel with depth w into a neighbour
// (lib/correspondence.cc:20-51, 88-100).
struct Warp
{
    double p, q, r, a, b, d;
    Warp(double const* M, double const* t, double u, double v, double w)
    {
        p = M[0] * u + M[1] * v + M[2];
        q = M[3] * u + M[4] * v + M[5];
        r = M[6] * u + M[7] * v + M[8];
        a = w * p + t[0];
        b = w * q + t[1];
        d = w * r + t[2];
    }
    double x(void) const { return a / d; }
    double y(void) const { return b / d; }
    void jacobian(double const* M, double w, double wx, double wy,
        double* jac) const
    {
        double const d2 = d * d;
        jac[0] = (wx * p + w * M[0]) / d - a * (wx * r + w * M[6]) / d2;
        jac[2] = (wy * p + w * M[1]) / d - a * (wy * r + w * M[7]) / d2;
        jac[1] = (wx * q + w * M[3]) / d - b * (wx * r + w * M[6]) / d2;
        jac[3] = (wy * q + w * M[4]) / d - b * (wy * r + w * M[7]) / d2;
    }
};

// Moore-Penrose solve of the symmetric PSD 16x16 lighting system through a
// cyclic Jacobi eigen decomposition (math::matrix_pseudo_inverse in the
// reference, lib/light_optimizer.cc:50-52 [MVE-unverified]).
void
solve_psd16(double const* A_in, double const* b, double* x)
{
    int const n = 16;
    double A[256], V[256];
    std::copy(A_in, A_in + 256, A);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j)
                off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300)
            break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                double const apq = A[p * n + q];
                if (apq == 0.0)
                    continue;
                double const theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double const tt = (theta >= 0 ? 1.0 : -1.0)
                    / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double const c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < n; ++k) {
                    double const u = A[k * n + p], v = A[k * n + q];
                    A[k * n + p] = c * u - s * v;
                    A[k * n + q] = s * u + c * v;
                }
                for (int k = 0; k < n; ++k) {
                    double const u = A[p * n + k], v = A[q * n + k];
                    A[p * n + k] = c * u - s * v;
                    A[q * n + k] = s * u + c * v;
                }
                for (int k = 0; k < n; ++k) {
                    double const u = V[k * n + p], v = V[k * n + q];
                    V[k * n + p] = c * u - s * v;
                    V[k * n + q] = s * u + c * v;
                }
            }
    }
    double largest = 0.0;
    for (int i = 0; i < n; ++i)
        largest = std::max(largest, std::fabs(A[i * n + i]));
    std::fill(x, x + n, 0.0);
    for (int k = 0; k < n; ++k) {
        double const ev = A[k * n + k];
        if (ev == 0.0 || std::fabs(ev) <= 1e-12 * largest)
            continue;
        double proj = 0.0;
        for (int i = 0; i < n; ++i)
            proj += V[i * n + k] * b[i];
        proj /= ev;
        for (int i = 0; i < n; ++i)
            x[i] += V[i * n + k] * proj;
    }
}

} // namespace

void
DepthOptimizer::check(int status, char const* what) const
{
    if (status != SMVS_OK)
        throw std::runtime_error(std::string(what) + ": " + smvs_last_error());
}

DepthOptimizer::DepthOptimizer(StereoView::Ptr main_view,
    std::vector<StereoView::Ptr> const& sub_views, Bundle::ConstPtr bundle,
    Options const& opts)
    : opts(opts), bundle(bundle), main_view(main_view), sub_views(sub_views)
{
    if (main_view == nullptr || sub_views.empty())
        throw std::invalid_argument("DepthOptimizer: missing views");
    this->prepare_correspondences();
    check(smvs_ctx_create(opts.device, main_view->get_width(),
        main_view->get_height(), (int)sub_views.size(), &this->ctx),
        "smvs_ctx_create");
    check(smvs_ctx_set_cameras(ctx, Mi.data(), ti.data(),
        main_view->get_flen(), main_view->get_inverse_flen()),
        "smvs_ctx_set_cameras");
}

DepthOptimizer::~DepthOptimizer(void)
{
    if (ctx != nullptr)
        smvs_ctx_destroy(ctx);
}

void
DepthOptimizer::prepare_correspondences(void)
{
    // lib/depth_optimizer.cc:679-699
    Mi.resize(9 * sub_views.size());
    ti.resize(3 * sub_views.size());
    for (std::size_t i = 0; i < sub_views.size(); ++i) {
        float M[9], t[3];
        main_view->get_camera().fill_reprojection(sub_views[i]->get_camera(),
            (float)main_view->get_width(), (float)main_view->get_height(),
            (float)sub_views[i]->get_width(), (float)sub_views[i]->get_height(),
            M, t);
        for (int j = 0; j < 9; ++j)
            Mi[9 * i + j] = M[j];
        for (int j = 0; j < 3; ++j)
            ti[3 * i + j] = t[j];
    }
}

FloatImage::Ptr
DepthOptimizer::depthmap_bilateral_filter(FloatImage::ConstPtr dm,
    FloatImage::ConstPtr ci, float sigma, int kernel_size)
{
    FloatImage::Ptr out = FloatImage::create(ci->width(), ci->height(), 1);
    check(smvs_bilateral_upsample(opts.device, dm->begin(), dm->width(),
        dm->height(), ci->begin(), ci->width(), ci->height(), ci->channels(),
        sigma, kernel_size, out->begin()), "smvs_bilateral_upsample");
    return out;
}

void
DepthOptimizer::create_initial_surface(void)
{
    // lib/depth_optimizer.cc:35-51
    int const init_scale = (int)std::max(std::ceil(std::log2(
        main_view->get_width() * main_view->get_height() / 1.7e6) / 2) + 4, 4.0);
    if (opts.use_sgm) {
        FloatImage::Ptr init = main_view->get_sgm_depth();
        if (init == nullptr)
            throw std::invalid_argument("use_sgm without an smvs-sgm embedding");
        init = depthmap_bilateral_filter(init, main_view->get_image());
        this->surface = Surface::create(bundle, main_view, init_scale, init);
        this->sgm_depth = init;
    } else
        this->surface = Surface::create(bundle, main_view, init_scale + 1);
}

void
DepthOptimizer::set_scale_everywhere(int scale)
{
    // StereoView::set_scale for the main view and every neighbour
    // (lib/depth_optimizer.cc:63-66, 99-103) on the device; the planes come
    // back once per scale for the host-side topology code.
    if (!images_uploaded) {
        ByteImage::ConstPtr mb = main_view->get_raw_bytes();
        check(smvs_ctx_upload_image(ctx, -1, mb->width(), mb->height(),
            mb->channels(), mb->begin()), "smvs_ctx_upload_image");
        for (std::size_t j = 0; j < sub_views.size(); ++j) {
            ByteImage::ConstPtr sb = sub_views[j]->get_raw_bytes();
            check(smvs_ctx_upload_image(ctx, (int)j, sb->width(), sb->height(),
                sb->channels(), sb->begin()), "smvs_ctx_upload_image");
        }
        if (opts.use_shading)
            check(smvs_ctx_upload_shading(ctx,
                main_view->get_shading_image()->begin(),
                main_view->get_shading_gradients()->begin()),
                "smvs_ctx_upload_shading");
        images_uploaded = true;
    }
    check(smvs_ctx_set_scale(ctx, scale), "smvs_ctx_set_scale");
    {
        FloatImage::Ptr g = FloatImage::create(main_view->get_width(),
            main_view->get_height(), 2);
        check(smvs_ctx_download_planes(ctx, -1, g->begin(), nullptr),
            "smvs_ctx_download_planes");
        main_view->set_scale_planes(g, nullptr);
    }
    for (std::size_t j = 0; j < sub_views.size(); ++j) {
        FloatImage::Ptr g = FloatImage::create(sub_views[j]->get_width(),
            sub_views[j]->get_height(), 2);
        FloatImage::Ptr hs = FloatImage::create(sub_views[j]->get_width(),
            sub_views[j]->get_height(), 3);
        check(smvs_ctx_download_planes(ctx, (int)j, g->begin(), hs->begin()),
            "smvs_ctx_download_planes");
        sub_views[j]->set_scale_planes(g, hs);
    }
}

void
DepthOptimizer::upload_surface(void)
{
    check(smvs_ctx_set_surface(ctx, surface->get_scale(),
        surface->get_num_patches_x(), surface->get_num_patches_y(),
        surface->get_pixel_start_x(), surface->get_pixel_start_y(),
        surface->node_values().data(), surface->node_validity().data(),
        surface->patch_validity().data(), subsurfaces.data()),
        "smvs_ctx_set_surface");
}

void
DepthOptimizer::fit_lighting(void)
{
    // LightOptimizer::fit_lighting_to_image, lib/light_optimizer.cc:22-55
    subsurfaces.resize(surface->get_num_patches(), 0);
    upload_surface();
    double A[256], b[16];
    check(smvs_light_accumulate(ctx, A, b), "smvs_light_accumulate");
    solve_psd16(A, b, this->lighting);
    this->lit = true;
}

void
DepthOptimizer::optimize(void)
{
    // lib/depth_optimizer.cc:53-162
    this->create_initial_surface();
    this->set_scale_everywhere(surface->get_scale());
    this->run_newton_iterations(opts.num_iterations);

    while (surface->get_scale() > opts.min_scale && surface->get_scale() > 0) {
        surface->subdivide_patches();
        this->set_scale_everywhere(surface->get_scale());
        surface->fill_patches_from_depth();
        if (opts.use_shading && surface->get_scale() < 4)
            this->fit_lighting();
        this->run_newton_iterations(opts.num_iterations);
    }
    main_view->write_depth_to_view(surface->get_depth_map(), opts.output_name);
    main_view->write_image_to_view(this->get_normals(), opts.output_name + "N");
}

FloatImage::Ptr
DepthOptimizer::get_depth(void)
{
    return surface->get_depth_map();
}

FloatImage::Ptr
DepthOptimizer::get_normals(void)
{
    return surface->get_normal_map(main_view->get_inverse_flen());
}

void
DepthOptimizer::run_newton_iterations(int num_iters)
{
    // lib/depth_optimizer.cc:164-358
    bool finished = false;
    for (int iter = 0; iter < num_iters; ++iter) {
        int const num_valid_patches = surface->count_valid_patches();
        if (iter == 0) {
            this->create_subview_surfaces();
            int deleted = std::numeric_limits<int>::max();
            while (deleted > 10)
                deleted = this->cut_boundaries();
        }

        // the whole Newton loop (:204-304) on the device
        upload_surface();
        smvs_gn_loop_params prm;
        prm.regularization = opts.regularization;
        prm.light_surf_regularization = opts.light_surf_regularization;
        prm.full_optimization = opts.full_optimization ? 1 : 0;
        prm.max_newton_steps = 200;
        prm.cg_max_iterations = 200;
        prm.cg_q_tolerance = 1e-3;
        prm.active_threshold = 0.15;
        prm.full_opt_threshold = 0.01;
        prm.use_lighting = lit ? 1 : 0;
        std::copy(lighting, lighting + 16, prm.lighting);
        prm.reset_active = 1;
        smvs_gn_loop_stats stats;
        check(smvs_gn_run_loop(ctx, &prm, &stats), "smvs_gn_run_loop");
        check(smvs_get_nodes(ctx, surface->node_values().data()),
            "smvs_get_nodes");
        log.push_back({ surface->get_scale(), iter, stats.newton_steps,
            num_valid_patches, stats.linear_iterations });

        if (finished)
            break;
        int deleted = std::numeric_limits<int>::max();
        while (deleted > 10)
            deleted = this->cut_boundaries();
        if (!opts.use_sgm) {
            surface->expand();
            this->create_subview_surfaces();
            deleted = std::numeric_limits<int>::max();
            while (deleted > 10)
                deleted = this->cut_boundaries();
        }
        surface->remove_isolated_patches();

        int const num_valid_new = surface->count_valid_patches();
        double const change = 1.0
            - (double)std::min(num_valid_new, num_valid_patches)
            / (double)std::max(num_valid_new, num_valid_patches);
        if (iter > 0 && (num_valid_new <= num_valid_patches
            || change < 0.05 * surface->get_scale()))
            finished = true;
    }
}

double
DepthOptimizer::mse_for_patch(std::size_t patch_id)
{
    // lib/depth_optimizer.cc:747-790
    int const ps = surface->get_patchsize();
    double n16[16];
    surface->fill_patch_nodes(patch_id, n16);
    PatchEval pe(n16);
    int px, py;
    surface->patch_origin(patch_id, &px, &py);
    FloatImage::ConstPtr main_grad = main_view->get_image_gradients();
    double error = 0.0, counter = 0.0;
    for (int j = 0; j < ps; ++j)
        for (int i = 0; i < ps; ++i) {
            double const u = (i + 0.5) / ps, v = (j + 0.5) / ps;
            double const w = pe.f(u, v), wx = pe.dx(u, v) / ps,
                wy = pe.dy(u, v) / ps;
            double const gm0 = main_grad->at(px + i, py + j, 0);
            double const gm1 = main_grad->at(px + i, py + j, 1);
            for (std::size_t s = 0; s < sub_views.size(); ++s) {
                if (!(subsurfaces[patch_id] & (1u << s)))
                    continue;
                FloatImage::ConstPtr sg = sub_views[s]->get_image_gradients();
                Warp wp(&Mi[9 * s], &ti[3 * s], px + i + 0.5, py + j + 0.5, w);
                double jac[4];
                wp.jacobian(&Mi[9 * s], w, wx, wy, jac);
                float const qx = (float)(wp.x() - 0.5), qy = (float)(wp.y() - 0.5);
                double const g0 = sg->linear_at(qx, qy, 0);
                double const g1 = sg->linear_at(qx, qy, 1);
                double const d0 = gm0 - (jac[0] * g0 + jac[1] * g1);
                double const d1 = gm1 - (jac[2] * g0 + jac[3] * g1);
                error += std::sqrt(d0 * d0 + d1 * d1);
                counter += 1.0;
            }
        }
    return counter == 0.0 ? 1.0 : error / counter;
}

double
DepthOptimizer::ncc_for_patch(std::size_t patch_id, std::size_t sub_id)
{
    // lib/depth_optimizer.cc:792-912
    FloatImage::ConstPtr main_image = main_view->get_image();
    FloatImage::ConstPtr sub_image = sub_views[sub_id]->get_image();
    int const ps = surface->get_patchsize();
    double n16[16];
    surface->fill_patch_nodes(patch_id, n16);
    PatchEval pe(n16);
    int px, py;
    surface->patch_origin(patch_id, &px, &py);
    struct Sample { double x, y, depth; };
    std::vector<Sample> samples;
    samples.reserve((size_t)ps * ps + 8 * ps + 16);
    for (int j = 0; j < ps; ++j)
        for (int i = 0; i < ps; ++i)
            samples.push_back({ (double)(px + i), (double)(py + j),
                pe.f((i + 0.5) / ps, (j + 0.5) / ps) });
    double const min_x = px, min_y = py, max_x = px + ps, max_y = py + ps;
    if (min_x > 1 && max_x < main_image->width() - 2 && min_y > 1
        && max_y < main_image->height() - 2) {
        samples.push_back({ min_x - 1, min_y - 1, n16[0] });
        samples.push_back({ max_x + 1, min_y - 1, n16[4] });
        samples.push_back({ min_x - 1, max_y + 1, n16[8] });
        samples.push_back({ max_x + 1, max_y + 1, n16[12] });
    }
    // the list grows while it is walked (:823-857)
    for (std::size_t i = 0; i < samples.size(); ++i) {
        Sample const s = samples[i];
        if (min_y > 2 && s.y == min_y) {
            samples.push_back({ s.x, s.y - 2, s.depth });
            samples.push_back({ s.x, s.y - 1, s.depth });
        }
        if (max_y < main_image->height() - 3 && s.y == max_y) {
            samples.push_back({ s.x, s.y + 2, s.depth });
            samples.push_back({ s.x, s.y + 1, s.depth });
        }
        if (min_x > 2 && s.x == min_x) {
            samples.push_back({ s.x - 2, s.y, s.depth });
            samples.push_back({ s.x - 1, s.y, s.depth });
        }
        if (max_x < main_image->width() - 3 && s.x == max_x) {
            samples.push_back({ s.x + 2, s.y, s.depth });
            samples.push_back({ s.x + 1, s.y, s.depth });
        }
    }
    std::size_t const n = samples.size();
    std::vector<double> v0(3 * n), v1(3 * n);
    double mean0[3] = { 0, 0, 0 }, mean1[3] = { 0, 0, 0 }, cnt[3] = { 0, 0, 0 };
    int const mc = main_image->channels(), sc = sub_image->channels();
    for (std::size_t i = 0; i < n; ++i) {
        Warp wp(&Mi[9 * sub_id], &ti[3 * sub_id], samples[i].x + 0.5,
            samples[i].y + 0.5, samples[i].depth);
        double const qx = wp.x() - 0.5, qy = wp.y() - 0.5;
        if (qx < 1 || qx > sub_image->width() - 2 || qy < 1
            || qy > sub_image->height() - 2)
            return -1;
        for (int c = 0; c < 3; ++c) {
            double const cm = main_image->at((int64_t)samples[i].x,
                (int64_t)samples[i].y, std::min(c, mc - 1));
            double const cs = sub_image->linear_at((float)qx, (float)qy,
                std::min(c, sc - 1));
            cnt[c] += 1.0;
            mean0[c] += (cm - mean0[c]) / cnt[c];
            mean1[c] += (cs - mean1[c]) / cnt[c];
            v0[3 * i + c] = cm;
            v1[3 * i + c] = cs;
        }
    }
    double n0 = 0.0, n1 = 0.0, dot = 0.0;
    for (std::size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            double const a = v0[3 * i + c] - mean0[c];
            double const b = v1[3 * i + c] - mean1[c];
            n0 += a * a;
            n1 += b * b;
            dot += a * b;
        }
    n0 = std::sqrt(n0);
    n1 = std::sqrt(n1);
    if (n0 + n1 < 0.001 * n)
        return 1;
    return dot / (n0 * n1);
}

int
DepthOptimizer::cut_boundaries(void)
{
    // lib/depth_optimizer.cc:360-431
    int deleted = 0;
    int const ps = surface->get_patchsize();
    float invproj[9];
    main_view->get_camera().fill_inverse_calibration(invproj,
        (float)main_view->get_width(), (float)main_view->get_height());
    std::size_t const num_patches = surface->get_num_patches();

    // depth discontinuities
    for (std::size_t p = 0; p < num_patches; ++p) {
        if (!surface->patch_validity()[p])
            continue;
        double n16[16];
        surface->fill_patch_nodes(p, n16);
        double const f[4] = { n16[0], n16[4], n16[8], n16[12] };
        int lo = 0, hi = 0;  // first minimum, last maximum (multimap order)
        for (int i = 1; i < 4; ++i) {
            if (f[i] < f[lo])
                lo = i;
            if (f[i] >= f[hi])
                hi = i;
        }
        double dd_factor = 5.0;
        if (lo + hi == 3)
            dd_factor *= 1.41421356237309504880;
        int px, py;
        surface->patch_origin(p, &px, &py);
        float const fx = (float)px + 0.5f, fy = (float)py + 0.5f;
        float v[3];
        for (int r = 0; r < 3; ++r)
            v[r] = invproj[3 * r] * fx + invproj[3 * r + 1] * fy
                + invproj[3 * r + 2] * 1.0f;
        float const vnorm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        double const threshold = dd_factor * f[lo] * invproj[0] * ps / vnorm;
        if (f[hi] - f[lo] > threshold) {
            surface->delete_patch(p);
            deleted += 1;
        }
    }
    // high-error patches on the border of the surface
    int const stride = surface->get_node_stride();
    for (std::size_t p = 0; p < num_patches; ++p) {
        if (!surface->patch_validity()[p])
            continue;
        std::size_t ids[4];
        surface->fill_node_ids_for_patch(p, ids);
        double const error = this->mse_for_patch(p);
        for (int k = 0; k < 4; ++k) {
            int const nx = (int)(ids[k] % stride), ny = (int)(ids[k] / stride);
            int missing = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    if ((dx || dy) && !surface->node_exists(nx + dx, ny + dy))
                        missing += 1;
            if (missing > 1 && error > 0.05) {
                surface->delete_patch(p);
                deleted += 1;
                break;
            }
        }
    }
    surface->remove_nodes_without_patch();
    return deleted;
}

void
DepthOptimizer::create_subview_surfaces(void)
{
    // lib/depth_optimizer.cc:433-604
    std::size_t const num_patches = surface->get_num_patches();
    subsurfaces.assign(num_patches, 0);
    std::size_t const S = sub_views.size();

    // z-buffer of the current surface (and the SGM depth) in every neighbour
    std::vector<FloatImage::Ptr> zbuf(S);
    for (std::size_t s = 0; s < S; ++s) {
        zbuf[s] = FloatImage::create(sub_views[s]->get_width() + 1,
            sub_views[s]->get_height() + 1, 1);
        zbuf[s]->fill(10000.0f);
    }
    FloatImage::Ptr depth = surface->get_depth_map();
    auto splat = [&](int x, int y, double w) {
        for (std::size_t s = 0; s < S; ++s) {
            Warp wp(&Mi[9 * s], &ti[3 * s], x + 0.5, y + 0.5, w);
            double const qx = wp.x() - 0.5, qy = wp.y() - 0.5;
            double const cutoffset = 3.0;
            if (qx < cutoffset || qx >= sub_views[s]->get_width() - cutoffset
                || qy < cutoffset || qy >= sub_views[s]->get_height() - cutoffset)
                continue;
            int const cx = (int)qx, cy = (int)qy;
            for (int dx = -1; dx < 2; ++dx)
                for (int dy = -1; dy < 2; ++dy)
                    if (wp.d < zbuf[s]->at(cx + dx, cy + dy, 0))
                        zbuf[s]->at(cx + dx, cy + dy, 0) = (float)wp.d;
        }
    };
    for (int x = 0; x < depth->width(); ++x)
        for (int y = 0; y < depth->height(); ++y) {
            if (depth->at(x, y, 0) != 0)
                splat(x, y, depth->at(x, y, 0));
            if (opts.use_sgm && sgm_depth->at(x, y, 0) != 0)
                splat(x, y, sgm_depth->at(x, y, 0));
        }

    int const ps = surface->get_patchsize();
    std::vector<double> w(ps * ps), wx(ps * ps), wy(ps * ps);
    for (std::size_t p = 0; p < num_patches; ++p) {
        if (!surface->patch_validity()[p])
            continue;
        double n16[16];
        surface->fill_patch_nodes(p, n16);
        PatchEval pe(n16);
        int px, py;
        surface->patch_origin(p, &px, &py);
        for (int j = 0; j < ps; ++j)
            for (int i = 0; i < ps; ++i) {
                double const u = (i + 0.5) / ps, v = (j + 0.5) / ps;
                w[j * ps + i] = pe.f(u, v);
                wx[j * ps + i] = pe.dx(u, v) / ps;
                wy[j * ps + i] = pe.dy(u, v) / ps;
            }
        for (std::size_t s = 0; s < S; ++s) {
            double const sw = sub_views[s]->get_width();
            double const sh = sub_views[s]->get_height();
            double const cutoffset = 0.03 * std::max(sw, sh);
            bool visible = true;
            for (int k = 0; k < ps * ps && visible; ++k) {
                Warp wp(&Mi[9 * s], &ti[3 * s], px + k % ps + 0.5,
                    py + k / ps + 0.5, w[k]);
                double const qx = wp.x() - 0.5, qy = wp.y() - 0.5;
                if (qx < cutoffset || qx >= sw - cutoffset || qy < cutoffset
                    || qy >= sh - cutoffset) {
                    visible = false;
                    break;
                }
                int const cx = (int)qx, cy = (int)qy;
                for (int dx = -1; dx < 2; ++dx)
                    for (int dy = -1; dy < 2; ++dy)
                        if (wp.d * 0.95 > zbuf[s]->at(cx + dx, cy + dy, 0))
                            visible = false;
            }
            if (!visible)
                continue;
            // anisotropy of the warp: ratio of squared singular values
            double worst = 0.0;
            for (int k = 0; k < ps * ps; ++k) {
                Warp wp(&Mi[9 * s], &ti[3 * s], px + k % ps + 0.5,
                    py + k / ps + 0.5, w[k]);
                double jac[4];
                wp.jacobian(&Mi[9 * s], w[k], wx[k], wy[k], jac);
                double const e = std::sqrt((jac[0] - jac[3]) * (jac[0] - jac[3])
                    + (jac[1] + jac[2]) * (jac[1] + jac[2]));
                double const g = std::sqrt((jac[0] + jac[3]) * (jac[0] + jac[3])
                    + (jac[1] - jac[2]) * (jac[1] - jac[2]));
                double const s0 = (e + g) / 2.0;
                double const s1 = std::fabs(s0 - e);
                double const hi = std::max(s0, s1), lo = std::min(s0, s1);
                worst = std::max(worst, (hi * hi) / (lo * lo));
            }
            if (worst > 8.0)
                continue;
            if (!opts.use_sgm && this->ncc_for_patch(p, s) < 0)
                continue;
            subsurfaces[p] |= (1u << s);
        }
    }
    std::size_t removed = 0;
    for (std::size_t p = 0; p < num_patches; ++p)
        if (surface->patch_validity()[p] && subsurfaces[p] == 0) {
            surface->delete_patch(p);
            removed += 1;
        }
    if (removed > 0)
        surface->remove_nodes_without_patch();
}

} // namespace smvs_amd
