// Host mirror of lib/depth_optimizer.cc driving the HIP hot path.
#include "depth_optimizer.h"

#include <iostream>
#include "topo_math.h"
#include <string>
#include <map>
#include <mutex>
#include <cstdlib>
#include <cstdio>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

#include "../../../include/smvs_hip.h"

// Wall-clock breakdown of optimize() on stderr when SMVS_HOST_TIMING is set
// (where the time between Newton batches goes: SURVEY 8(f)-2).
namespace {
struct HostTimers {
    bool on = std::getenv("SMVS_HOST_TIMING") != nullptr;
    std::map<std::string, double> acc;
    std::mutex lock;   // optimizers of several views may run in threads
    ~HostTimers() { report(); }
    void report()
    {
        std::lock_guard<std::mutex> guard(lock);
        if (!on || acc.empty())
            return;
        double total = 0.0;
        for (auto const& kv : acc)
            total += kv.second;
        for (auto const& kv : acc)
            std::fprintf(stderr, "[smvs host] %-28s %9.2f ms (%4.1f%%)\n",
                kv.first.c_str(), 1e3 * kv.second, 100.0 * kv.second / total);
        acc.clear();
    }
};
HostTimers g_timers;
struct ScopedHostTimer {
    char const* name;
    std::chrono::steady_clock::time_point t0;
    explicit ScopedHostTimer(char const* n) : name(n),
        t0(std::chrono::steady_clock::now()) {}
    ~ScopedHostTimer()
    {
        if (g_timers.on) {
            std::lock_guard<std::mutex> guard(g_timers.lock);
            g_timers.acc[name] += std::chrono::duration<double>(
                std::chrono::steady_clock::now() - t0).count();
        }
    }
};
}

namespace smvs_amd {

namespace {

// Projection of a main-view pixel with depth w into a neighbour
// (lib/correspondence.cc:20-51, 88-100).
typedef smvs_topo::Warp Warp;

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

// Debug aid: with SMVS_DUMP_DIR set, the surface state at the stages of a
// Newton batch goes to <dir>/s<scale>_i<iter>_<tag>.bin in the layout of the
// oracle's ORC_DUMP_DIR files (tools/compare_dumps.py).
void
DepthOptimizer::dump_state(int iter, char const* tag) const
{
    char const* dir = std::getenv("SMVS_DUMP_DIR");
    if (dir == nullptr)
        return;
    Surface::Ptr const surf = this->download_surface();
    std::string const path = std::string(dir) + "/s"
        + std::to_string(surf->get_scale()) + "_i" + std::to_string(iter)
        + "_" + tag + ".bin";
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (f == nullptr)
        return;
    int const hdr[3] = { surf->get_scale(), surf->get_num_patches_x(),
        surf->get_num_patches_y() };
    std::size_t const nn = surf->get_num_nodes();
    std::size_t const np = surf->get_num_patches();
    std::vector<uint32_t> vis(np, 0);
    if (device_surface)
        check(smvs_surface_download(ctx, nullptr, nullptr, nullptr, vis.data()),
            "smvs_surface_download");
    else
        for (std::size_t p = 0; p < np && p < subsurfaces.size(); ++p)
            vis[p] = subsurfaces[p];
    std::fwrite(hdr, sizeof(int), 3, f);
    std::fwrite(static_cast<Surface const&>(*surf).node_values().data(),
        sizeof(double), 4 * nn, f);
    std::fwrite(surf->node_validity().data(), 1, nn, f);
    std::fwrite(surf->patch_validity().data(), 1, np, f);
    std::fwrite(vis.data(), sizeof(uint32_t), np, f);
    std::fclose(f);
}

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
    // debugging aid: the grid surgery on the host Surface with an upload /
    // download per batch, as before round 4 (A/B against the device path)
    this->host_surgery = std::getenv("SMVS_HOST_SURGERY") != nullptr;
    this->prepare_correspondences();
    check(smvs_ctx_create(opts.device, main_view->get_width(),
        main_view->get_height(), (int)sub_views.size(), &this->ctx),
        "smvs_ctx_create");
    check(smvs_ctx_set_solver(ctx, opts.solver), "smvs_ctx_set_solver");
    check(smvs_ctx_set_cameras(ctx, Mi.data(), ti.data(),
        main_view->get_flen(), main_view->get_inverse_flen()),
        "smvs_ctx_set_cameras");
}

DepthOptimizer::DepthOptimizer(StereoView::Ptr main_view,
    std::vector<StereoView::Ptr> const& sub_views, Surface::Ptr surface,
    Options const& opts)
    : DepthOptimizer(main_view, sub_views, Bundle::ConstPtr(), opts)
{
    // lib/depth_optimizer.h:127-134
    this->surface = surface;
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
    FloatImage::Ptr out = FloatImage::create_for_overwrite(ci->width(), ci->height(), 1);
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
    // (first of all: the nine image transfers are only enqueued, and run while
    // the host converts the SGM map / projects the bundle below)
    this->upload_images();
    if (opts.use_sgm) {
        // depthmap_bilateral_filter(init, main_view->get_image()) guided by
        // the main image on the device; the filtered map also stays there for
        // create_subview_surfaces (:463-466 splats the same map)
        if (!host_surgery) {
            // ... and for Surface::create: the nodes are initialised from it
            // where it lies (no 8 MB round trip, no host Surface).  The map goes
            // to the device as the view stores it; get_sgm_depth()'s conversion
            // to z-depth (0.5 M pixels, ~1.2 ms of the host with the GPU waiting
            // for the map) runs in the kernel that fetches it, with the same
            // float operations.
            // (a map the SGM front end has just produced is still z-depth: both of
            // the view's conversions -- into the embedding and back -- then run
            // in that kernel)
            FloatImage::Ptr stored = main_view->get_deferred_depth("smvs-sgm");
            bool const still_z = stored != nullptr;
            if (!still_z)
                stored = main_view->get_embedding("smvs-sgm");
            if (stored == nullptr)
                throw std::invalid_argument("use_sgm without an smvs-sgm embedding");
            float invproj[9];
            main_view->get_camera().fill_inverse_calibration(invproj, (float)stored->width(),
                (float)stored->height());
            FloatImage::Ptr filtered;
            if (opts.debug_lvl > 1)     // :44-45
                filtered = FloatImage::create_for_overwrite(main_view->get_width(),
                    main_view->get_height(), 1);
            check(smvs_ctx_sgm_init_depth_mve(ctx, stored->begin(), stored->width(),
                stored->height(), invproj, still_z ? 1 : 0, 5.0f, 5,
                filtered ? filtered->begin() : nullptr), "smvs_ctx_sgm_init_depth_mve");
            if (filtered)
                main_view->write_depth_to_view(filtered, "smvs-sgm-filtered");
            check(smvs_surface_create(ctx, init_scale, nullptr, nullptr, nullptr, 0,
                &valid_patches), "smvs_surface_create");
            this->surface_on_device();
            return;
        }
        FloatImage::Ptr init = main_view->get_sgm_depth();
        if (init == nullptr)
            throw std::invalid_argument("use_sgm without an smvs-sgm embedding");
        FloatImage::Ptr full = FloatImage::create_for_overwrite(main_view->get_width(),
            main_view->get_height(), 1);
        check(smvs_ctx_sgm_init_depth(ctx, init->begin(), init->width(),
            init->height(), 5.0f, 5, full->begin()), "smvs_ctx_sgm_init_depth");
        init = full;
        this->surface = Surface::create(bundle, main_view, init_scale, init);
        this->sgm_depth = init;
    } else {
        if (bundle == nullptr)
            throw std::invalid_argument("DepthOptimizer: no bundle to "
                "initialise the surface from (use_sgm is off)");
        if (!host_surgery) {
            // initialize_depth_from_bundle (surface.cc:90-130) on the host --
            // a few thousand projections -- the node initialisation on the device
            std::vector<int32_t> pixels;
            std::vector<float> depths;
            Surface::project_bundle(bundle, main_view->get_camera(),
                main_view->get_view_id(), main_view->get_width(),
                main_view->get_height(), &pixels, &depths);
            check(smvs_surface_create(ctx, init_scale + 1, nullptr, pixels.data(),
                depths.data(), (int)pixels.size(), &valid_patches),
                "smvs_surface_create");
            this->surface_on_device();
            return;
        }
        this->surface = Surface::create(bundle, main_view, init_scale + 1);
    }
}

// The context's surface is the one that counts; a host Surface object is
// only materialised on demand (download_surface).
void
DepthOptimizer::surface_on_device(void)
{
    check(smvs_surface_info(ctx, &geom, nullptr), "smvs_surface_info");
    device_surface = true;
    surface.reset();
    uploaded_surface = nullptr;
}

int
DepthOptimizer::current_scale(void) const
{
    return device_surface ? geom.scale : surface->get_scale();
}

void
DepthOptimizer::set_scale_everywhere(int scale)
{
    // StereoView::set_scale for the main view and every neighbour
    // (lib/depth_optimizer.cc:63-66, 99-103) on the device.
    ScopedHostTimer timer("set_scale (device)");
    this->upload_images();
    check(smvs_ctx_set_scale(ctx, scale), "smvs_ctx_set_scale");
    // The planes stay on the device: every consumer (Newton loop, patch MSE,
    // NCC) runs there.  StereoView::get_image_gradients() of the views keeps
    // what the caller last set; download_scale_planes() fetches them.
}

void
DepthOptimizer::upload_images(void)
{
    if (!images_uploaded) {
        // (asynchronous: the views' bytes are page-locked (pinned_images.cc) and
        // outlive the optimizer; the nine transfers run on the context's copy
        // stream under the first kernels that need them)
        ByteImage::ConstPtr mb = main_view->get_raw_bytes();
        check(smvs_ctx_upload_image_async(ctx, -1, mb->width(), mb->height(),
            mb->channels(), mb->begin()), "smvs_ctx_upload_image_async");
        for (std::size_t j = 0; j < sub_views.size(); ++j) {
            ByteImage::ConstPtr sb = sub_views[j]->get_raw_bytes();
            check(smvs_ctx_upload_image_async(ctx, (int)j, sb->width(), sb->height(),
                sb->channels(), sb->begin()), "smvs_ctx_upload_image_async");
        }
        if (opts.use_shading)
            check(smvs_ctx_upload_shading(ctx,
                main_view->get_shading_image()->begin(),
                main_view->get_shading_gradients()->begin()),
                "smvs_ctx_upload_shading");
        images_uploaded = true;
    }
}

void
DepthOptimizer::download_scale_planes(void)
{
    {
        FloatImage::Ptr g = FloatImage::create_for_overwrite(main_view->get_width(),
            main_view->get_height(), 2);
        check(smvs_ctx_download_planes(ctx, -1, g->begin(), nullptr),
            "smvs_ctx_download_planes");
        main_view->set_scale_planes(g, nullptr);
    }
    for (std::size_t j = 0; j < sub_views.size(); ++j) {
        FloatImage::Ptr g = FloatImage::create_for_overwrite(sub_views[j]->get_width(),
            sub_views[j]->get_height(), 2);
        FloatImage::Ptr hs = FloatImage::create_for_overwrite(sub_views[j]->get_width(),
            sub_views[j]->get_height(), 3);
        check(smvs_ctx_download_planes(ctx, (int)j, g->begin(), hs->begin()),
            "smvs_ctx_download_planes");
        sub_views[j]->set_scale_planes(g, hs);
    }
}

void
DepthOptimizer::upload_surface(void)
{
    // Nothing to do when the device already holds exactly this surface (the
    // same object at the same revision with the same visibility masks): the
    // Newton loop's result is downloaded into the host surface, so the
    // cut_boundaries / get_depth / get_normals that follow it find the
    // device copy current.
    if (device_surface)
        return;   // (it lives in the context)
    if (surface == nullptr)
        throw std::logic_error("DepthOptimizer: no surface (call optimize() "
            "first or use the constructor that takes one)");
    Surface const& s = *surface;
    // (a surface handed to the constructor, or one that changed its grid, has
    // no visibility masks yet: get_depth() / get_normals() before optimize(),
    // lib/depth_optimizer.h:53-61)
    if (subsurfaces.size() != (std::size_t)s.get_num_patches()) {
        subsurfaces.assign(s.get_num_patches(), 0);
        subs_rev += 1;
    }
    if (uploaded_surface == surface.get() && uploaded_rev == s.revision()
        && uploaded_subs_rev == subs_rev)
        return;
    check(smvs_ctx_set_surface(ctx, s.get_scale(),
        s.get_num_patches_x(), s.get_num_patches_y(),
        s.get_pixel_start_x(), s.get_pixel_start_y(),
        s.node_values().data(), s.node_validity().data(),
        s.patch_validity().data(), subsurfaces.data()),
        "smvs_ctx_set_surface");
    uploaded_surface = surface.get();
    uploaded_rev = s.revision();
    uploaded_subs_rev = subs_rev;
}

// GlobalLighting::render_normal_map (lib/global_lighting.cc:61-79) with
// sh::evaluate_4_band (lib/spherical_harmonics.h:53-73, 133-151): per pixel the
// dot product of the 16 lighting parameters with the scaled basis; pixels whose
// normal is not of unit length (no surface) stay zero.
static FloatImage::Ptr
render_normal_map(double const* lighting, FloatImage::ConstPtr normals)
{
    FloatImage::Ptr image = FloatImage::create(normals->width(), normals->height(), 1);
    int64_t const n = (int64_t)normals->get_pixel_amount();
    for (int64_t i = 0; i < n; ++i) {
        double const x = normals->at(i, 0), y = normals->at(i, 1), z = normals->at(i, 2);
        if (std::fabs(std::sqrt(x * x + y * y + z * z) - 1.0) > 1e-6)
            continue;
        double const x2 = x * x, y2 = y * y, z2 = z * z;
        double const sh[16] = { 1.0, y, z, x, x * y, y * z, -x2 - y2 + 2.0 * z2, x * z,
            x * x - y * y, (3.0 * x2 - y2) * y, x * y * z, (4.0 * z2 - x2 - y2) * y,
            (2.0 * z2 - 3.0 * x2 - 3.0 * y2) * z, (4.0 * z2 - x2 - y2) * x, (x2 - y2) * z,
            (x2 - 3.0 * y2) * x };
        double v = 0.0;
        for (int k = 0; k < 16; ++k)
            v += lighting[k] * sh[k];
        image->at(i, 0) = (float)v;
    }
    return image;
}

// get_sphere_normals (lib/global_lighting.cc:24-50)
static FloatImage::Ptr
sphere_normals(int dim)
{
    if (dim % 2 == 0)
        dim += 1;
    FloatImage::Ptr normals = FloatImage::create(dim, dim, 3);
    for (int x = 0; x < dim; ++x)
        for (int y = 0; y < dim; ++y) {
            float const nx = 2.0f * (float)(x - dim / 2) / (float)dim;
            float const ny = 2.0f * (float)(dim / 2 - y) / (float)dim;
            if ((nx * nx + ny * ny) > 1.)
                continue;
            float const nz = std::sqrt(1 - nx * nx - ny * ny);
            int64_t const i = (int64_t)y * dim + x;
            normals->at(i, 0) = nx;
            normals->at(i, 1) = ny;
            normals->at(i, 2) = nz;
        }
    return normals;
}

void
DepthOptimizer::write_debug_depth(std::string const& postfix)
{
    // lib/depth_optimizer.h:150-160: "smvs-L<scale><postfix>" (the calls inside
    // the Newton loop, :242, write the same name as the one after it: the
    // embedding a caller finds is the last one of the scale)
    if (opts.debug_lvl < 2)
        return;
    main_view->write_depth_to_view(this->get_depth(),
        "smvs-L" + std::to_string(current_scale()) + postfix);
}

void
DepthOptimizer::write_debug_shading(bool with_albedo)
{
    // lib/depth_optimizer.cc:119-127, 139-156
    if (opts.debug_lvl < 2 || !lit)
        return;
    FloatImage::Ptr shaded = render_normal_map(lighting, this->get_normals());
    main_view->write_image_to_view(shaded, "smvs-shaded");
    main_view->write_image_to_view(render_normal_map(lighting, sphere_normals(555)),
        "smvs-shaded-sphere");
    if (!with_albedo || main_view->get_linear_image() == nullptr)
        return;
    FloatImage::Ptr albedo = main_view->get_linear_image()->duplicate();
    int64_t const n = (int64_t)albedo->get_pixel_amount();
    for (int64_t p = 0; p < n; ++p)
        for (int c = 0; c < albedo->channels(); ++c)
            albedo->at(p, c) = shaded->at(p, 0) > 0.0 ? albedo->at(p, c) / shaded->at(p, 0)
                : 0.0f;
    main_view->write_image_to_view(albedo, "smvs-implicit-albedo");
}

std::vector<RecordedLoop>&
recorded_loops(void)
{
    static thread_local std::vector<RecordedLoop> loops;
    return loops;
}

bool&
recording_loops(void)
{
    static thread_local bool on = false;
    return on;
}

void
DepthOptimizer::fit_lighting(void)
{
    // LightOptimizer::fit_lighting_to_image, lib/light_optimizer.cc:22-55
    if (!device_surface
        && subsurfaces.size() != (std::size_t)surface->get_num_patches()) {
        subsurfaces.resize(surface->get_num_patches(), 0);
        subs_rev += 1;
    }
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
    {
        ScopedHostTimer timer("create_initial_surface");
        this->create_initial_surface();
    }
    // Options::debug_lvl (lib/depth_optimizer.h:36): level 1 and above print
    // what the reference prints per scale and per iteration (:58-60, 84-86,
    // 92-94, 112-113, 132-134, 185-187, 306-316), fed from the batch log and
    // the device's kernel timers; level 2 also leaves the reference's
    // intermediate embeddings in the main view (smvs-sgm-filtered, smvs-initial,
    // smvs-L<scale>[-exp], smvs-shaded, smvs-shaded-sphere, smvs-implicit-albedo:
    // write_debug_depth / write_debug_shading); level 3's per-neighbour
    // reprojection images (reproject_neighbor, :701-745) are not produced.
    auto const scale_start = [this](int scale) {
        if (opts.debug_lvl > 0)
            std::cout << "########### Scale " << scale << " ###########" << std::endl;
        return std::chrono::steady_clock::now();
    };
    auto const scale_end = [this](std::chrono::steady_clock::time_point t0) {
        if (opts.debug_lvl > 0)
            std::cout << "Scale " << current_scale() << " took "
                << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()
                << "s" << std::endl;
    };
    if (opts.debug_lvl > 0)
        check(smvs_profile_enable(ctx, 1), "smvs_profile_enable");
    auto scale_timer = scale_start(current_scale());
    this->set_scale_everywhere(current_scale());
    if (opts.debug_lvl > 1)     // :68-70
        main_view->write_depth_to_view(this->get_depth(), "smvs-initial");
    this->run_newton_iterations(opts.num_iterations);
    scale_end(scale_timer);
    this->write_debug_depth();  // :87

    while (current_scale() > opts.min_scale && current_scale() > 0) {
        scale_timer = scale_start(current_scale() - 1);
        {
            ScopedHostTimer timer("subdivide_patches");
            if (device_surface) {
                // (fill_patches_from_depth below reports the valid patches)
                check(smvs_surface_subdivide(ctx, nullptr),
                    "smvs_surface_subdivide");
                check(smvs_surface_info(ctx, &geom, nullptr), "smvs_surface_info");
            } else
                surface->subdivide_patches();
        }
        this->set_scale_everywhere(current_scale());
        this->write_debug_depth();  // :104
        {
            ScopedHostTimer timer("fill_patches_from_depth");
            if (device_surface)
                check(smvs_surface_fill_patches_from_depth(ctx, &valid_patches),
                    "smvs_surface_fill_patches_from_depth");
            else
                surface->fill_patches_from_depth();
        }
        if (opts.use_shading && current_scale() < 4) {
            if (opts.debug_lvl > 0)
                std::cout << "######## with Lighting ########" << std::endl;
            ScopedHostTimer timer("fit_lighting");
            this->fit_lighting();
        }
        this->write_debug_shading(false);   // :119-127
        this->run_newton_iterations(opts.num_iterations);
        scale_end(scale_timer);
        this->write_debug_depth();  // :135
    }
    this->write_debug_shading(true);        // :139-156
    {
        // get_depth() + get_normals() (:150-160) in one pass over the surface
        ScopedHostTimer timer("depth + normal maps");
        this->upload_surface();
        FloatImage::Ptr dm = FloatImage::create_for_overwrite(main_view->get_width(),
            main_view->get_height(), 1);
        FloatImage::Ptr nm = FloatImage::create_for_overwrite(main_view->get_width(),
            main_view->get_height(), 3);
        // (the depth already in the convention write_depth_to_view stores,
        // stereo_view.h:100-119: converted where it is computed)
        float invproj[9];
        main_view->get_camera().fill_inverse_calibration(invproj,
            (float)main_view->get_width(), (float)main_view->get_height());
        check(smvs_get_maps(ctx, invproj, dm->begin(), nm->begin()), "smvs_get_maps");
        main_view->write_image_to_view(dm, opts.output_name);
        main_view->write_image_to_view(nm, opts.output_name + "N");
    }
    g_timers.report();
}

// The device surface as a host Surface object (debugging, dump_state).
Surface::Ptr
DepthOptimizer::download_surface(void) const
{
    if (!device_surface)
        return surface;
    std::size_t const N = (std::size_t)(geom.npx + 1) * (geom.npy + 1);
    std::size_t const P = (std::size_t)geom.npx * geom.npy;
    std::vector<double> nodes(4 * N);
    std::vector<uint8_t> nv(N), pv(P);
    check(smvs_surface_download(ctx, nodes.data(), nv.data(), pv.data(), nullptr),
        "smvs_surface_download");
    return Surface::from_arrays(main_view->get_width(), main_view->get_height(),
        geom.scale, geom.npx, geom.npy, geom.start_x, geom.start_y, nodes, nv, pv);
}

FloatImage::Ptr
DepthOptimizer::get_depth(void)
{
    // Surface::get_depth_map (lib/surface.cc:155-168) on the device
    this->upload_surface();
    FloatImage::Ptr dm = FloatImage::create_for_overwrite(main_view->get_width(),
        main_view->get_height(), 1);
    check(smvs_get_depth_map(ctx, dm->begin()), "smvs_get_depth_map");
    return dm;
}

FloatImage::Ptr
DepthOptimizer::get_normals(void)
{
    // Surface::get_normal_map (lib/surface.cc:170-183) on the device
    this->upload_surface();
    FloatImage::Ptr nm = FloatImage::create_for_overwrite(main_view->get_width(),
        main_view->get_height(), 3);
    check(smvs_get_normal_map(ctx, nm->begin()), "smvs_get_normal_map");
    return nm;
}

void
DepthOptimizer::run_newton_iterations(int num_iters)
{
    // lib/depth_optimizer.cc:164-358
    bool finished = false;
    for (int iter = 0; iter < num_iters; ++iter) {
        int const num_valid_patches = device_surface ? valid_patches
            : surface->count_valid_patches();
        if (opts.debug_lvl > 0 && iter == 0)
            std::cout << "Surface Status - Valid patches: " << num_valid_patches
                << std::endl;
        if (iter == 0) {
            {
                ScopedHostTimer timer("create_subview_surfaces");
                this->create_subview_surfaces();
            }
            ScopedHostTimer timer("cut_boundaries");
            this->cut_boundaries_until_stable();
        }

        // the whole Newton loop (:204-304) on the device
        {
            ScopedHostTimer timer("upload_surface");
            upload_surface();
        }
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
        if (recording_loops()) {
            smvs_ctx* clone = nullptr;
            check(smvs_ctx_clone_loop_state(ctx, &clone), "smvs_ctx_clone_loop_state");
            recorded_loops().push_back({ clone, prm, current_scale(), iter });
        }
        smvs_gn_loop_stats stats;
        double loop_seconds = 0.0;
        {
            ScopedHostTimer timer("device Newton loop");
            if (opts.debug_lvl > 0)
                check(smvs_profile_reset(ctx), "smvs_profile_reset");
            auto const t0 = std::chrono::steady_clock::now();
            check(smvs_gn_run_loop(ctx, &prm, &stats), "smvs_gn_run_loop");
            loop_seconds = std::chrono::duration<double>(
                std::chrono::steady_clock::now() - t0).count();
            if (!device_surface) {
                check(smvs_get_nodes(ctx, surface->node_values().data()),
                    "smvs_get_nodes");
                // host and device nodes are the same again
                if (uploaded_surface == surface.get())
                    uploaded_rev = surface->revision();
            }
        }
        log.push_back({ current_scale(), iter, stats.newton_steps,
            num_valid_patches, stats.linear_iterations, stats.active_patch_steps,
            loop_seconds });
        if (opts.debug_lvl > 0) {
            // lib/depth_optimizer.cc:306-316.  Construction = the patch kernel
            // (+ the assembly kernel when the streaming solver runs), solver =
            // the resident solve (its assembly prologue inside) or the streaming
            // kernels: HIP-event times of the loop's launches.
            double ms[SMVS_K_COUNT] = { 0.0 };
            long long launches[SMVS_K_COUNT] = { 0 };
            check(smvs_profile_get(ctx, ms, launches), "smvs_profile_get");
            double const steps = (double)stats.newton_steps;
            double const build = ms[SMVS_K_PATCH] + ms[SMVS_K_ASSEMBLE];
            double const solve = ms[SMVS_K_CG_RESIDENT] + ms[SMVS_K_CG_SPMV]
                + ms[SMVS_K_CG_UPDATE] + ms[SMVS_K_CG_INIT];
            std::cout << "### Finished iteration: " << iter << std::endl;
            std::cout << "Number of Newton steps: " << stats.newton_steps << std::endl;
            std::cout << "Avg construction time: " << build / steps << "ms" << std::endl;
            std::cout << "Avg solver time: " << solve / steps << "ms" << std::endl;
            // (integer division, as the reference's std::size_t counters)
            std::cout << "Avg solver iterations: "
                << (stats.newton_steps > 0 ? stats.linear_iterations / stats.newton_steps : 0)
                << std::endl;
            if (opts.debug_lvl > 1)
                std::cout << "Active patch-steps: " << stats.active_patch_steps
                    << ", device loop " << 1e3 * loop_seconds << " ms" << std::endl;
        }
        this->dump_state(iter, "newton");
        this->write_debug_depth();  // :317

        if (finished)
            break;
        {
            ScopedHostTimer timer("cut_boundaries");
            this->cut_boundaries_until_stable();
        }
        this->dump_state(iter, "cut");
        if (!opts.use_sgm) {
            {
                ScopedHostTimer timer("expand");
                if (device_surface)
                    check(smvs_surface_expand(ctx, nullptr, nullptr),
                        "smvs_surface_expand");
                else
                    surface->expand();
            }
            this->write_debug_depth("-exp");    // :332
            {
                ScopedHostTimer timer("create_subview_surfaces");
                this->create_subview_surfaces();
            }
            ScopedHostTimer timer("cut_boundaries");
            this->cut_boundaries_until_stable();
        }
        {
            ScopedHostTimer timer("remove_isolated_patches");
            if (device_surface)
                check(smvs_surface_remove_isolated_patches(ctx, &valid_patches),
                    "smvs_surface_remove_isolated_patches");
            else
                surface->remove_isolated_patches();
        }

        this->write_debug_depth();  // :352
        int const num_valid_new = device_surface ? valid_patches
            : surface->count_valid_patches();
        double const change = 1.0
            - (double)std::min(num_valid_new, num_valid_patches)
            / (double)std::max(num_valid_new, num_valid_patches);
        if (iter > 0 && (num_valid_new <= num_valid_patches
            || change < 0.05 * current_scale()))
            finished = true;
    }
}

int
DepthOptimizer::cut_boundaries_until_stable(void)
{
    // The `while (deleted > 10) cut_boundaries()` loops of
    // lib/depth_optimizer.cc:186-190, 323-337 with cut_boundaries (:360-431)
    // and mse_for_patch (:747-790) on the device.
    this->upload_surface();
    float invproj[9];
    main_view->get_camera().fill_inverse_calibration(invproj,
        (float)main_view->get_width(), (float)main_view->get_height());
    int total = 0;
    if (device_surface) {
        // (validity stays where it is; every deletion is one valid patch less)
        check(smvs_topology_cut_boundaries(ctx, invproj, nullptr, nullptr, &total),
            "smvs_topology_cut_boundaries");
        valid_patches -= total;
        return total;
    }
    // host surgery: the host applies the resulting validity to its Surface
    std::size_t const num_patches = surface->get_num_patches();
    std::vector<uint8_t> pv(num_patches), nv(surface->get_num_nodes());
    check(smvs_topology_cut_boundaries(ctx, invproj, pv.data(), nv.data(),
        &total), "smvs_topology_cut_boundaries");
    for (std::size_t p = 0; p < num_patches; ++p)
        if (surface->patch_validity()[p] && !pv[p])
            surface->delete_patch(p);
    surface->remove_nodes_without_patch();
    return total;
}

void
DepthOptimizer::create_subview_surfaces(void)
{
    // lib/depth_optimizer.cc:433-604.  The per-(patch, neighbour) tests --
    // z-buffer visibility, warp anisotropy, NCC -- run on the device
    // (smvs_topology_subviews), and so does the deletion of the patches
    // nobody sees (:592-603, smvs_surface_delete_unseen_patches).
    if (device_surface) {
        // (use_sgm: the filtered SGM map is resident, create_initial_surface)
        check(smvs_topology_subviews(ctx, nullptr, opts.use_sgm ? 0 : 1, nullptr),
            "smvs_topology_subviews");
        // (no count asked for, so no synchronisation: the batch's last
        // operation reports the valid patches, run_newton_iterations)
        check(smvs_surface_delete_unseen_patches(ctx, nullptr, nullptr),
            "smvs_surface_delete_unseen_patches");
        return;
    }
    std::size_t const num_patches = surface->get_num_patches();
    subsurfaces.assign(num_patches, 0);
    subs_rev += 1;
    this->upload_surface();
    check(smvs_topology_subviews(ctx, nullptr, opts.use_sgm ? 0 : 1,
        subsurfaces.data()), "smvs_topology_subviews");
    subs_rev += 1;   // (the device still holds the masks that were uploaded)

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
