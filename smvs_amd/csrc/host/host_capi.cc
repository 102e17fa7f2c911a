#include "host_capi.h"
#include "png_io.h"
#include "jpeg_io.h"

#include <algorithm>
#include <map>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

#include "depth_optimizer.h"
#include "sgm_stereo.h"
#include "view_selection.h"
#include "view_queue.h"
#include "scene_io.h"
#include "gauss_newton_step.h"
#include "conjugate_gradient.h"

#include <cmath>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>

using namespace smvs_amd;

static thread_local std::string g_host_error;

extern "C" const char *
smvs_host_last_error(void)
{
    return g_host_error.c_str();
}

static void fill_log(DepthOptimizer const& optimizer, smvs_host_log *log);

// the main view's embeddings after this thread's last smvs_host_optimize
static std::map<std::string, FloatImage::Ptr>&
last_embeddings(void)
{
    static thread_local std::map<std::string, FloatImage::Ptr> m;
    return m;
}

static StereoView::Ptr
make_view(smvs_host_view const& v, bool linear, bool gamma = false)
{
    ByteImage::Ptr img = ByteImage::create_for_overwrite(v.width, v.height, v.channels);
    std::memcpy(img->begin(), v.bytes, (size_t)v.width * v.height * v.channels);
    CameraInfo cam;
    cam.flen = v.flen;
    std::copy(v.rot, v.rot + 9, cam.rot);
    std::copy(v.trans, v.trans + 3, cam.trans);
    return StereoView::create(v.view_id, img, cam, linear, gamma);
}

static Bundle::Ptr
make_bundle(smvs_host_bundle const* b)
{
    if (b == nullptr)
        return nullptr;
    Bundle::Ptr bundle(new Bundle());
    bundle->features.resize(b->num_features);
    for (int i = 0; i < b->num_features; ++i) {
        std::copy(b->positions + 3 * i, b->positions + 3 * i + 3,
            bundle->features[i].pos);
        bundle->features[i].view_ids.assign(b->ref_views + b->ref_offsets[i],
            b->ref_views + b->ref_offsets[i + 1]);
    }
    return bundle;
}

extern "C" int
smvs_host_optimize(const smvs_host_view *main_in, const smvs_host_view *subs_in,
    int n_subs, const smvs_host_bundle *bundle_in, const float *sgm_depth,
    int sgm_w, int sgm_h, float *sgm_roundtrip, const smvs_host_options *o,
    float *depth_out, float *normals_out, smvs_host_log *log)
{
    try {
        // (the previous call's embeddings go back to the page-locked pool BEFORE
        // this call asks it for its maps: kept until the next call's end, the 33 MB
        // of a 1920 x 1080 depth + normal pair were allocated afresh every time --
        // 6.4 instead of 0.7 ms for `depth + normal maps`)
        last_embeddings().clear();
        StereoView::Ptr main_view = make_view(*main_in, o->use_shading != 0, o->gamma_correction != 0);
        std::vector<StereoView::Ptr> subs;
        for (int j = 0; j < n_subs; ++j)
            subs.push_back(make_view(subs_in[j], false));
        Bundle::Ptr bundle = make_bundle(bundle_in);
        if (sgm_depth != nullptr) {
            FloatImage::Ptr d = FloatImage::create_for_overwrite(sgm_w, sgm_h, 1);
            std::memcpy(d->begin(), sgm_depth, sizeof(float) * sgm_w * sgm_h);
            main_view->write_depth_to_view_deferred(d, "smvs-sgm");
            if (sgm_roundtrip != nullptr) {
                FloatImage::Ptr back = main_view->get_sgm_depth();
                std::memcpy(sgm_roundtrip, back->begin(),
                    sizeof(float) * sgm_w * sgm_h);
            }
        }
        DepthOptimizer::Options opts;
        opts.regularization = o->regularization;
        opts.light_surf_regularization = o->light_surf_regularization;
        opts.num_iterations = o->num_iterations;
        opts.min_scale = o->min_scale;
        opts.use_shading = o->use_shading != 0;
        opts.use_sgm = o->use_sgm != 0;
        opts.full_optimization = o->full_optimization != 0;
        opts.device = o->device;
        opts.solver = o->solver;
        // (the test harness sets Options::debug_lvl through the environment:
        // smvs_host_options is mirrored field for field by smvs_amd/host.py)
        if (const char *lvl = std::getenv("SMVS_DEBUG_LVL"))
            opts.debug_lvl = std::atoi(lvl);
        DepthOptimizer optimizer(main_view, subs, bundle, opts);
        optimizer.optimize();
        size_t const npix = (size_t)main_in->width * main_in->height;
        if (depth_out != nullptr)
            std::memcpy(depth_out, optimizer.get_depth()->begin(),
                sizeof(float) * npix);
        if (normals_out != nullptr)
            std::memcpy(normals_out, optimizer.get_normals()->begin(),
                sizeof(float) * 3 * npix);
        if (log != nullptr)
            fill_log(optimizer, log);
        last_embeddings() = main_view->get_embeddings();
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_embedding_names(char *names_out, int cap)
{
    std::string all;
    for (auto const& kv : last_embeddings())
        all += kv.first + "\n";
    if (names_out != nullptr && cap > 0) {
        std::size_t const n = std::min(all.size(), (std::size_t)cap - 1);
        std::memcpy(names_out, all.data(), n);
        names_out[n] = 0;
    }
    return (int)last_embeddings().size();
}

extern "C" int
smvs_host_embedding(const char *name, float *out, long long cap_floats, int *whc)
{
    auto it = last_embeddings().find(name ? name : "");
    if (it == last_embeddings().end() || it->second == nullptr)
        return -1;
    FloatImage::Ptr img = it->second;
    long long const n = (long long)img->get_pixel_amount() * img->channels();
    if (whc != nullptr) {
        whc[0] = img->width();
        whc[1] = img->height();
        whc[2] = img->channels();
    }
    if (out == nullptr || cap_floats < n)
        return (int)n;
    std::memcpy(out, img->begin(), sizeof(float) * (std::size_t)n);
    return 0;
}

extern "C" int
smvs_host_release_recorded_loops(int destroy)
{
    auto& loops = recorded_loops();
    int const n = (int)loops.size();
    if (destroy)
        for (auto& l : loops)
            (void)smvs_ctx_destroy(l.ctx);
    loops.clear();
    return n;
}

extern "C" int
smvs_host_record_loops(int on)
{
    if (on)
        (void)smvs_host_release_recorded_loops(1);
    recording_loops() = on != 0;
    return 0;
}

extern "C" int
smvs_host_recorded_loops(void **ctxs, void *params, int *scales, int *iters, int cap)
{
    auto const& loops = recorded_loops();
    smvs_gn_loop_params *prm = static_cast<smvs_gn_loop_params *>(params);
    for (int i = 0; i < (int)loops.size() && i < cap; ++i) {
        if (ctxs != nullptr)
            ctxs[i] = loops[i].ctx;
        if (prm != nullptr)
            prm[i] = loops[i].params;
        if (scales != nullptr)
            scales[i] = loops[i].scale;
        if (iters != nullptr)
            iters[i] = loops[i].iter;
    }
    return (int)loops.size();
}

extern "C" int
smvs_host_sgm_depth(const smvs_host_view *main_in, const smvs_host_view *subs_in,
    int n_subs, const smvs_host_bundle *bundle_in, int sgm_scale,
    float min_depth, float max_depth, int device, float *depth_out, int *out_w,
    int *out_h)
{
    try {
        StereoView::Ptr main_view = make_view(*main_in, false);
        std::vector<StereoView::Ptr> subs;
        for (int j = 0; j < n_subs; ++j)
            subs.push_back(make_view(subs_in[j], false));
        Bundle::Ptr bundle = make_bundle(bundle_in);
        SGMStereo::Options opts;
        opts.scale = sgm_scale;
        opts.num_steps = 128;
        opts.min_depth = min_depth;
        opts.max_depth = max_depth;
        opts.device = device;
        FloatImage::Ptr d = reconstruct_sgm_depth_for_view(opts, main_view, subs,
            bundle);
        if (out_w != nullptr)
            *out_w = d->width();
        if (out_h != nullptr)
            *out_h = d->height();
        if (depth_out != nullptr)
            std::memcpy(depth_out, d->begin(),
                sizeof(float) * d->get_pixel_amount());
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_select_neighbors(const smvs_host_view *views_in, int n_views,
    const smvs_host_bundle *bundle_in, int view, int num_neighbors, int *out,
    int *n_out)
{
    try {
        if (views_in == nullptr || out == nullptr || n_out == nullptr
            || view < 0 || view >= n_views || num_neighbors < 0)
            throw std::invalid_argument("smvs_host_select_neighbors: bad argument");
        ViewSelection::ViewList views((std::size_t)n_views);
        for (int i = 0; i < n_views; ++i) {
            smvs_host_view const& v = views_in[i];
            ViewSelection::ViewInfo& info = views[i];
            info.present = v.width > 0;
            info.id = v.view_id;
            info.cam.flen = v.flen;
            std::copy(v.rot, v.rot + 9, info.cam.rot);
            std::copy(v.trans, v.trans + 3, info.cam.trans);
            info.has_image = v.bytes != nullptr;
            info.width = v.width;
            info.height = v.height;
        }
        ViewSelection::Options opts;
        opts.num_neighbors = (std::size_t)num_neighbors;
        Bundle::Ptr bundle = make_bundle(bundle_in);
        ViewSelection selection(opts, views, bundle);
        std::vector<std::size_t> const chosen
            = selection.get_neighbors_for_view((std::size_t)view);
        *n_out = (int)chosen.size();
        for (std::size_t k = 0; k < chosen.size(); ++k)
            out[k] = (int)chosen[k];
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_surface_script(const smvs_host_view *main_in,
    const smvs_host_bundle *bundle_in, const float *init_depth, int init_scale,
    const int *ops, int n_ops, int delete_every, int *info, double *nodes_out,
    uint8_t *node_valid_out, uint8_t *patch_valid_out)
{
    try {
        if (main_in == nullptr || info == nullptr || nodes_out == nullptr
            || node_valid_out == nullptr || patch_valid_out == nullptr
            || (n_ops > 0 && ops == nullptr) || delete_every < 1)
            throw std::invalid_argument("smvs_host_surface_script: bad argument");
        StereoView::Ptr main_view = make_view(*main_in, false);
        Bundle::Ptr bundle = make_bundle(bundle_in);
        FloatImage::Ptr init;
        if (init_depth != nullptr) {
            init = FloatImage::create_for_overwrite(main_in->width, main_in->height, 1);
            std::memcpy(init->begin(), init_depth,
                sizeof(float) * (size_t)main_in->width * main_in->height);
        }
        Surface::Ptr surface = Surface::create(bundle, main_view, init_scale, init);
        for (int k = 0; k < n_ops; ++k)
            switch (ops[k]) {
            case 1: surface->expand(); break;
            case 2: surface->subdivide_patches(); break;
            case 3: surface->fill_patches_from_depth(); break;
            case 4: surface->remove_isolated_patches(); break;
            case 5: {
                int seen = 0;
                for (std::size_t p = 0; p < surface->patch_validity().size(); ++p)
                    if (surface->patch_validity()[p] && (++seen % delete_every) == 0)
                        surface->delete_patch(p);
                surface->remove_nodes_without_patch();
                break;
            }
            default:
                throw std::invalid_argument("smvs_host_surface_script: unknown operation");
            }
        Surface const& s = *surface;
        info[0] = s.get_scale();
        info[1] = s.get_num_patches_x();
        info[2] = s.get_num_patches_y();
        info[3] = s.get_pixel_start_x();
        info[4] = s.get_pixel_start_y();
        std::copy(s.node_values().begin(), s.node_values().end(), nodes_out);
        std::copy(s.node_validity().begin(), s.node_validity().end(), node_valid_out);
        std::copy(s.patch_validity().begin(), s.patch_validity().end(), patch_valid_out);
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

// The same script on the surface of a device context (csrc/surface.hip).
extern "C" int
smvs_host_surface_script_device(const smvs_host_view *main_in,
    const smvs_host_bundle *bundle_in, const float *init_depth, int init_scale,
    const int *ops, int n_ops, int delete_every, int device, int *info,
    double *nodes_out, uint8_t *node_valid_out, uint8_t *patch_valid_out)
{
    smvs_ctx *ctx = nullptr;
    try {
        if (main_in == nullptr || info == nullptr || nodes_out == nullptr
            || node_valid_out == nullptr || patch_valid_out == nullptr
            || (n_ops > 0 && ops == nullptr) || delete_every < 1)
            throw std::invalid_argument("smvs_host_surface_script_device: bad argument");
        auto check = [](int rc, char const* what) {
            if (rc != SMVS_OK)
                throw std::runtime_error(std::string(what) + ": " + smvs_last_error());
        };
        StereoView::Ptr main_view = make_view(*main_in, false);
        check(smvs_ctx_create(device, main_in->width, main_in->height, 1, &ctx),
            "smvs_ctx_create");
        int valid = 0;
        if (init_depth != nullptr) {
            check(smvs_surface_create(ctx, init_scale, init_depth, nullptr, nullptr, 0,
                &valid), "smvs_surface_create");
        } else {
            Bundle::Ptr bundle = make_bundle(bundle_in);
            std::vector<int32_t> pixels;
            std::vector<float> depths;
            Surface::project_bundle(bundle, main_view->get_camera(),
                main_view->get_view_id(), main_in->width, main_in->height, &pixels,
                &depths);
            if (pixels.empty())
                throw std::invalid_argument("smvs_host_surface_script_device: the "
                    "bundle has no feature in this view");
            check(smvs_surface_create(ctx, init_scale, nullptr, pixels.data(),
                depths.data(), (int)pixels.size(), &valid), "smvs_surface_create");
        }
        for (int k = 0; k < n_ops; ++k)
            switch (ops[k]) {
            case 1: check(smvs_surface_expand(ctx, nullptr, &valid), "expand"); break;
            case 2: check(smvs_surface_subdivide(ctx, &valid), "subdivide"); break;
            case 3: check(smvs_surface_fill_patches_from_depth(ctx, &valid), "fill"); break;
            case 4: check(smvs_surface_remove_isolated_patches(ctx, &valid), "isolated"); break;
            case 5: check(smvs_surface_delete_every(ctx, delete_every, &valid), "delete"); break;
            default:
                throw std::invalid_argument("smvs_host_surface_script_device: unknown operation");
            }
        smvs_surface_geometry g;
        int counted = 0;
        check(smvs_surface_info(ctx, &g, &counted), "smvs_surface_info");
        if (counted != valid)
            throw std::runtime_error("smvs_host_surface_script_device: the operation's "
                "count of valid patches differs from a recount");
        info[0] = g.scale;
        info[1] = g.npx;
        info[2] = g.npy;
        info[3] = g.start_x;
        info[4] = g.start_y;
        info[5] = valid;
        check(smvs_surface_download(ctx, nodes_out, node_valid_out, patch_valid_out,
            nullptr), "smvs_surface_download");
        smvs_ctx_destroy(ctx);
        return 0;
    } catch (std::exception const& e) {
        if (ctx != nullptr)
            smvs_ctx_destroy(ctx);
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_depth_range(const smvs_host_view *view_in,
    const smvs_host_bundle *bundle_in, float *range2)
{
    try {
        if (view_in == nullptr || bundle_in == nullptr || range2 == nullptr)
            throw std::invalid_argument("smvs_host_depth_range: bad argument");
        StereoView::Ptr view = make_view(*view_in, false);
        Bundle::Ptr bundle = make_bundle(bundle_in);
        SGMStereo::fill_depth_range_for_view(bundle, view, range2);
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_reprojection(const smvs_host_view *source,
    const smvs_host_view *destination, float *M9, float *t3)
{
    try {
        if (source == nullptr || destination == nullptr || M9 == nullptr
            || t3 == nullptr)
            throw std::invalid_argument("smvs_host_reprojection: bad argument");
        CameraInfo src, dst;
        src.flen = source->flen;
        std::copy(source->rot, source->rot + 9, src.rot);
        std::copy(source->trans, source->trans + 3, src.trans);
        dst.flen = destination->flen;
        std::copy(destination->rot, destination->rot + 9, dst.rot);
        std::copy(destination->trans, destination->trans + 3, dst.trans);
        src.fill_reprojection(dst, (float)source->width, (float)source->height,
            (float)destination->width, (float)destination->height, M9, t3);
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_sgm_image(const smvs_host_view *view_in, int halvings, uint8_t *out,
    int *out_w, int *out_h)
{
    try {
        if (view_in == nullptr || out == nullptr || out_w == nullptr
            || out_h == nullptr || halvings < 0)
            throw std::invalid_argument("smvs_host_sgm_image: bad argument");
        StereoView::Ptr view = make_view(*view_in, false);
        ByteImage::ConstPtr img = view->get_byte_image();
        for (int i = 0; i < halvings; ++i)
            img = imgtools::rescale_half_size(img);
        *out_w = img->width();
        *out_h = img->height();
        std::memcpy(out, img->begin(), (size_t)img->width() * img->height());
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_surface_maps(const smvs_host_view *main_in, const smvs_host_view *subs_in,
    int n_subs, const smvs_host_bundle *bundle_in, const float *init_depth,
    int init_scale, int device, float *depth_out, float *normals_out)
{
    try {
        if (main_in == nullptr || subs_in == nullptr || n_subs < 1
            || depth_out == nullptr || normals_out == nullptr)
            throw std::invalid_argument("smvs_host_surface_maps: bad argument");
        StereoView::Ptr main_view = make_view(*main_in, false);
        std::vector<StereoView::Ptr> subs;
        for (int j = 0; j < n_subs; ++j)
            subs.push_back(make_view(subs_in[j], false));
        Bundle::Ptr bundle = make_bundle(bundle_in);
        FloatImage::Ptr init;
        if (init_depth != nullptr) {
            init = FloatImage::create_for_overwrite(main_in->width, main_in->height, 1);
            std::memcpy(init->begin(), init_depth,
                sizeof(float) * (size_t)main_in->width * main_in->height);
        }
        Surface::Ptr surface = Surface::create(bundle, main_view, init_scale, init);
        DepthOptimizer::Options opts;
        opts.device = device;
        // lib/depth_optimizer.h:53-61: the surface constructor, then the
        // maps without a prior optimize()
        DepthOptimizer optimizer(main_view, subs, surface, opts);
        size_t const npix = (size_t)main_in->width * main_in->height;
        std::memcpy(depth_out, optimizer.get_depth()->begin(), sizeof(float) * npix);
        std::memcpy(normals_out, optimizer.get_normals()->begin(),
            sizeof(float) * 3 * npix);
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

static void
fill_log(DepthOptimizer const& optimizer, smvs_host_log *log)
{
    log->count = 0;
    for (auto const& e : optimizer.get_log()) {
        if (log->count >= SMVS_HOST_LOG_MAX)
            break;
        int const i = log->count++;
        log->scale[i] = e.scale;
        log->iter[i] = e.iter;
        log->newton_steps[i] = e.newton_steps;
        log->valid_patches[i] = e.valid_patches;
        log->cg_iterations[i] = e.cg_iterations;
        log->active_patch_steps[i] = e.active_patch_steps;
        log->loop_seconds[i] = e.loop_seconds;
    }
    log->has_lighting = optimizer.has_lighting() ? 1 : 0;
    std::copy(optimizer.get_lighting(), optimizer.get_lighting() + 16,
        log->lighting);
}

extern "C" int
smvs_host_optimize_views(const smvs_host_view *mains, const smvs_host_view *subs_in,
    int n_jobs, int n_subs, const smvs_host_bundle *bundle_in,
    const smvs_host_options *o, int sgm_scale, int first_device, int num_devices,
    int views_in_flight, int keep_job, float *depth_out, float *normals_out,
    double *job_seconds, double *total_seconds, smvs_host_log *logs)
{
    try {
        if (mains == nullptr || subs_in == nullptr || o == nullptr || n_jobs < 1
            || n_subs < 1 || num_devices < 1 || views_in_flight < 1)
            throw std::invalid_argument("smvs_host_optimize_views: bad argument");
        Bundle::Ptr bundle = make_bundle(bundle_in);
        typedef std::chrono::steady_clock Clock;
        // per-phase wall time over all tasks (stderr with SMVS_HOST_TIMING)
        static const char *const phase_names[5] = { "StereoView::create x (1 + subs)",
            "SGM front end", "DepthOptimizer ctor", "optimize()", "maps for the caller (one job)" };
        std::mutex phase_lock;
        double phase_s[5] = { 0, 0, 0, 0, 0 };
        auto lap = [&](Clock::time_point& from, int phase) {
            Clock::time_point const now = Clock::now();
            std::lock_guard<std::mutex> guard(phase_lock);
            phase_s[phase] += std::chrono::duration<double>(now - from).count();
            from = now;
        };
        std::vector<std::future<void>> done;
        Clock::time_point const t_start = Clock::now();
        {
            ViewQueue queue(num_devices, views_in_flight);
            for (int job = 0; job < n_jobs; ++job)
                done.push_back(queue.add_task([=, &lap](ViewQueue::Slot const& slot) {
                    // app/smvsrecon.cc:662-732
                    Clock::time_point const t0 = Clock::now();
                    Clock::time_point tp = t0;
                    StereoView::Ptr main_view = make_view(mains[job],
                        o->use_shading != 0);
                    std::vector<StereoView::Ptr> subs;
                    for (int j = 0; j < n_subs; ++j)
                        subs.push_back(make_view(subs_in[(size_t)job * n_subs + j],
                            false));
                    lap(tp, 0);
                    int const device = first_device + slot.device;
                    bool const use_sgm = sgm_scale >= 0;
                    if (use_sgm) {
                        SGMStereo::Options sgm_opts;
                        sgm_opts.scale = sgm_scale;
                        sgm_opts.num_steps = 128;
                        sgm_opts.device = device;
                        (void)reconstruct_sgm_depth_for_view(sgm_opts, main_view, subs,
                            bundle);
                    }
                    DepthOptimizer::Options opts;
                    opts.regularization = o->regularization;
                    opts.light_surf_regularization = o->light_surf_regularization;
                    opts.num_iterations = o->num_iterations;
                    opts.min_scale = o->min_scale;
                    opts.use_shading = o->use_shading != 0;
                    opts.use_sgm = use_sgm;
                    opts.full_optimization = o->full_optimization != 0;
                    opts.device = device;
                    opts.solver = o->solver;
                    lap(tp, 1);
                    DepthOptimizer optimizer(main_view, subs, bundle, opts);
                    lap(tp, 2);
                    optimizer.optimize();
                    lap(tp, 3);
                    // (optimize() has written the depth / normal embeddings,
                    // lib/depth_optimizer.cc:158-161; the maps are fetched once
                    // more only for the job whose result the caller wants)
                    if (job == keep_job) {
                        FloatImage::Ptr depth = optimizer.get_depth();
                        FloatImage::Ptr normals = optimizer.get_normals();
                        size_t const npix = (size_t)mains[job].width * mains[job].height;
                        if (depth_out != nullptr)
                            std::memcpy(depth_out, depth->begin(), sizeof(float) * npix);
                        if (normals_out != nullptr)
                            std::memcpy(normals_out, normals->begin(),
                                sizeof(float) * 3 * npix);
                    }
                    lap(tp, 4);
                    if (logs != nullptr)
                        fill_log(optimizer, &logs[job]);
                    if (job_seconds != nullptr)
                        job_seconds[job] = std::chrono::duration<double>(
                            Clock::now() - t0).count();
                }));
            queue.wait_idle();
        }
        if (total_seconds != nullptr)
            *total_seconds = std::chrono::duration<double>(Clock::now() - t_start).count();
        if (std::getenv("SMVS_HOST_TIMING") != nullptr)
            for (int i = 0; i < 5; ++i)
                std::fprintf(stderr, "[smvs views] %-34s %8.2f ms per view\n",
                    phase_names[i], 1e3 * phase_s[i] / n_jobs);
        for (auto& f : done)
            f.get();   // rethrows a task's exception
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_view_queue_selftest(int n_tasks, int num_devices, int views_in_flight,
    int throwing_task, int *device_hist, int *worker_hist)
{
    try {
        if (n_tasks < 0 || device_hist == nullptr || worker_hist == nullptr)
            throw std::invalid_argument("smvs_host_view_queue_selftest: bad argument");
        std::vector<int> ran(n_tasks, 0), on_device(n_tasks, -1), on_worker(n_tasks, -1);
        std::vector<std::future<void>> done;
        {
            ViewQueue queue(num_devices, views_in_flight);
            if (queue.num_workers() != num_devices * views_in_flight)
                throw std::logic_error("ViewQueue: wrong number of workers");
            for (int i = 0; i < n_tasks; ++i)
                done.push_back(queue.add_task([&, i](ViewQueue::Slot const& slot) {
                    ran[i] += 1;
                    on_device[i] = slot.device;
                    on_worker[i] = slot.worker;
                    std::this_thread::sleep_for(std::chrono::milliseconds(2));
                    if (i == throwing_task)
                        throw std::runtime_error("task failed on purpose");
                }));
            queue.wait_idle();
        }
        std::fill(device_hist, device_hist + num_devices, 0);
        std::fill(worker_hist, worker_hist + num_devices * views_in_flight, 0);
        for (int i = 0; i < n_tasks; ++i) {
            bool threw = false;
            try {
                done[i].get();
            } catch (std::runtime_error const&) {
                threw = true;
            }
            if (threw != (i == throwing_task))
                throw std::logic_error("ViewQueue: exception on the wrong future");
            if (ran[i] != 1 || on_device[i] < 0 || on_device[i] >= num_devices)
                throw std::logic_error("ViewQueue: a task did not run exactly once");
            if (on_worker[i] % num_devices != on_device[i])
                throw std::logic_error("ViewQueue: worker / device binding");
            device_hist[on_device[i]] += 1;
            worker_hist[on_worker[i]] += 1;
        }
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_gn_solve_step(const smvs_host_view *main_in, const smvs_host_view *subs_in,
    int n_subs, const smvs_host_bundle *bundle_in, int init_scale,
    double regularization, int device, float *main_grad, float *sub_grad,
    float *sub_hess, double *Mi_out, double *ti_out, float *flen2, double *g_out,
    double *x_out, double *H9_out, double *P_out, int *cg_out)
{
    try {
        if (main_in == nullptr || subs_in == nullptr || n_subs < 1)
            throw std::invalid_argument("smvs_host_gn_solve_step: bad argument");
        StereoView::Ptr main_view = make_view(*main_in, false);
        std::vector<StereoView::Ptr> subs;
        for (int j = 0; j < n_subs; ++j)
            subs.push_back(make_view(subs_in[j], false));
        Bundle::Ptr bundle = make_bundle(bundle_in);
        Surface::Ptr surface = Surface::create(bundle, main_view, init_scale);
        // lib/depth_optimizer.cc:63-66
        main_view->set_scale(surface->get_scale());
        for (auto& v : subs)
            v->set_scale(surface->get_scale());
        // lib/depth_optimizer.cc:679-699
        std::vector<Matrix3d> Mi(n_subs);
        std::vector<Vec3d> ti(n_subs);
        for (int j = 0; j < n_subs; ++j) {
            float M[9], t[3];
            main_view->get_camera().fill_reprojection(subs[j]->get_camera(),
                (float)main_view->get_width(), (float)main_view->get_height(),
                (float)subs[j]->get_width(), (float)subs[j]->get_height(), M, t);
            for (int k = 0; k < 9; ++k)
                Mi[j].m[k] = M[k];
            for (int k = 0; k < 3; ++k)
                ti[j].v[k] = t[k];
        }
        std::size_t const N = surface->get_num_nodes(), Pn = surface->get_num_patches();
        std::vector<std::vector<std::size_t>> subsurfaces(Pn);
        for (std::size_t p = 0; p < Pn; ++p)
            if (surface->patch_validity()[p])
                for (int j = 0; j < n_subs; ++j)
                    subsurfaces[p].push_back((std::size_t)j);
        std::vector<char> active(N);
        for (std::size_t n = 0; n < N; ++n)
            active[n] = surface->node_validity()[n] ? 1 : 0;

        GaussNewtonStep::Options gn_opts;
        gn_opts.regularization = regularization;
        gn_opts.device = device;
        GaussNewtonStep gauss_newton_step(gn_opts, main_view, subs, Mi, ti);
        GaussNewtonStep::SparseMatrix hessian, precond;
        GaussNewtonStep::DenseVector gradient;
        gauss_newton_step.construct(surface, subsurfaces, active, nullptr,
            &hessian, &gradient, &precond);
        // lib/depth_optimizer.cc:245-254
        double norm = 0.0;
        for (double v : gradient)
            norm += v * v;
        ConjugateGradient::Options cg_opts;
        cg_opts.error_tolerance = std::sqrt(norm) * 0.01;
        cg_opts.max_iterations = 200;
        ConjugateGradient cg(cg_opts, device);
        ConjugateGradient::Vector b(gradient.size()), delta;
        for (std::size_t i = 0; i < b.size(); ++i)
            b[i] = -gradient[i];
        ConjugateGradient::Status const status = cg.solve(hessian, b, &delta, &precond);

        std::size_t const npix = (size_t)main_in->width * main_in->height;
        if (main_grad != nullptr)
            std::memcpy(main_grad, main_view->get_image_gradients()->begin(),
                sizeof(float) * 2 * npix);
        for (int j = 0; j < n_subs; ++j) {
            std::size_t const sp = (size_t)subs[j]->get_width() * subs[j]->get_height();
            if (sp != npix)
                throw std::invalid_argument("smvs_host_gn_solve_step: the planes "
                    "are returned for neighbours of the main view's size");
            if (sub_grad != nullptr)
                std::memcpy(sub_grad + 2 * npix * j,
                    subs[j]->get_image_gradients()->begin(), sizeof(float) * 2 * npix);
            if (sub_hess != nullptr)
                std::memcpy(sub_hess + 3 * npix * j,
                    subs[j]->get_image_hessian()->begin(), sizeof(float) * 3 * npix);
            if (Mi_out != nullptr)
                std::copy(Mi[j].m, Mi[j].m + 9, Mi_out + 9 * j);
            if (ti_out != nullptr)
                std::copy(ti[j].v, ti[j].v + 3, ti_out + 3 * j);
        }
        if (flen2 != nullptr) {
            flen2[0] = main_view->get_flen();
            flen2[1] = main_view->get_inverse_flen();
        }
        if (g_out != nullptr)
            std::copy(gradient.begin(), gradient.end(), g_out);
        if (x_out != nullptr)
            std::copy(delta.begin(), delta.end(), x_out);
        if (H9_out != nullptr)
            std::copy(hessian.blocks.begin(), hessian.blocks.end(), H9_out);
        if (P_out != nullptr)
            for (std::size_t n = 0; n < N; ++n)
                std::copy(precond.blocks.begin() + (n * 9 + 4) * 16,
                    precond.blocks.begin() + (n * 9 + 5) * 16, P_out + n * 16);
        if (cg_out != nullptr) {
            cg_out[0] = status.num_iterations;
            cg_out[1] = (int)status.info;
        }
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_reconstruct_scene(const char *scene_dir,
    const smvs_host_recon_settings *o, const int *view_ids, int n_view_ids,
    int *reconstructed_out, int max_reconstructed, int *n_reconstructed,
    int *n_skipped, double *seconds, int *input_scale_used)
{
    try {
        if (scene_dir == nullptr || o == nullptr)
            throw std::invalid_argument("smvs_host_reconstruct_scene: bad argument");
        ReconSettings conf;
        if (o->image_embedding != nullptr)
            conf.image_embedding = o->image_embedding;
        conf.regularization = o->regularization;
        conf.output_scale = o->output_scale;
        conf.use_shading = o->use_shading != 0;
        conf.use_sgm = o->use_sgm != 0;
        conf.force_recon = o->force_recon != 0;
        conf.force_sgm = o->force_sgm != 0;
        conf.full_optimization = o->full_optimization != 0;
        conf.sgm_min = o->sgm_min;
        conf.sgm_max = o->sgm_max;
        conf.sgm_scale = o->sgm_scale;
        conf.num_neighbors = (std::size_t)o->num_neighbors;
        conf.min_neighbors = (std::size_t)o->min_neighbors;
        conf.first_device = o->first_device;
        conf.num_devices = o->num_devices;
        conf.views_in_flight = o->views_in_flight;
        conf.input_scale = o->input_scale;
        if (o->max_pixels > 0)
            conf.max_pixels = (std::size_t)o->max_pixels;
        if (view_ids != nullptr)
            conf.view_ids.assign(view_ids, view_ids + n_view_ids);
        ReconReport const report = reconstruct_scene(scene_dir, conf);
        if (reconstructed_out != nullptr && max_reconstructed > 0)
            std::copy_n(report.reconstructed.begin(),
                std::min<std::size_t>(report.reconstructed.size(),
                    (std::size_t)max_reconstructed), reconstructed_out);
        if (n_reconstructed != nullptr)
            *n_reconstructed = (int)report.reconstructed.size();
        if (n_skipped != nullptr)
            *n_skipped = (int)(report.skipped_few_neighbors.size()
                + report.already_done.size());
        if (seconds != nullptr)
            *seconds = report.seconds;
        if (input_scale_used != nullptr)
            *input_scale_used = report.input_scale;
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_scene_info(const char *scene_dir, const char *image_embedding,
    int max_views, int *n_views, int *present, float *flen, float *rot9,
    float *trans3, int *width, int *height, int *n_features)
{
    try {
        if (scene_dir == nullptr || image_embedding == nullptr || n_views == nullptr)
            throw std::invalid_argument("smvs_host_scene_info: bad argument");
        Scene::Ptr scene = Scene::create(scene_dir);
        std::vector<SceneView> const& views = scene->get_views();
        *n_views = (int)views.size();
        if ((int)views.size() > max_views)
            throw std::invalid_argument("smvs_host_scene_info: more views than room");
        for (std::size_t i = 0; i < views.size(); ++i) {
            SceneView const& v = views[i];
            present[i] = v.present ? 1 : 0;
            flen[i] = v.camera.flen;
            std::copy(v.camera.rot, v.camera.rot + 9, rot9 + 9 * i);
            std::copy(v.camera.trans, v.camera.trans + 3, trans3 + 3 * i);
            int whc[3] = { 0, 0, 0 };
            if (v.present && v.has_image(image_embedding))
                (void)v.image_size(image_embedding, whc);
            width[i] = whc[0];
            height[i] = whc[1];
        }
        if (n_features != nullptr) {
            try {
                *n_features = (int)scene->get_bundle()->features.size();
            } catch (std::exception const&) {
                *n_features = -1;
            }
        }
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_mvei_roundtrip(const char *in_path, const char *out_path)
{
    try {
        if (in_path == nullptr || out_path == nullptr)
            throw std::invalid_argument("smvs_host_mvei_roundtrip: bad argument");
        int whct[4];
        if (!mvei_header(in_path, whct))
            throw std::runtime_error(std::string("not an .mvei file: ") + in_path);
        if (whct[3] == 1)
            save_mvei(out_path, ByteImage::ConstPtr(load_mvei_u8(in_path)));
        else
            save_mvei(out_path, FloatImage::ConstPtr(load_mvei_float(in_path)));
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_load_byte_image(const char *path, int *whc, uint8_t *pixels, size_t capacity)
{
    try {
        if (path == nullptr || whc == nullptr)
            throw std::invalid_argument("smvs_host_load_byte_image: bad argument");
        std::string const p = path;
        auto ends_with = [&](char const* ext) {
            std::size_t const n = std::strlen(ext);
            return p.size() > n && p.compare(p.size() - n, n, ext) == 0;
        };
        ByteImage::Ptr img = ends_with(".png") ? load_png_u8(p)
            : (ends_with(".jpg") || ends_with(".jpeg")) ? load_jpeg_u8(p) : load_mvei_u8(p);
        whc[0] = img->width();
        whc[1] = img->height();
        whc[2] = img->channels();
        std::size_t const n = (std::size_t)img->width() * img->height() * img->channels();
        if (pixels != nullptr) {
            if (capacity < n)
                throw std::invalid_argument("smvs_host_load_byte_image: buffer too small");
            std::memcpy(pixels, img->begin(), n);
        }
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_save_png(const char *path, const uint8_t *pixels, int width, int height,
    int channels)
{
    try {
        if (path == nullptr || pixels == nullptr || width < 1 || height < 1)
            throw std::invalid_argument("smvs_host_save_png: bad argument");
        ByteImage::Ptr img = ByteImage::create_for_overwrite(width, height, channels);
        std::memcpy(img->begin(), pixels, (std::size_t)width * height * channels);
        save_png_u8(path, img);
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_rescale_half_size_gaussian(const uint8_t *pixels, int width, int height,
    int channels, uint8_t *out)
{
    try {
        if (pixels == nullptr || out == nullptr || channels < 1)
            throw std::invalid_argument("smvs_host_rescale_half_size_gaussian: bad argument");
        ByteImage::Ptr img = ByteImage::create_for_overwrite(width, height, channels);
        std::memcpy(img->begin(), pixels, (std::size_t)width * height * channels);
        ByteImage::Ptr half = rescale_half_size_gaussian(img);
        std::memcpy(out, half->begin(),
            (std::size_t)half->width() * half->height() * half->channels());
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}

extern "C" int
smvs_host_block_multiply(int num_nodes, int node_stride, const double *blocks9,
    const double *x, double *y)
{
    try {
        if (num_nodes < 1 || node_stride < 1 || blocks9 == nullptr || x == nullptr
            || y == nullptr)
            throw std::invalid_argument("smvs_host_block_multiply: bad argument");
        BlockStencilMatrix A;
        A.num_nodes = (std::size_t)num_nodes;
        A.node_stride = (std::size_t)node_stride;
        A.blocks.assign(blocks9, blocks9 + (std::size_t)num_nodes * 9 * 16);
        ConjugateGradient::Functor const& op = A;      // (through the interface)
        ConjugateGradient::Vector const r = op.multiply(
            ConjugateGradient::Vector(x, x + 4 * (std::size_t)num_nodes));
        std::copy(r.begin(), r.end(), y);
        return 0;
    } catch (std::exception const& e) {
        g_host_error = e.what();
        return -1;
    }
}
