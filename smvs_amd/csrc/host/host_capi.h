/* C entry points of the host mirror (libsmvs_host.so): lets tests / tools
 * drive smvs_amd::DepthOptimizer and smvs_amd::SGMStereo without a C++
 * toolchain.  Not part of the device boundary (that is include/smvs_hip.h). */
#ifndef SMVS_HOST_CAPI_H
#define SMVS_HOST_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int width, height, channels;
    const uint8_t *bytes;       /* interleaved u8 image */
    float flen;                 /* CameraInfo::flen */
    float rot[9], trans[3];
    int view_id;
} smvs_host_view;

typedef struct {
    int num_features;
    const float *positions;     /* num_features * 3 */
    const int *ref_offsets;     /* num_features + 1 */
    const int *ref_views;
} smvs_host_bundle;

typedef struct {                /* DepthOptimizer::Options */
    double regularization;
    double light_surf_regularization;
    int num_iterations;
    int min_scale;
    int use_shading;
    int use_sgm;
    int full_optimization;
    int device;
    int solver;                 /* smvs_solver_mode of include/smvs_hip.h */
    int gamma_correction;       /* StereoView::create's fourth argument for the main view
                                   (app/smvsrecon.cc:52, 669), with use_shading only */
} smvs_host_options;

#define SMVS_HOST_LOG_MAX 256
typedef struct {
    int count;
    int scale[SMVS_HOST_LOG_MAX];
    int iter[SMVS_HOST_LOG_MAX];
    int newton_steps[SMVS_HOST_LOG_MAX];
    int valid_patches[SMVS_HOST_LOG_MAX];
    int cg_iterations[SMVS_HOST_LOG_MAX];
    long long active_patch_steps[SMVS_HOST_LOG_MAX];   /* sum over the steps */
    double loop_seconds[SMVS_HOST_LOG_MAX];            /* device Newton loop */
    int has_lighting;
    double lighting[16];
} smvs_host_log;

const char *smvs_host_last_error(void);

/* DepthOptimizer(main, subs, bundle, opts).optimize().
 * sgm_depth: optional z-depth map (sgm_w x sgm_h) stored as the "smvs-sgm"
 * embedding; sgm_depth_roundtrip (optional, same size) receives what
 * StereoView::get_sgm_depth() hands to the optimizer. */
int smvs_host_optimize(const smvs_host_view *main_view,
    const smvs_host_view *subs, int n_subs, const smvs_host_bundle *bundle,
    const float *sgm_depth, int sgm_w, int sgm_h, float *sgm_depth_roundtrip,
    const smvs_host_options *opts, float *depth_out, float *normals_out,
    smvs_host_log *log);

/* The embeddings the last smvs_host_optimize of this thread left in its main
 * view (write_depth_to_view / write_image_to_view: the result "smvs" / "smvsN",
 * and with Options::debug_lvl >= 2 -- SMVS_DEBUG_LVL -- the reference's
 * intermediate ones, lib/depth_optimizer.cc:44-45, 68-70, 119-156,
 * depth_optimizer.h:150-160).  names_out: the names, separated by newlines
 * (truncated to cap bytes); returns their number.  smvs_host_embedding copies
 * one (whc = width, height, channels; returns 0, -1 if there is none, or the
 * number of floats needed when cap_floats is too small). */
int smvs_host_embedding_names(char *names_out, int cap);
int smvs_host_embedding(const char *name, float *out, long long cap_floats, int *whc);

/* Loop recording (measurement, see RecordedLoop in depth_optimizer.h):
 * smvs_host_record_loops(1) drops what this thread recorded before (the clones
 * are destroyed) and starts recording, (0) stops; every smvs_host_optimize of
 * the thread in between leaves one entry per Newton batch.
 * smvs_host_recorded_loops copies up to `cap` entries -- context handle, loop
 * parameters (a smvs_gn_loop_params each), scale, iteration -- and returns how
 * many were recorded.  The contexts belong to the recorder until
 * smvs_host_release_recorded_loops: destroy != 0 destroys them, 0 hands them
 * to the caller (who ends each with smvs_ctx_destroy). */
int smvs_host_record_loops(int on);
int smvs_host_recorded_loops(void **ctxs, void *params, int *scales, int *iters,
    int cap);
int smvs_host_release_recorded_loops(int destroy);

/* reconstruct_sgm_depth_for_view (app/smvsrecon.cc:346-384): returns the
 * merged z-depth map at SGM resolution ((w+1)>>scale ...). */
int smvs_host_sgm_depth(const smvs_host_view *main_view,
    const smvs_host_view *subs, int n_subs, const smvs_host_bundle *bundle,
    int sgm_scale, float min_depth, float max_depth, int device,
    float *depth_out, int *out_w, int *out_h);

/* smvs::Surface on its own (lib/surface.cc): Surface::create (from the bundle
 * when init_depth is NULL, else from the W x H depth map) followed by a
 * script of operations -- 1 expand, 2 subdivide_patches,
 * 3 fill_patches_from_depth, 4 remove_isolated_patches, 5 delete every
 * delete_every-th valid patch + remove_nodes_without_patch.  No device
 * involved.  info = { scale, npx, npy, start_x, start_y }; the arrays are
 * caller-sized (for the finest scale the script reaches). */
int smvs_host_surface_script(const smvs_host_view *main_view,
    const smvs_host_bundle *bundle, const float *init_depth, int init_scale,
    const int *ops, int n_ops, int delete_every, int *info, double *nodes_out,
    uint8_t *node_valid_out, uint8_t *patch_valid_out);
/* The same script on the surface of a device context (include/smvs_hip.h,
 * "grid surgery"); info[5] = valid patches as the last operation reported. */
int smvs_host_surface_script_device(const smvs_host_view *main_view,
    const smvs_host_bundle *bundle, const float *init_depth, int init_scale,
    const int *ops, int n_ops, int delete_every, int device, int *info6,
    double *nodes_out, uint8_t *node_valid_out, uint8_t *patch_valid_out);

/* The per-view tasks of smvsrecon (app/smvsrecon.cc:658-733) for n_jobs
 * reference views on a smvs_amd::ViewQueue: per job StereoView::create for the
 * main view and its neighbours, optionally reconstruct_sgm_depth_for_view
 * (sgm_scale >= 0; a negative value skips the SGM front end and initialises
 * from the bundle), DepthOptimizer(...).optimize(), depth + normal maps.
 * Job i uses mains[i] and subs[i * n_subs .. (i + 1) * n_subs).  Worker w runs
 * on device first_device + w % num_devices with views_in_flight workers per
 * device.  Outputs (each may be NULL): depth_out / normals_out of job
 * `keep_job` only (W * H and W * H * 3 floats of that job's main view),
 * job_seconds[n_jobs] = wall time of each task, *total_seconds = wall time
 * from the first task queued to the last finished, logs[n_jobs]. */
int smvs_host_optimize_views(const smvs_host_view *mains,
    const smvs_host_view *subs, int n_jobs, int n_subs,
    const smvs_host_bundle *bundle, const smvs_host_options *opts, int sgm_scale,
    int first_device, int num_devices, int views_in_flight, int keep_job,
    float *depth_out, float *normals_out, double *job_seconds,
    double *total_seconds, smvs_host_log *logs);

/* The reference's own Newton-step calls through the compatibility classes
 * (lib/depth_optimizer.cc:225-262): Surface::create(bundle, main, init_scale),
 * StereoView::set_scale(scale of the surface) on every view (host planes),
 * GaussNewtonStep(opts, main, subs, Mi, ti).construct(surface, subsurfaces =
 * every neighbour for every patch, active = every node, lighting = NULL, &H,
 * &g, &P), then ConjugateGradient({200, 0.01 |g|, 1e-3}).solve(H, -g, &x, &P).
 * Outputs (caller-sized, each may be NULL): the planes the step used --
 * main_grad[W*H*2], sub_grad[n_subs][W*H*2], sub_hess[n_subs][W*H*3] -- the
 * reprojections Mi[n_subs*9], ti[n_subs*3], flen2 = { flen, inverse flen }, and
 * g[4N], x[4N], H9[N*9*16], P[N*16], cg = { iterations, info }. */
int smvs_host_gn_solve_step(const smvs_host_view *main_view,
    const smvs_host_view *subs, int n_subs, const smvs_host_bundle *bundle,
    int init_scale, double regularization, int device, float *main_grad,
    float *sub_grad, float *sub_hess, double *Mi, double *ti, float *flen2,
    double *g, double *x, double *H9, double *P, int *cg);

/* smvsrecon's scene-level run (app/smvsrecon.cc:400-745) on an MVE scene
 * directory whose input embedding exists as <image_embedding>.mvei: view list,
 * ViewSelection, one ViewQueue task per reference view, result embeddings
 * written into the view directories.  view_ids may be NULL (every view).
 * reconstructed_out[max_reconstructed] receives the ids of the reconstructed
 * views (may be NULL with max_reconstructed = 0); *n_reconstructed is their
 * number -- when it exceeds max_reconstructed only the first
 * max_reconstructed ids were stored (the scene has been reconstructed; the
 * caller can size the buffer from smvs_host_scene_info). */
typedef struct {
    const char *image_embedding;    /* "undistorted" */
    float regularization;           /* alpha, 1.0 */
    int output_scale;               /* -o */
    int use_shading, use_sgm, force_recon, force_sgm, full_optimization;
    float sgm_min, sgm_max;
    int sgm_scale;
    int num_neighbors, min_neighbors;
    int first_device, num_devices, views_in_flight;
    int input_scale;                /* -s; < 0: automatic (app/smvsrecon.cc:477-500) */
    int max_pixels;                 /* --max-pixels, 1700000 */
} smvs_host_recon_settings;
int smvs_host_reconstruct_scene(const char *scene_dir,
    const smvs_host_recon_settings *settings, const int *view_ids, int n_view_ids,
    int *reconstructed_out, int max_reconstructed, int *n_reconstructed,
    int *n_skipped, double *seconds, int *input_scale_used);

/* Byte-image containers of a view directory (csrc/host/png_io.cc,
 * scene_io.cc): load `path` (.png or .mvei, u8) -> width, height, channels and,
 * if pixels != NULL with room for capacity bytes, the interleaved data;
 * save a u8 image as PNG; mve::image::rescale_half_size_gaussian<uint8_t>
 * (out sized ((w + 1) / 2) * ((h + 1) / 2) * c). */
int smvs_host_load_byte_image(const char *path, int *whc, uint8_t *pixels,
    size_t capacity);
int smvs_host_save_png(const char *path, const uint8_t *pixels, int width, int height,
    int channels);
int smvs_host_rescale_half_size_gaussian(const uint8_t *pixels, int width, int height,
    int channels, uint8_t *out);

/* MVE scene I/O without a device: parses the scene (views/<x>.mve/meta.ini,
 * synth_0.out) -> number of list entries, and per entry (caller-sized arrays of
 * at least max_views): present, flen, rot[9], trans[3], and the size of the
 * image embedding (0 x 0 when missing); *n_features of the bundle (-1: none).
 * mvei_roundtrip: loads in_path (u8 or float .mvei) and saves it to out_path. */
int smvs_host_scene_info(const char *scene_dir, const char *image_embedding,
    int max_views, int *n_views, int *present, float *flen, float *rot9,
    float *trans3, int *width, int *height, int *n_features);
int smvs_host_mvei_roundtrip(const char *in_path, const char *out_path);

/* smvs_amd::ViewQueue on its own (no device involved): n_tasks tasks that
 * record the slot they ran on; task `throwing_task` (or -1) throws.
 * device_hist[num_devices] and worker_hist[num_devices * views_in_flight]
 * receive the number of tasks per device / worker.  Returns 0 when every task
 * ran exactly once, the throwing task's exception arrived through its future
 * and no other future carried one. */
int smvs_host_view_queue_selftest(int n_tasks, int num_devices,
    int views_in_flight, int throwing_task, int *device_hist, int *worker_hist);

/* DepthOptimizer(main, subs, Surface::Ptr, opts) followed by get_depth() /
 * get_normals() WITHOUT optimize() (lib/depth_optimizer.h:53-61): the maps of
 * the surface Surface::create builds (from the bundle, or from init_depth). */
int smvs_host_surface_maps(const smvs_host_view *main_view,
    const smvs_host_view *subs, int n_subs, const smvs_host_bundle *bundle,
    const float *init_depth, int init_scale, int device, float *depth_out,
    float *normals_out);

/* Host-only pieces of the SGM front end, for CPU tests (no device involved):
 *   smvs_host_depth_range: SGMStereo::fill_depth_range_for_view
 *     (lib/sgm_stereo.cc:669-720) -> range[2];
 *   smvs_host_reprojection: CameraInfo::fill_reprojection from `source` to
 *     `destination` at their image sizes -> M[9], t[3] (lib/sgm_stereo.cc:56-62);
 *   smvs_host_sgm_image: StereoView::get_byte_image (desaturate<uint8_t>,
 *     lib/stereo_view.cc:86-95) followed by `halvings` rescale_half_size
 *     (lib/sgm_stereo.cc:31-39) -> out (caller-sized), *out_w, *out_h. */
int smvs_host_depth_range(const smvs_host_view *view,
    const smvs_host_bundle *bundle, float *range2);
int smvs_host_reprojection(const smvs_host_view *source,
    const smvs_host_view *destination, float *M9, float *t3);
int smvs_host_sgm_image(const smvs_host_view *view, int halvings, uint8_t *out,
    int *out_w, int *out_h);

/* smvs::ViewSelection(opts, views, bundle).get_neighbors_for_view(view)
 * (lib/view_selection.cc:14-161; bundle may be NULL: position-based).
 * A view whose `bytes` is NULL has no image in the embedding; width <= 0
 * marks a hole in the view list (a null View::Ptr).  out: room for n_views
 * indices; *n_out receives how many were written. */
int smvs_host_select_neighbors(const smvs_host_view *views, int n_views,
    const smvs_host_bundle *bundle, int view, int num_neighbors, int *out,
    int *n_out);

/* smvs_amd::BlockStencilMatrix::multiply (BlockSparseMatrix<4>::multiply,
 * lib/block_sparse_matrix.h:276-298) on the host, no device involved:
 * y = A x for the block stencil blocks9[num_nodes][9][16]. */
int smvs_host_block_multiply(int num_nodes, int node_stride, const double *blocks9,
    const double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
