/*
 * smvs_hip.h -- C ABI of the MI355X (gfx950) depth-optimisation hot path.
 *
 * This is the drop-in boundary: a host DepthOptimizer (ours in
 * smvs_amd/csrc/host/, or the reference's lib/depth_optimizer.cc with the
 * binding shown in INTEGRATION.md) calls these entry points instead of
 * GaussNewtonStep / ConjugateGradient / Surface::update_nodes /
 * LightOptimizer / SGMStereo.  file:line citations are relative to the
 * flanggut/smvs tree and name the reference interface each entry replaces.
 *
 * Conventions
 *  - every function returns 0 (SMVS_OK) or a negative smvs_status; nothing
 *    throws; smvs_last_error() gives the text of the last failure of the
 *    calling thread.  The reference throws std::invalid_argument on
 *    dimension mismatch (block_sparse_matrix.h:136-138, sse_vector.cc:22-23,
 *    conjugate_gradient.h:77-78): the host wrapper re-throws from the status.
 *  - pointers are caller-owned HOST memory unless the name ends in _dev.
 *  - images use MVE's layout: interleaved channels, row-major,
 *    index (y*W + x)*C + c.
 *  - one context is used by one host thread at a time; different contexts
 *    may be used concurrently (one reference view per context, as the
 *    reference's one-task-per-view pool, app/smvsrecon.cc:658-733).
 *  - a context owns its device buffers and one HIP stream.
 */
#ifndef SMVS_HIP_H
#define SMVS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smvs_ctx smvs_ctx;

typedef enum {
    SMVS_OK = 0,
    SMVS_ERR_INVALID = -1,   /* bad argument / dimension mismatch */
    SMVS_ERR_HIP = -2,       /* HIP runtime error (see smvs_last_error) */
    SMVS_ERR_STATE = -3,     /* call order violated (e.g. solve before construct) */
    SMVS_ERR_NOMEM = -4
} smvs_status;

/* ConjugateGradient::ReturnInfo, conjugate_gradient.h:22-27 */
typedef enum {
    SMVS_CG_CONVERGENCE = 0,
    SMVS_CG_MAX_ITERATIONS = 1,
    SMVS_CG_INVALID_INPUT = 2
} smvs_cg_info;

#define SMVS_MAX_SUBS 16

const char *smvs_last_error(void);
/* Number of HIP devices visible (0 when none / no driver). */
int smvs_device_count(void);

/* ------------------------------------------------------------------ */
/* context                                                            */
/* ------------------------------------------------------------------ */

/* One reference view (StereoView main + n_subs neighbours,
 * depth_optimizer.h:48-51).  width/height = main view size. */
int smvs_ctx_create(int device, int width, int height, int n_subs,
    smvs_ctx **out);
/* Parks the context (its device buffers and stream stay allocated) for the
 * next smvs_ctx_create of the same device / size / neighbour count: a view's
 * few hundred MB are not returned to the driver between views.
 * smvs_release_workspaces() frees the parked contexts. */
int smvs_ctx_destroy(smvs_ctx *ctx);
int smvs_ctx_synchronize(smvs_ctx *ctx);

/* Which implementation of ConjugateGradient::solve (conjugate_gradient.h:72-202)
 * the context uses -- not in the reference, which has one.  All of them run
 * the same preconditioned CG with the same termination rules (:136-198):
 *   AUTO          the fastest one that applies: the chip-resident solver when
 *                 the node grid fits the chip (<= 256 tiles of <= 512 nodes),
 *                 with ONE grid-wide exchange per iteration (eight dot
 *                 products of the current vectors are reduced before the
 *                 step length is known; r.r, z.r, x.(b + r) of the updated
 *                 vectors follow from them one step ahead); otherwise the
 *                 streaming kernels
 *   STREAMING     assembly kernel + two launches per iteration (csrc/cg.hip);
 *                 any grid size
 *   RESIDENT_REF  the chip-resident solver with the reference's operation
 *                 order: d.Ad first, then r.r, z.r, x.(b + r) of the updated
 *                 vectors (two exchanges per iteration); falls back to
 *                 STREAMING when the grid does not fit
 * A mode that cannot run (co-residency lost, grid too large) degrades to
 * STREAMING; the choice never changes results beyond the summation order. */
typedef enum {
    SMVS_SOLVER_AUTO = 0,
    SMVS_SOLVER_STREAMING = 1,
    SMVS_SOLVER_RESIDENT_REF = 2
} smvs_solver_mode;
int smvs_ctx_set_solver(smvs_ctx *ctx, int mode);

/* DepthOptimizer::prepare_correspondences, depth_optimizer.cc:679-699:
 * Mi[n_subs][9] row-major and ti[n_subs][3], already widened from the float
 * CameraInfo::fill_reprojection result; flen / inv_flen =
 * StereoView::get_flen / get_inverse_flen (stereo_view.h:132-148). */
int smvs_ctx_set_cameras(smvs_ctx *ctx, const double *Mi, const double *ti,
    float flen, float inv_flen);

/* Per-scale planes of the main view, StereoView::get_image_gradients /
 * get_shading_image / get_shading_gradients (stereo_view.h:43-47).
 * shading / shading_grad may be NULL (no -S). */
int smvs_ctx_upload_main(smvs_ctx *ctx, const float *grad2,
    const float *shading1, const float *shading_grad2);
/* Per-scale planes of neighbour `sub`: get_image_gradients (2 ch) and
 * get_image_hessian (3 ch: I_xx, I_xy, I_yy), stereo_view.cc:167-187. */
int smvs_ctx_upload_sub(smvs_ctx *ctx, int sub, int width, int height,
    const float *grad2, const float *hess3);

/* Scale space on the device: what StereoView's constructor
 * and StereoView::set_scale compute on the host (stereo_view.cc:16-46, 97-188).
 * smvs_ctx_upload_image stores the u8 image of the main view (view = -1) or of
 * neighbour `view` (interleaved channels, 1 or 3).  smvs_ctx_set_scale then
 * produces, for every view with an image, the gradient / Hessian planes of
 * that scale (byte -> float, Gaussian blur with sigma = 0.12 * 2^scale + 0.2,
 * luminance, 3x3 quadratic fit) directly in the context: nothing crosses PCIe
 * per scale.  smvs_ctx_download_planes hands planes back to host-side
 * topology code; hess3 may be NULL (the main view keeps no Hessian). */
int smvs_ctx_upload_image(smvs_ctx *ctx, int view, int width, int height,
    int channels, const uint8_t *bytes);
/* The same without waiting for the transfer (StereoView::create x 9 at the
 * start of a view's task, app/smvsrecon.cc:662-691, costs the reference
 * nothing on the device; here it is 56 MB over PCIe): with `bytes` in
 * page-locked memory (smvs_pinned_alloc) the DMA is enqueued on a copy stream
 * of the context and the call returns at once; the byte -> float conversion
 * runs where the image is first needed (smvs_ctx_set_scale converts view by
 * view, so a view's blur and gradients overlap the next views' transfers).
 * `bytes` must stay valid and unchanged until smvs_ctx_synchronize / the
 * context's destruction.  Pageable memory: identical to smvs_ctx_upload_image. */
int smvs_ctx_upload_image_async(smvs_ctx *ctx, int view, int width, int height,
    int channels, const uint8_t *bytes);
int smvs_ctx_set_scale(smvs_ctx *ctx, int scale);
int smvs_ctx_download_planes(smvs_ctx *ctx, int view, float *grad2,
    float *hess3);
/* shading image + gradients alone (scale independent, stereo_view.cc:64-84) */
int smvs_ctx_upload_shading(smvs_ctx *ctx, const float *shading1,
    const float *shading_grad2);

/* Surface state (surface.h:106-120) as flat arrays.
 *   nodes[(npx+1)*(npy+1)][4] = f, dx, dy, dxy (patch units);
 *   node_valid / patch_valid: non-null node / patch;
 *   patch_vis[p] bit j <=> j in subsurfaces[p] (depth_optimizer.h:108).
 * Resets the active set to "all valid nodes" (depth_optimizer.cc:204-212). */
int smvs_ctx_set_surface(smvs_ctx *ctx, int scale, int npx, int npy,
    int start_x, int start_y, const double *nodes, const uint8_t *node_valid,
    const uint8_t *patch_valid, const uint32_t *patch_vis);
/* active_nodes (std::vector<char>, depth_optimizer.cc:205); NULL = all valid */
int smvs_ctx_set_active(smvs_ctx *ctx, const uint8_t *active);
int smvs_get_active(smvs_ctx *ctx, uint8_t *active, int *num_active);
int smvs_get_nodes(smvs_ctx *ctx, double *nodes);
int smvs_set_nodes(smvs_ctx *ctx, const double *nodes);
/* Device-resident copy of the nodes (not in the reference, whose surface lives
 * in host memory): save keeps the current nodes in HBM, restore brings them
 * back on the context's stream without a transfer -- a caller that reruns the
 * optimisation from one start (bench.py, parameter sweeps) uploads it once.
 * restore without a save, or after smvs_ctx_set_surface changed the grid,
 * returns SMVS_ERR_STATE. */
int smvs_ctx_save_nodes(smvs_ctx *ctx);
int smvs_ctx_restore_nodes(smvs_ctx *ctx);
/* A second context that holds everything the Newton loop of `src` reads at
 * this moment -- cameras, the scale's gradient / Hessian planes of all views,
 * the surface (nodes, validity, visibility masks), the lighting -- copied on
 * the device, with its nodes saved (smvs_ctx_restore_nodes).  Not in the
 * reference: it is how bench.py keeps the start state of every Newton batch of
 * one DepthOptimizer::optimize (lib/depth_optimizer.cc:219-304 is entered 15
 * times per view) resident in HBM and replays exactly those loops as its timed
 * region.  The clone is an ordinary context (smvs_ctx_destroy). */
int smvs_ctx_clone_loop_state(smvs_ctx *src, smvs_ctx **out);

/* ------------------------------------------------------------------ */
/* Gauss-Newton step                                                  */
/* ------------------------------------------------------------------ */

/* GaussNewtonStep::construct, gauss_newton_step.cc:33-143, with
 * GaussNewtonStep::Options {regularization, light_surf_regularization}
 * (gauss_newton_step.h:28-34); lighting16 = GlobalLighting::Params or NULL.
 * H, g, P stay on the device.  num_active_patches (may be NULL) = patches
 * with at least one active node (gauss_newton_step.cc:73-79). */
int smvs_gn_construct(smvs_ctx *ctx, double regularization,
    double light_surf_regularization, const double *lighting16,
    int *num_active_patches);

/* Test / parity hooks: download the assembled system.  H9[n][s][16]: 4x4
 * row-major block (row node n, col node n + dy*(npx+1) + dx), slot
 * s = (dy+1)*3 + dx+1; zero where the reference holds no block.
 * g[4n+k]; P[n][16] inverted diagonal blocks.  Any pointer may be NULL. */
int smvs_gn_download(smvs_ctx *ctx, double *H9, double *g, double *P);
/* per-patch 16x16 systems and 16-gradients before assembly
 * (sub_hessian / sub_gradient, gauss_newton_step.cc:61-62); entries of
 * patches that were not evaluated are unspecified.  The device keeps the
 * upper triangle (the part the assembly reads, gauss_newton_step.cc:103,
 * 113-119); the lower one comes back as its mirror. */
int smvs_gn_download_patch_systems(smvs_ctx *ctx, double *Hp, double *gp);
/* Overwrite the assembled system (solver tests on arbitrary systems). */
int smvs_gn_upload(smvs_ctx *ctx, const double *H9, const double *g,
    const double *P);

/* ConjugateGradient::solve, conjugate_gradient.h:72-202, on b = -g with the
 * block-Jacobi preconditioner, x0 = 0.  error_tolerance < 0 selects the
 * optimizer's rule error_tolerance = 0.01 * ||g|| (depth_optimizer.cc:247).
 * num_iterations / info as ConjugateGradient::Status. */
int smvs_cg_solve(smvs_ctx *ctx, int max_iterations, double error_tolerance,
    double q_tolerance, int *num_iterations, int *info);
int smvs_cg_download_x(smvs_ctx *ctx, double *x);
int smvs_cg_upload_x(smvs_ctx *ctx, const double *x);

/* depth_optimizer.cc:267-303: NaN guard on delta[0], fill_node_reprojections,
 * Surface::update_nodes (surface.cc:957-981), fill_node_reprojections, then
 * either the mean reprojection delta (full_optimization) or the new active
 * set (node active if any pixel of an incident patch moved by more than
 * `threshold` = 0.15).  When nan_flag is set nothing was updated. */
int smvs_update_and_reactivate(smvs_ctx *ctx, double threshold,
    int full_optimization, int *num_active, double *mean_delta,
    int *nan_flag);

/* The whole Newton loop of one outer iteration, depth_optimizer.cc:204-304:
 * four launches per step, no host round trip inside a step, and the next step
 * is enqueued before the previous one has ended (the loop condition of
 * :219-220 / :267-268 / :284-288 is evaluated on the device).
 * The call returns when the loop has ended and the stream is idle.  Between
 * the call's steps nothing is materialised for smvs_gn_download (H, g, P live
 * in registers); smvs_cg_download_x holds the last step's delta. */
typedef struct {
    double regularization;             /* DepthOptimizer::Options */
    double light_surf_regularization;
    int full_optimization;
    int max_newton_steps;              /* 200, depth_optimizer.cc:219 */
    int cg_max_iterations;             /* 200, :246 */
    double cg_q_tolerance;             /* 1e-3, conjugate_gradient.h:34 */
    double active_threshold;           /* 0.15, :296 */
    double full_opt_threshold;         /* 0.01, :285 */
    int use_lighting;                  /* lighting != nullptr */
    double lighting[16];
    int reset_active;                  /* 1: start from all valid nodes */
} smvs_gn_loop_params;

typedef struct {
    int newton_steps;
    int linear_iterations;             /* sum of CG iterations, :257 */
    long long active_patch_steps;      /* sum over steps of active patches */
    int final_active_nodes;
    int nan_break;                     /* loop left through :267-268 */
} smvs_gn_loop_stats;

int smvs_gn_run_loop(smvs_ctx *ctx, const smvs_gn_loop_params *params,
    smvs_gn_loop_stats *stats);

/* ------------------------------------------------------------------ */
/* surface outputs + lighting                                         */
/* ------------------------------------------------------------------ */

/* Surface::get_depth_map / get_normal_map, surface.cc:155-183 (float
 * images, zero where no patch). depth[W*H], normals[W*H*3]. */
int smvs_get_depth_map(smvs_ctx *ctx, float *depth);
int smvs_get_normal_map(smvs_ctx *ctx, float *normals);

/* Both maps in one pass over the surface (one kernel, two transfers, one
 * synchronisation) -- the end of DepthOptimizer::optimize,
 * depth_optimizer.cc:150-160.  inv_calibration9 != NULL: the depth map comes
 * back as StereoView::write_depth_to_view stores it (stereo_view.h:100-119:
 * MVE's ray-length convention, depth times the length of the pixel's viewing
 * ray under CameraInfo::fill_inverse_calibration); NULL: z-depth, as
 * smvs_get_depth_map. */
int smvs_get_maps(smvs_ctx *ctx, const float *inv_calibration9, float *depth,
    float *normals);

/* Page-locked host memory for the large buffers that cross the boundary (the
 * views' u8 images, the 33 MB of depth + normal maps per 1920x1080 view): a
 * transfer from / to such a buffer is one DMA at link rate, a pageable buffer
 * goes through a staging copy on the host (3-4 ms per view at 1920x1080 x 9).
 * Not in the reference, whose images live in pageable mve::Image storage; any
 * pointer works everywhere, pinned ones are just faster.  Buffers are pooled
 * by size (page-locking costs more than the copy it saves) and go back to the
 * driver with smvs_release_workspaces(). */
int smvs_pinned_alloc(size_t bytes, void **out);
int smvs_pinned_free(void *ptr);

/* LightOptimizer::fit_lighting_to_image accumulation,
 * light_optimizer.cc:32-49, over the uploaded shading image: A[16][16],
 * b[16] (the 16x16 pseudo inverse stays on the host).  The _dev variant
 * leaves 272 doubles (A then b) in a device buffer for an optional RCCL
 * all-reduce ("shared lighting" extension) and returns its
 * device pointer. */
int smvs_light_accumulate(smvs_ctx *ctx, double *A256, double *b16);
int smvs_light_accumulate_dev(smvs_ctx *ctx, double **Ab272_dev);
/* Reads the context's 272-double buffer back (after the caller's collective
 * has summed it in place over the views that share their lighting). */
int smvs_light_download(smvs_ctx *ctx, double *A256, double *b16);
/* The other direction: puts normal equations into the context's buffer -- sums
 * a caller has kept (a lock-step round continued later), or a known pattern
 * (bench.py checks the RCCL all-reduce of smvs_rccl.h with one). */
int smvs_light_upload(smvs_ctx *ctx, const double *A256, const double *b16);

/* ------------------------------------------------------------------ */
/* SGM                                                                */
/* ------------------------------------------------------------------ */

/* SGMStereo::run_sgm, sgm_stereo.cc:98-124, for one (main, neighbour)
 * pair of u8 images already at SGM scale (sgm_stereo.cc:31-39):
 * census cost volume over num_steps inverse-depth planes, 8-path
 * aggregation (constant P2, the SSE branch :361-406), WTA.
 * M[9], t[3]: float reprojection main -> neighbour at SGM resolution
 * (sgm_stereo.cc:154-160).  Outputs (any may be NULL):
 *   depth[w*h]  (sgm_stereo.cc:274-306),
 *   argmin[w*h] winning plane index,
 *   cost[w*h*num_steps], sgm[w*h*num_steps] u16 volumes (parity tests). */
int smvs_sgm_run(int device, const uint8_t *main_img, int w, int h,
    const uint8_t *neighbor_img, int nw, int nh, const float *M,
    const float *t, float min_depth, float max_depth, int num_steps,
    uint16_t penalty1, uint16_t penalty2, float *depth, int32_t *argmin,
    uint16_t *cost, uint16_t *sgm);

/* The whole SGM initialisation of one reference view on the device:
 * reconstruct_sgm_depth_for_view, app/smvsrecon.cc:346-384, i.e. for each of
 * the first one or two neighbours SGMStereo::reconstruct (sgm_stereo.cc:46-96:
 * run_sgm main -> neighbour, run_sgm neighbour -> main, left/right
 * consistency check :64-91 with integer pixel coordinates, 3 % border and
 * depth ratio 0.8) and the merge of the two checked maps (:366-377).  The
 * four cost volumes, the depth maps and the check never leave the device;
 * only the merged z-depth map (w*h floats) is copied back.
 * Images: u8, one channel, already at SGM scale (SGMStereo's constructor,
 * sgm_stereo.cc:27-39).  M_fwd / t_fwd: CameraInfo::fill_reprojection main ->
 * neighbour at the two SGM image sizes (also the matrix of the check, :56-62);
 * M_bwd / t_bwd: neighbour -> main.  range_main / range_neighbor: {min, max}
 * depth of the two runs (SGMStereo::fill_depth_range_for_view, :669-720, or
 * the caller's fixed range). */
typedef struct {
    const uint8_t *image;
    int width, height;
    float M_fwd[9], t_fwd[3];
    float M_bwd[9], t_bwd[3];
    float range_main[2];
    float range_neighbor[2];
} smvs_sgm_neighbor;

int smvs_sgm_depth_for_view(int device, const uint8_t *main_img, int w, int h,
    const smvs_sgm_neighbor *neighbors, int n_neighbors, int num_steps,
    uint16_t penalty1, uint16_t penalty2, float *depth);

/* The same with SGMStereo's constructor (sgm_stereo.cc:27-39) on the device as
 * well: the images are the views' FULL-resolution u8 embeddings (interleaved,
 * `channels` / neighbor_channels[k] = 1 or 3); StereoView::get_byte_image
 * (desaturate<uint8_t>, stereo_view.cc:86-95) and `halvings` x
 * mve::image::rescale_half_size run on the device.  w, h and the neighbours'
 * width / height are the full-resolution sizes; the reprojections, the depth
 * ranges and the output map refer to the SGM-scale sizes ((s + 1) >> 1 per
 * halving). */
int smvs_sgm_depth_for_view_raw(int device, const uint8_t *main_img, int w, int h,
    int channels, const smvs_sgm_neighbor *neighbors, const int *neighbor_channels,
    int n_neighbors, int halvings, int num_steps, uint16_t penalty1,
    uint16_t penalty2, float *depth);

/* DepthOptimizer::depthmap_bilateral_filter, depth_optimizer.cc:957-1004 */
int smvs_bilateral_upsample(int device, const float *dm, int dm_w, int dm_h,
    const float *ci, int w, int h, int channels, float sigma,
    int kernel_size, float *out);

/* The same filter inside a view's context (DepthOptimizer::create_initial_surface,
 * depth_optimizer.cc:35-51: init = depthmap_bilateral_filter(sgm_depth,
 * main_view->get_image())): guided by the main image smvs_ctx_upload_image
 * left on the device (SMVS_ERR_STATE without one), result written to
 * out[W*H] (may be NULL) AND kept in the context as the SGM depth map the
 * visibility tests of smvs_topology_subviews compare with -- the reference
 * hands the same filtered map to both (:41-45, :463-466).  dm == NULL forgets
 * the resident map. */
int smvs_ctx_sgm_init_depth(smvs_ctx *ctx, const float *dm, int dm_w, int dm_h,
    float sigma, int kernel_size, float *out);

/* The same with the map as the view stores it: dm_mve[dm_w*dm_h] is the
 * "smvs-sgm" embedding in MVE's ray-length convention
 * (StereoView::write_depth_to_view, stereo_view.h:100-119) and
 * inv_calibration9 the inverse calibration of an image of the map's size
 * (StereoView::get_sgm_depth, stereo_view.h:121-135:
 * mve::image::depthmap_convert_conventions(depth, invproj, false)).  The
 * conversion to z-depth runs on the device with the host's float operations
 * (same bits); a caller that would otherwise convert 0.5 M pixels on the host
 * while the GPU waits for the map calls this one (create_initial_surface,
 * depth_optimizer.cc:35-45).  dm_is_z_depth != 0: dm is still the z-depth map
 * the SGM front end produced (reconstruct_sgm_depth_for_view,
 * app/smvsrecon.cc:346-384) and has not been stored yet: the kernel applies
 * write_depth_to_view's conversion (`dm *= len`) and then get_sgm_depth's
 * (`dm *= 1.0 / len`) -- the round trip through the embedding the reference
 * makes, same bits, without the host touching 2 x 0.5 M pixels. */
int smvs_ctx_sgm_init_depth_mve(smvs_ctx *ctx, const float *dm, int dm_w, int dm_h,
    const float *inv_calibration9, int dm_is_z_depth, float sigma, int kernel_size, float *out);

/* ------------------------------------------------------------------ */
/* topology tests between Newton batches              */
/* ------------------------------------------------------------------ */

/* The per-(patch, neighbour) part of DepthOptimizer::create_subview_surfaces,
 * depth_optimizer.cc:433-590, for the surface of smvs_ctx_set_surface and the
 * images of smvs_ctx_upload_image: z-buffer of the surface depth map (and of
 * sgm_depth[W*H]; NULL: the map smvs_ctx_sgm_init_depth left in the context,
 * if any, else none; the reference splats it when use_sgm is set, :463-466)
 * in every neighbour, then border / occlusion test (:505-530),
 * warp anisotropy <= 8 (:532-575) and, if use_ncc (the reference's !use_sgm,
 * :577-580), ncc_for_patch >= 0 (:792-912).
 * patch_vis_out[num_patches] (may be NULL: nothing is copied back): bit j =
 * patch visible in neighbour j; 0 for invalid patches.  The mask also becomes
 * the context's patch visibility.
 * Deleting the patches without any neighbour (:592-603) stays with the
 * caller, which owns the topology. */
int smvs_topology_subviews(smvs_ctx *ctx, const float *sgm_depth, int use_ncc,
    uint32_t *patch_vis_out);

/* DepthOptimizer::mse_for_patch, depth_optimizer.cc:747-790, for every valid
 * patch with the visibility set by smvs_ctx_set_surface: mean gradient
 * mismatch against the visible neighbours (1.0 without any); -1 for invalid
 * patches.  cut_boundaries (:360-431) thresholds it at 0.05. */
int smvs_topology_patch_mse(smvs_ctx *ctx, double *mse_out);

/* The `while (deleted > 10) cut_boundaries()` loops of the optimiser
 * (depth_optimizer.cc:186-190, 323-337; cut_boundaries :360-431) on the
 * surface of smvs_ctx_set_surface (incl. its visibility masks): every pass
 * deletes patches with a depth discontinuity (:371-399) and patches that have
 * a node with more than one missing neighbour node and mse_for_patch > 0.05
 * (:401-428), then nodes without a patch (Surface::remove_nodes_without_patch).
 * inv_calibration9: CameraInfo::fill_inverse_calibration of the main view.
 * patch_valid_out[num_patches], node_valid_out[num_nodes] (either may be NULL:
 * not copied back): validity after the last pass (also the context's);
 * *total_deleted (may be NULL). */
int smvs_topology_cut_boundaries(smvs_ctx *ctx, const float *inv_calibration9,
    uint8_t *patch_valid_out, uint8_t *node_valid_out, int *total_deleted);

/* ------------------------------------------------------------------ */
/* grid surgery of the surface on the device          */
/* ------------------------------------------------------------------ */

/* smvs::Surface's topology operations (lib/surface.h:36-57) on the surface the
 * context holds, so that DepthOptimizer::optimize (depth_optimizer.cc:53-162)
 * never moves the surface across PCIe between its Newton batches.  The grid
 * geometry follows the reference's integer rules (surface.cc:28-37, 983-1012)
 * and is mirrored on the host by smvs_surface_info; node values are the
 * reference's double arithmetic in its operation order.  Every call leaves the
 * number of valid (non-null) patches in *num_valid_patches (may be NULL) --
 * the one quantity the optimiser's outer loop needs (:339-356) -- and resets
 * the visibility masks when the grid changed.  A call that is asked for no
 * count (every output pointer NULL) only ENQUEUES its kernels on the
 * context's stream and returns without waiting for the device: the sequence
 * between two Newton batches costs one synchronisation, with its last call. */
typedef struct {
    int scale, patchsize;      /* patchsize = 2^scale */
    int npx, npy;              /* patches; nodes = (npx + 1) * (npy + 1) */
    int start_x, start_y;      /* pixel of node (0, 0) */
} smvs_surface_geometry;

/* Surface::create, surface.cc:19-53: the grid of `scale` with its nodes
 * initialised from a depth map (initialize_node_from_depth :667-760 for every
 * node, fill_holes :630-651, remove_nodes_without_patch :762-869).  The map --
 * Surface::depth, kept in the context for the fill_patches_from_depth calls
 * that follow every subdivision (depth_optimizer.cc:97-106) -- is
 *   depth != NULL                  depth[W*H], positive = valid (:74-79);
 *   depth == NULL, n_points > 0    zero except the listed pixels: the bundle's
 *                                  features projected into the view by the
 *                                  caller (initialize_depth_from_bundle
 *                                  :90-130), at most one entry per pixel;
 *   depth == NULL, n_points == 0   the filtered SGM map that
 *                                  smvs_ctx_sgm_init_depth left in the context
 *                                  (create_initial_surface,
 *                                  depth_optimizer.cc:41-45). */
int smvs_surface_create(smvs_ctx *ctx, int scale, const float *depth,
    const int32_t *point_pixel, const float *point_depth, int n_points,
    int *num_valid_patches);
/* Surface::fill_patches_from_depth, surface.cc:140-152 */
int smvs_surface_fill_patches_from_depth(smvs_ctx *ctx, int *num_valid_patches);
/* Surface::subdivide_patches, surface.cc:983-1107: scale - 1, five new nodes
 * per patch (a later patch overwrites an earlier one's edge midpoints, as the
 * reference's loop does), old nodes with rescaled derivatives, then
 * fill_holes + remove_nodes_without_patch. */
int smvs_surface_subdivide(smvs_ctx *ctx, int *num_valid_patches);
/* Surface::expand, surface.cc:482-628: two rounds of extrapolated border
 * nodes (check_swap_nodes :472-480), fill_holes, remove_nodes_without_patch.
 * *num_filled (may be NULL) = its return value, the patches filled. */
int smvs_surface_expand(smvs_ctx *ctx, int *num_filled, int *num_valid_patches);
/* Surface::remove_isolated_patches, surface.cc:887-927 (the in-place,
 * column-by-column walk reproduced exactly) + remove_nodes_without_patch. */
int smvs_surface_remove_isolated_patches(smvs_ctx *ctx, int *num_valid_patches);
/* The tail of create_subview_surfaces, depth_optimizer.cc:592-603: patches
 * whose visibility mask (smvs_topology_subviews) is empty are deleted, then
 * remove_nodes_without_patch. */
int smvs_surface_delete_unseen_patches(smvs_ctx *ctx, int *num_deleted,
    int *num_valid_patches);
/* Test hook of the scripted sequences (tests/test_host_surface_cpu.py runs the
 * same scripts on the host mirror): deletes every `every`-th valid patch in
 * id order, then remove_nodes_without_patch. */
int smvs_surface_delete_every(smvs_ctx *ctx, int every, int *num_valid_patches);
/* Geometry of the context's surface (either pointer may be NULL). */
int smvs_surface_info(smvs_ctx *ctx, smvs_surface_geometry *geometry,
    int *num_valid_patches);
/* The arrays of smvs_ctx_set_surface back on the host (any may be NULL):
 * nodes[(npx+1)*(npy+1)][4], node_valid, patch_valid, patch_vis. */
int smvs_surface_download(smvs_ctx *ctx, double *nodes, uint8_t *node_valid,
    uint8_t *patch_valid, uint32_t *patch_vis);

/* ------------------------------------------------------------------ */
/* consumer of the depth / normal maps                */
/* ------------------------------------------------------------------ */

/* MeshGenerator::cut_depth_maps, mesh_generator.cc:24-158, with the normal
 * preparation of generate_mesh (:189-208) and ViewProjection (:300-342): the
 * cross-view consistency cut over all views at once.
 *   depth   : width*height floats, MVE convention (ray length) as stored by
 *             StereoView::write_depth_to_view (the "smvs-B<scale>" embedding);
 *             overwritten with the cut map ("smvs-cut", :229)
 *   normals : width*height*3 floats, camera space as DepthOptimizer writes
 *             them ("smvs-B<scale>N"); overwritten with the world-space normals
 *             (:199-207), which the point export reads afterwards
 *   flen / rot / trans : mve::CameraInfo (world -> camera)
 * With a single view only the normals are transformed (:211). */
typedef struct {
    int width, height;
    float flen;
    float rot[9], trans[3];
    float *depth;
    float *normals;
} smvs_mesh_view;

int smvs_cut_depth_maps(int device, smvs_mesh_view *views, int n_views);

/* The context-free entry points above (smvs_sgm_run, smvs_sgm_depth_for_view,
 * smvs_bilateral_upsample, smvs_cut_depth_maps) draw their device buffers,
 * stream and pinned staging memory from a per-device pool of workspaces that
 * is kept between calls (the reference allocates its volumes per SGMStereo
 * object, lib/sgm_stereo.cc:192-225; a device allocation per call costs more
 * than the kernels).  Concurrent callers get different workspaces.  This
 * returns the pooled memory (workspaces and parked contexts) to the driver;
 * -> number of objects freed. */
int smvs_release_workspaces(void);

/* ------------------------------------------------------------------ */
/* measurement                                                        */
/* ------------------------------------------------------------------ */

/* Kernel classes timed with HIP events on the context's stream. */
enum {
    SMVS_K_PATCH = 0,     /* per-patch J^T W J + g  (K1-K4) */
    SMVS_K_ASSEMBLE,      /* block gather + 4x4 LDL (K5)    */
    SMVS_K_CG_SPMV,       /* block-stencil SpMV + d.Ad (K6) */
    SMVS_K_CG_UPDATE,     /* x, r, z update + reductions (K7/K8) */
    SMVS_K_CG_INIT,
    SMVS_K_REACTIVATE,    /* K9 */
    SMVS_K_MISC,
    SMVS_K_CG_RESIDENT,   /* whole PCG solve in one launch, H in registers */
    SMVS_K_COUNT
};
/* Kernel classes of the context-free front end (SGM, bilateral upsample),
 * timed with HIP events on the call's workspace stream while enabled.
 * smvs_sgm_profile returns the accumulated ms / launches (either may be NULL)
 * and then, if enable >= 0, switches the timing on (1) or off (0) and clears
 * the accumulators; enable < 0 only reads. */
enum {
    SMVS_SGM_K_CENSUS = 0,   /* census of the main image (K11) */
    SMVS_SGM_K_WARP,         /* plane-sweep warp (K12) */
    SMVS_SGM_K_COST,         /* census of the warped planes + Hamming cost (K11, K13) */
    SMVS_SGM_K_PATHS,        /* 8-path aggregation (K14) */
    SMVS_SGM_K_WTA,          /* argmin + depth (K15) */
    SMVS_SGM_K_LR_CHECK,     /* left / right consistency (K16) */
    SMVS_SGM_K_MERGE,        /* two-neighbour merge (K16) */
    SMVS_SGM_K_BILATERAL,    /* joint bilateral upsample (K17) */
    SMVS_SGM_K_COUNT
};
int smvs_sgm_profile(int enable, double *ms, long long *launches);

int smvs_profile_enable(smvs_ctx *ctx, int on);
int smvs_profile_reset(smvs_ctx *ctx);
/* ms[SMVS_K_COUNT] accumulated kernel time, launches[SMVS_K_COUNT] */
int smvs_profile_get(smvs_ctx *ctx, double *ms, long long *launches);

#ifdef __cplusplus
}
#endif
#endif /* SMVS_HIP_H */
