/*
 * smvs_oracle_opt.h -- CPU restatement of DepthOptimizer::optimize and the
 * host-side steps around the Newton loop.  TEST INFRASTRUCTURE ONLY
 * (see smvs_oracle.h; everything here is "parity unpinned").
 */
#ifndef SMVS_ORACLE_OPT_H
#define SMVS_ORACLE_OPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int width, height, channels;   /* u8 image, interleaved channels */
    const uint8_t *bytes;
    float flen;                    /* mve::CameraInfo::flen (normalised) */
    float rot[9], trans[3];        /* world -> camera */
    int view_id;
} orc_view_input;

typedef struct {                   /* mve::Bundle subset */
    int num_features;
    const float *positions;        /* num_features * 3 */
    const int *ref_offsets;        /* num_features + 1 */
    const int *ref_views;          /* view ids */
} orc_bundle;

typedef struct {                   /* DepthOptimizer::Options, depth_optimizer.h:30-42 */
    double regularization;
    double light_surf_regularization;
    int num_iterations;
    int min_scale;
    int use_shading;
    int use_sgm;
    int full_optimization;
    int sgm_width, sgm_height;     /* size of the "smvs-sgm" depth map */
    int gamma_correction;          /* StereoView::create(.., use_shading, gamma_correction),
                                      app/smvsrecon.cc:52, 669: inverse sRGB gamma on the
                                      main view's linear image (with use_shading only) */
} orc_opt_options;

#define ORC_OPT_LOG_MAX 256
typedef struct {
    int count;
    int scale[ORC_OPT_LOG_MAX];
    int iter[ORC_OPT_LOG_MAX];
    int newton_steps[ORC_OPT_LOG_MAX];
    int valid_patches[ORC_OPT_LOG_MAX];
    int cg_iterations[ORC_OPT_LOG_MAX];
    int final_scale, final_patches;
    int has_lighting;
    double lighting[16];
    /* measurement (bench.py's cpu_baseline.optimize), not part of the
     * algorithm: wall time of each batch's Newton loop (construct + solve +
     * update; depth_optimizer.cc:219-304) and the sum over its steps of the
     * patches with an active node (gauss_newton_step.cc:73-79) */
    double loop_seconds[ORC_OPT_LOG_MAX];
    long long active_patch_steps[ORC_OPT_LOG_MAX];
} orc_opt_log;

/* Test access to the topology tests between Newton batches on caller-provided
 * state.  The orc_surface arrays (patch_vis, patch_valid, node_valid) are
 * modified in place like the optimiser's own surface. */
typedef struct {
    int w, h, c;
    const float *image;   /* byte_to_float_image, w*h*c (StereoView::get_image) */
    const float *grad;    /* current scale, w*h*2 */
    float flen;           /* CameraInfo::flen (main view only) */
} orc_topo_view;

/* create_subview_surfaces, depth_optimizer.cc:433-604 (sgm_depth: filtered,
 * full resolution, or NULL) */
void orc_topology_subviews(orc_surface *s, const orc_topo_view *main_view,
    const orc_topo_view *subs, int n_subs, const double *Mi, const double *ti,
    const float *sgm_depth, int use_sgm);
/* mse_for_patch, :747-790, for every valid patch (-1 otherwise) */
void orc_topology_patch_mse(orc_surface *s, const orc_topo_view *main_view,
    const orc_topo_view *subs, int n_subs, const double *Mi, const double *ti,
    double *mse_out);
/* `while (deleted > 10) deleted = cut_boundaries();` (:186-190, :360-431);
 * returns the total number of deleted patches */
int orc_topology_cut_boundaries(orc_surface *s, const orc_topo_view *main_view,
    const orc_topo_view *subs, int n_subs, const double *Mi, const double *ti);

/* depth_optimizer.cc:53-162.  sgm_depth: what StereoView::get_sgm_depth()
 * returns (z-depth, sgm_width x sgm_height) or NULL.  depth_out W*H,
 * normals_out W*H*3 (either may be NULL). */
int orc_optimize(const orc_view_input *main_view, const orc_view_input *subs,
    int n_subs, const orc_bundle *bundle, const float *sgm_depth,
    const orc_opt_options *opts, float *depth_out, float *normals_out,
    orc_opt_log *log);

/* stereo_view.cc:97-188 on a 1-channel float image */
void orc_gradients_and_hessian(const float *input, int w, int h,
    float *gradient2, float *hessian3);
/* CameraInfo::fill_reprojection [MVE-unverified] */
void orc_fill_reprojection(const float *Ks_inv, const float *Rs,
    const float *ts, const float *Kd, const float *Rd, const float *td,
    float *M, float *t);
/* StereoView constructor + set_scale for one u8 image (stereo_view.cc:16-46) */
void orc_scale_planes(const uint8_t *bytes, int w, int h, int c, int scale,
    float *grad2, float *hess3);
/* mve::image::rescale_half_size<uint8_t> [MVE-unverified] */
void orc_rescale_half_size_u8(const uint8_t *in, int w, int h, uint8_t *out);
/* mve::image::rescale_half_size_gaussian<uint8_t> [MVE-unverified M29] (smvs_oracle_front.c) */
void orc_rescale_half_size_gaussian_u8(const uint8_t *in, int w, int h, int c, uint8_t *out);


/* ---- callers either side of the optimiser (smvs_oracle_front.c) ---- */

/* SGMStereo::fill_depth_range_for_view, sgm_stereo.cc:669-720 */
void orc_sgm_depth_range(const orc_bundle *bundle, const orc_view_input *view,
    float *range2);
/* reconstruct_sgm_depth_for_view, app/smvsrecon.cc:346-384: two
 * SGMStereo::reconstruct calls (sgm_stereo.cc:46-96: depth range from the
 * bundle when max_depth == 0, run_sgm both ways, L/R check) and the merge.
 * roundtrip != 0 additionally applies write_depth_to_view + get_sgm_depth
 * (stereo_view.h:108-130), i.e. depth_out is what the optimiser reads.
 * depth_out: ((w+1)>>scale ...) floats; out_w / out_h may be NULL. */
int orc_sgm_depth_for_view(const orc_view_input *main_view,
    const orc_view_input *neighbors, int n_neighbors, const orc_bundle *bundle,
    int sgm_scale, float min_depth, float max_depth, int num_steps,
    int penalty1, int penalty2, int roundtrip, float *depth_out, int *out_w,
    int *out_h);
/* StereoView::get_byte_image + `halvings` rescale_half_size (the SGM input
 * image); out is caller-sized */
int orc_sgm_image(const orc_view_input *view, int halvings, uint8_t *out,
    int *out_w, int *out_h);
/* CameraInfo::fill_reprojection from src to dst at their image sizes */
void orc_view_reprojection(const orc_view_input *src, const orc_view_input *dst,
    float *M9, float *t3);

/* lib/surface.cc on its own: Surface::create (from the bundle when init_depth
 * is NULL) followed by a script of operations -- 1 expand, 2 subdivide_patches,
 * 3 fill_patches_from_depth, 4 remove_isolated_patches, 5 delete every
 * delete_every-th valid patch + remove_nodes_without_patch.  info = { scale,
 * npx, npy, start_x, start_y }; the arrays are caller-sized. */
int orc_surface_script(const orc_view_input *main_view, const orc_bundle *bundle,
    const float *init_depth, int init_scale, const int *ops, int n_ops,
    int delete_every, int *info, double *nodes_out, uint8_t *node_valid_out,
    uint8_t *patch_valid_out);

/* Measurement control for bench.py's cpu_baseline (no counterpart in the
 * reference): n > 0 runs the Newton loop of the first batch of every scale of
 * orc_optimize with n OpenMP threads, independent of orc_set_threads. */
void orc_set_first_batch_loop_threads(int n);

/* what ViewSelection reads of an mve::View (view_selection.cc:23-159) */
typedef struct {
    int present;                   /* 0: a null entry of the scene's view list */
    int id;                        /* mve::View::get_id() */
    float flen;
    float rot[9], trans[3];
    int has_image;                 /* has_image(opts.embedding) */
    int width, height;             /* of that embedding */
} orc_scene_view;

/* smvs::ViewSelection::get_neighbors_for_view (view_selection.cc:14-161):
 * bundle-based with a bundle, position-based with NULL.  out: indices into
 * the view list (the reference's ids-as-indices included), best first;
 * room for n_views entries.  Returns the number of neighbours. */
int orc_select_neighbors(int n_views, const orc_scene_view *views,
    const orc_bundle *bundle, int view, int num_neighbors, int *out);

/* generate_mesh's normal preparation (mesh_generator.cc:189-208) and
 * MeshGenerator::cut_depth_maps (:24-158); depth[i] (ray length) and
 * normals[i] (camera space) are overwritten with the cut maps / world-space
 * normals. */
int orc_cut_depth_maps(int n_views, const orc_view_input *cams, const int *w,
    const int *h, float **depth, float **normals);

#ifdef __cplusplus
}
#endif
#endif
