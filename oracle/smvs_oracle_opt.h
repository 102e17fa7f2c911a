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
} orc_opt_log;

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

#ifdef __cplusplus
}
#endif
#endif
