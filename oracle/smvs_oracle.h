/*
 * smvs_oracle.h -- CPU restatement of the smvs depth-optimisation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * Every function restates, in plain C99, the arithmetic of the reference
 * function it cites (file:line relative to the flanggut/smvs tree), in the
 * reference's own operation order (the SSE4.1 branches where the reference
 * build selects them, lib/defines.h:21 + lib/Makefile:4).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - ldl_inverse is checked bit-for-bit against the reference's own header
 *     compiled from /root/reference (oracle/_ref, built by oracle/Makefile).
 *   - bicubic, LDL, dense-vector and block-sparse pieces are checked against
 *     the known-answer values in the reference's gtest files
 *     (tests/golden/reference_known_answers.json).
 *   - Correspondence / surface_derivative / spherical_harmonics derivatives
 *     are checked with the reference's own finite-difference tests, ported.
 *   - GaussNewtonStep::construct, ConjugateGradient::solve, the active-set
 *     update, LightOptimizer and SGMStereo have NO test or golden vector in
 *     the reference and the reference cannot be built here (it needs the
 *     un-vendored MVE library): for those functions PARITY IS UNPINNED --
 *     the restatement is a careful reading of the source, nothing more.
 *   - MVE semantics (Image::linear_at, matrix ops order) are recalled, not
 *     read: MVE is not on disk.  [MVE-unverified]
 */
#ifndef SMVS_ORACLE_H
#define SMVS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- bicubic patch (lib/bicubic_patch.cc) ---------------- */

/* nodes[16] = {n00,n10,n01,n11} x {f,dx,dy,dxy}; coeffs[i*4+j] multiplies
 * x^i y^j (bicubic_patch.cc:56-86). */
void orc_bicubic_coeffs(const double *nodes, double *coeffs);
/* kind: 0 f, 1 dx, 2 dy, 3 dxy, 4 dxx, 5 dyy (bicubic_patch.cc:121-187). */
double orc_bicubic_eval(const double *coeffs, int kind, double x, double y);
/* dn[96] = 4 nodes x {f,dx,dy,dxy,dxx,dyy} x 4 params
 * (bicubic_patch.cc:258-316). */
void orc_node_derivatives(double x, double y, double *dn);
/* bicubic_patch.cc:318-339 */
void orc_node_derivatives_for_patchsize(double x, double y, double patchsize,
    double *dn);
/* surface.cc:929-955 */
void orc_node_derivatives_for_pixel(int pixel_id, int patchsize, double *dn);

/* surface_patch.cc:57-120.  Returns number of sampled pixels.  Any output
 * pointer may be NULL.  second = (dxy, dxx, dyy) per pixel. */
int orc_patch_values_at_pixels(const double *nodes, int pixel_x, int pixel_y,
    int size, int subsample, double *pixels /*n*2*/, double *depths,
    double *first /*n*2*/, double *second /*n*3*/, int *pids);

/* ---------------- correspondence (lib/correspondence.cc) -------------- */

typedef struct {
    double p, q, r;
    double t[3];
    double w;
    double w_prime[2];
    double a, b, d, d2;
    double p_prime[2], q_prime[2], r_prime[2];
} orc_corr;

void orc_corr_update(orc_corr *c, const double *M, const double *t,
    double u, double v, double w, double w_dx, double w_dy);
void orc_corr_fill(const orc_corr *c, double *proj2);
void orc_corr_fill_jacobian(const orc_corr *c, double *jac4);
void orc_corr_fill_derivative(const orc_corr *c, const double *dn,
    double *c_dn /*16*2*/);
void orc_corr_fill_jacobian_derivative_grad(const orc_corr *c,
    const double *grad2, const double *dn, double *jac_dn /*16*2*/);

/* ---------------- surface derivative (lib/surface_derivative.cc) ------ */

void orc_fill_normal(double x, double y, double inv_flen, double w,
    double dx, double dy, double *n3);
void orc_normal_derivative(const double *dn, double x, double y, double f,
    double w, double dx, double dy, double *deriv48);
void orc_normal_divergence(double x, double y, double f, double w,
    double dx, double dy, double dxy, double dxx, double dyy, double *div6);
void orc_normal_divergence_deriv(const double *dn, double x, double y,
    double f, double w, double dx, double dy, double dxy, double dxx,
    double dyy, double *deriv96);

/* ---------------- spherical harmonics (lib/spherical_harmonics.h) ----- */

void orc_sh_evaluate_4_band(const double *n3, double *sh16);
void orc_sh_derivative_4_band(const double *n3, double *deriv48);

/* ---------------- dense / block algebra ------------------------------- */

/* ldl_decomposition.h:43-92 */
void orc_ldl_inverse(double *A, int size);
/* sse_vector.cc:19-41 (SSE4.1 branch) */
double orc_vec_dot(const double *a, const double *b, size_t n);

/* sse_vector.cc:139-213 multiply_add (sign > 0) / multiply_sub */
void orc_vec_multiply_add(const double *a, const double *b, double factor,
    int sign, double *out, size_t n);

/* MVE Image<float>::linear_at, interleaved channels [MVE-unverified]. */
float orc_linear_at_f32(const float *img, int w, int h, int c,
    float x, float y, int ch);
uint8_t orc_linear_at_u8(const uint8_t *img, int w, int h, int c,
    float x, float y, int ch);

/* ---------------- surface + views (flat mirrors) ---------------------- */

typedef struct {
    int width, height;          /* pixel size of the main view */
    int scale, patchsize;
    int npx, npy;               /* patches in x / y */
    int start_x, start_y;       /* pixel_start_x / pixel_start_y */
    double *nodes;              /* (npx+1)*(npy+1)*4: f,dx,dy,dxy */
    uint8_t *node_valid;        /* nodes[i] != nullptr */
    uint8_t *patch_valid;       /* patches[i] != nullptr */
    uint32_t *patch_vis;        /* bit j set <=> sub view j in subsurfaces[p]
                                   (list order is ascending sub id,
                                   depth_optimizer.cc:510-584) */
} orc_surface;

typedef struct {
    int width, height;
    const float *grad;          /* W*H*2 interleaved (I_x, I_y) */
    const float *hess;          /* W*H*3 interleaved (I_xx, I_xy, I_yy) */
} orc_subview;

typedef struct {
    int width, height;
    float flen, inv_flen;       /* StereoView::get_flen / get_inverse_flen */
    const float *grad;          /* W*H*2 */
    const float *shading;       /* W*H*1 or NULL */
    const float *shading_grad;  /* W*H*2 or NULL */
    int n_subs;
    const orc_subview *subs;
    const double *M;            /* n_subs*9 row-major */
    const double *t;            /* n_subs*3 */
} orc_views;

typedef struct {
    double regularization;
    double light_surf_regularization;
} orc_gn_options;

/* Restatement selector for gauss_newton_step.cc:252-383 (0: SSE4.1 branch
 * with SSE2 intrinsics, default; 1: the same branch lane by lane in scalar
 * code; 2: the reference's scalar fallback :335-383). */
void orc_set_k2_mode(int mode);
/* OpenMP threads used by the per-patch loops of orc_gn_construct (results do
 * not depend on it). */
void orc_set_threads(int n);
int orc_get_threads(void);

/* gauss_newton_step.cc:145-518 for one patch: g16 += , H256 += (only
 * entries col2 >= col are touched). lighting16 may be NULL. */
void orc_gn_patch(const orc_views *views, const orc_surface *surf,
    const orc_gn_options *opts, const double *lighting16, int patch_id,
    const double *node_derivatives /* ps*ps*96 */, double *g16, double *H256);

/* Assembled system in "stencil" storage: for node n and slot s (0..8,
 * s = (dy+1)*3 + (dx+1)) H9[(n*9+s)*16 ..] is the 4x4 row-major block
 * (row node n, col node n + dy*stride + dx); present9[n*9+s] tells whether
 * the reference's std::map holds that block (gauss_newton_step.cc:99-121).
 * P[n*16..] = inverted diagonal block (block_sparse_matrix.h:300-316),
 * g[4n..] gradient. Returns number of patches evaluated (active patches). */
int orc_gn_construct(const orc_views *views, const orc_surface *surf,
    const orc_gn_options *opts, const double *lighting16,
    const uint8_t *active_nodes, double *H9, uint8_t *present9, double *g,
    double *P);

/* block_sparse_matrix.h:276-298 on the stencil storage (column-scatter
 * order of the reference). */
void orc_block_spmv(int num_nodes, int node_stride, const double *H9,
    const uint8_t *present9, const double *x, double *y);

/* conjugate_gradient.h:72-202 with block-Jacobi preconditioner P.
 * info: 0 convergence, 1 max iterations, 2 invalid input. */
int orc_cg_solve(int num_nodes, int node_stride, const double *H9,
    const uint8_t *present9, const double *P, const double *b, double *x,
    int max_iterations, double error_tolerance, double q_tolerance,
    int *num_iterations);

/* depth_optimizer.cc:271-303 : reproject, update_nodes, reproject, compare.
 * Updates surf->nodes and active_nodes in place.  Returns new number of
 * active nodes (or -1 when full_optimization: then *mean_delta is set). */
int orc_update_and_reactivate(const orc_views *views, orc_surface *surf,
    const double *delta, uint8_t *active_nodes, int full_optimization,
    double *mean_delta);

/* surface.cc:155-183 */
void orc_depth_map(const orc_surface *surf, float *depth /*W*H*/);
void orc_normal_map(const orc_surface *surf, float inv_flen,
    float *normals /*W*H*3*/);

/* light_optimizer.cc:22-49: accumulate A (16x16 row-major), b (16). */
void orc_light_accumulate(const float *normals, const float *image,
    int num_pixels, double *A256, double *b16);
/* pseudo inverse solve A^+ b for the symmetric PSD A (light_optimizer.cc:50-52,
 * math::matrix_pseudo_inverse is MVE [MVE-unverified]). */
void orc_light_solve(const double *A256, const double *b16, double *params16);

/* ---------------- SGM (lib/sgm_stereo.cc) ----------------------------- */

/* sgm_stereo.cc:126-148, for a W x H x C interleaved u8 image. */
void orc_census_filter(const uint8_t *img, int w, int h, int c,
    uint64_t *out);
/* sgm_stereo.cc:192-203 */
void orc_sgm_depths(float min_depth, float max_depth, int num_steps,
    float *depths);
/* sgm_stereo.cc:150-190 with float M[9], t[3] */
void orc_sgm_warp(const uint8_t *neighbor, int nw, int nh, const float *M,
    const float *t, const float *depths, int num_steps, int w, int h,
    uint8_t *warped /* w*h*num_steps */);
/* sgm_stereo.cc:192-244 -> u16 cost volume [p*num_steps + d] */
void orc_sgm_cost_volume(const uint8_t *main_img, int w, int h,
    const uint8_t *neighbor, int nw, int nh, const float *M, const float *t,
    const float *depths, int num_steps, uint16_t *cost);
/* sgm_stereo.cc:429-667 (SSE branch semantics: constant P2) */
void orc_sgm_aggregate(const uint16_t *cost, int w, int h, int num_steps,
    uint16_t p1, uint16_t p2, uint16_t *sgm);
/* 1: evaluate the path recurrence literally as the reference's O(D^2) SSE
 * loop; 0 (default): the equivalent O(D) form. */
void orc_sgm_set_literal(int on);
/* One step of the path recurrence in the reference's two builds (Q18):
 * scalar sgm_stereo.cc:310-346 (penalty2 adapted to |i1 - i2|) and SSE
 * :361-406 (constant penalty2, evaluated literally). */
void orc_sgm_path_step_scalar(const uint16_t *prev, const uint16_t *cost, int D,
    int i1, int i2, uint16_t p1, uint16_t p2, uint16_t *out);
void orc_sgm_path_step_sse(const uint16_t *prev, const uint16_t *cost, int D,
    uint16_t p1, uint16_t p2, uint16_t *out);
/* sgm_stereo.cc:274-306 */
void orc_sgm_depth_from_volume(const uint16_t *sgm, const uint8_t *main_img,
    int w, int h, const float *depths, int num_steps, float *depth,
    int32_t *argmin /* may be NULL */);
/* sgm_stereo.cc:64-91 L/R check, in place on d_main */
void orc_sgm_lr_check(float *d_main, int w, int h, const float *d_neig,
    int nw, int nh, const float *M, const float *t);

/* depth_optimizer.cc:957-1004 */
void orc_bilateral_upsample(const float *dm, int dm_w, int dm_h,
    const float *ci, int w, int h, int channels, float sigma,
    int kernel_size, float *out);

#ifdef __cplusplus
}
#endif
#endif
