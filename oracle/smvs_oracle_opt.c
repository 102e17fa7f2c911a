/*
 * smvs_oracle_opt.c -- CPU restatement of the host side of the path:
 * StereoView::set_scale (lib/stereo_view.cc), Surface topology
 * (lib/surface.cc) and DepthOptimizer::optimize / run_newton_iterations /
 * create_subview_surfaces / cut_boundaries (lib/depth_optimizer.cc).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference test covers these
 * functions and the reference cannot be built here); MVE image operations are
 * [MVE-unverified].
 */
#include "smvs_oracle.h"
#include "smvs_oracle_opt.h"

#include <float.h>
#include <math.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define POW2(x) ((x) * (x))

/* ====================================================================== */
/* images                                                                  */
/* ====================================================================== */

/* mve::image::blur_gaussian<float> [MVE-unverified]: separable, kernel half
 * width ceil(2.884 sigma), weights exp(-i^2 / (2 sigma^2)), clamped border,
 * Accum<float>::normalized(). */
static float *
blur_gaussian(const float *in, int w, int h, int c, float sigma)
{
    size_t const n = (size_t)w * h * c;
    float *out = (float *)malloc(sizeof(float) * n);
    if (fabsf(sigma) < 0.1f)
    {
        memcpy(out, in, sizeof(float) * n);
        return out;
    }
    int const ks = (int)ceilf(sigma * 2.884f);
    float *kernel = (float *)malloc(sizeof(float) * (ks + 1));
    for (int i = 0; i < ks + 1; ++i)
        kernel[i] = expf(-((float)i * (float)i) / (2.0f * sigma * sigma));
    float *sep = (float *)malloc(sizeof(float) * n);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int cc = 0; cc < c; ++cc)
            {
                float av = 0.0f, aw = 0.0f;
                for (int i = -ks; i <= ks; ++i)
                {
                    int idx = x + i;
                    idx = idx < 0 ? 0 : (idx > w - 1 ? w - 1 : idx);
                    float const kw = kernel[i < 0 ? -i : i];
                    av += in[((size_t)y * w + idx) * c + cc] * kw;
                    aw += kw;
                }
                sep[((size_t)y * w + x) * c + cc] = av / aw;
            }
    for (int x = 0; x < w; ++x)
        for (int y = 0; y < h; ++y)
            for (int cc = 0; cc < c; ++cc)
            {
                float av = 0.0f, aw = 0.0f;
                for (int i = -ks; i <= ks; ++i)
                {
                    int idx = y + i;
                    idx = idx < 0 ? 0 : (idx > h - 1 ? h - 1 : idx);
                    float const kw = kernel[i < 0 ? -i : i];
                    av += sep[((size_t)idx * w + x) * c + cc] * kw;
                    aw += kw;
                }
                out[((size_t)y * w + x) * c + cc] = av / aw;
            }
    free(sep);
    free(kernel);
    return out;
}

/* DESATURATE_LUMINANCE [MVE-unverified]: 0.21 R + 0.72 G + 0.07 B */
static float *
desaturate(const float *in, int w, int h, int c)
{
    float *out = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (size_t p = 0; p < (size_t)w * h; ++p)
        out[p] = c >= 3 ? in[p * c] * 0.21f + in[p * c + 1] * 0.72f
            + in[p * c + 2] * 0.07f : in[p * c];
    return out;
}

/* stereo_view.cc:97-188 */
void
orc_gradients_and_hessian(const float *input, int w, int h, float *gradient,
    float *hessian)
{
    memset(gradient, 0, sizeof(float) * 2 * (size_t)w * h);
    if (hessian != NULL)
        memset(hessian, 0, sizeof(float) * 3 * (size_t)w * h);
    /* the 6x9 least-squares quadratic-fit matrix: rows xx, yy, xy, x, y, 1;
     * window ordered x-offset outer, y-offset inner (:172-174) */
    double M[6][9];
    int col = 0;
    for (int a = -1; a < 2; ++a)
        for (int b = -1; b < 2; ++b, ++col)
        {
            M[0][col] = a == 0 ? -1.0 / 3.0 : 1.0 / 6.0;
            M[1][col] = b == 0 ? -1.0 / 3.0 : 1.0 / 6.0;
            M[2][col] = a * b == 0 ? 0.0 : (a * b > 0 ? 1.0 / 4.0 : -1.0 / 4.0);
            M[3][col] = a == 0 ? 0.0 : (a > 0 ? 1.0 / 6.0 : -1.0 / 6.0);
            M[4][col] = b == 0 ? 0.0 : (b > 0 ? 1.0 / 6.0 : -1.0 / 6.0);
            M[5][col] = (a == 0 && b == 0) ? 5.0 / 9.0
                : ((a == 0 || b == 0) ? 2.0 / 9.0 : -1.0 / 9.0);
        }
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x)
        {
            double v[9];
            int c = 0;
            for (int a = -1; a < 2; ++a)
                for (int b = -1; b < 2; ++b)
                    v[c++] = input[(size_t)(y + b) * w + (x + a)];
            double r[6];
            for (int k = 0; k < 6; ++k)
            {
                double s = 0.0;
                for (int i = 0; i < 9; ++i)
                    s += M[k][i] * v[i];
                r[k] = s;
            }
            size_t const p = (size_t)y * w + x;
            gradient[2 * p + 0] = (float)r[3];
            gradient[2 * p + 1] = (float)r[4];
            if (hessian == NULL)
                continue;
            hessian[3 * p + 0] = (float)(2.0 * r[0]);
            hessian[3 * p + 1] = (float)r[2];
            hessian[3 * p + 2] = (float)(2.0 * r[1]);
        }
}

typedef struct {
    int w, h, c, view_id;
    float *image;        /* byte_to_float_image, w*h*c */
    float *grad, *hess;  /* current scale */
    float *shading, *shading_grad;
    float K[9], Kinv[9];
    float rot[9], trans[3];
} OView;

/* CameraInfo::fill_calibration / fill_inverse_calibration [MVE-unverified] */
static void
fill_calibration(float flen, int w, int h, float *K, float *Kinv)
{
    float const dim = (float)(w > h ? w : h);
    float const ax = flen * dim, ay = flen * dim;
    float const K_[9] = { ax, 0, (float)w * 0.5f, 0, ay, (float)h * 0.5f, 0, 0, 1 };
    float const Ki[9] = { 1.0f / ax, 0, -(float)w * 0.5f / ax, 0, 1.0f / ay,
        -(float)h * 0.5f / ay, 0, 0, 1 };
    memcpy(K, K_, sizeof(K_));
    memcpy(Kinv, Ki, sizeof(Ki));
}

static void
mat3_mul(const float *A, const float *B, float *C)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
        {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k)
                s += A[3 * r + k] * B[3 * k + c];
            C[3 * r + c] = s;
        }
}

/* CameraInfo::fill_reprojection [MVE-unverified]:
 * M = K_d R_d R_s^T K_s^-1,  t = K_d (t_d - R_d R_s^T t_s), all float */
void
orc_fill_reprojection(const float *Ks_inv, const float *Rs, const float *ts,
    const float *Kd, const float *Rd, const float *td, float *M, float *t)
{
    float RsT[9], Rrel[9], tmp[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            RsT[3 * r + c] = Rs[3 * c + r];
    mat3_mul(Rd, RsT, Rrel);
    mat3_mul(Kd, Rrel, tmp);
    mat3_mul(tmp, Ks_inv, M);
    float v[3];
    for (int r = 0; r < 3; ++r)
    {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += Rrel[3 * r + k] * ts[k];
        v[r] = td[r] - s;
    }
    for (int r = 0; r < 3; ++r)
    {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += Kd[3 * r + k] * v[k];
        t[r] = s;
    }
}

static void
oview_init(OView *v, const orc_view_input *in, int linear)
{
    memset(v, 0, sizeof(*v));
    v->w = in->width; v->h = in->height; v->c = in->channels;
    v->view_id = in->view_id;
    size_t const n = (size_t)v->w * v->h * v->c;
    v->image = (float *)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; ++i)
        v->image[i] = (float)in->bytes[i] / 255.0f;
    fill_calibration(in->flen, v->w, v->h, v->K, v->Kinv);
    memcpy(v->rot, in->rot, sizeof(v->rot));
    memcpy(v->trans, in->trans, sizeof(v->trans));
    if (linear)
    {
        /* StereoView::initialize_linear (stereo_view.cc:64-84): linear == 2 with
         * gamma_correction -- the duplicate of the image through
         * mve::image::gamma_correct_inv_srgb<float> (tests/golden/README.md M9)
         * before it is desaturated; the photometric planes keep the image */
        float *lin = v->image;
        if (linear == 2)
        {
            lin = (float *)malloc(sizeof(float) * n);
            for (size_t i = 0; i < n; ++i)
            {
                float const x = v->image[i];
                lin[i] = x <= 0.04045f ? x / 12.92f
                    : powf((x + 0.055f) / 1.055f, 2.4f);
            }
        }
        v->shading = desaturate(lin, v->w, v->h, v->c);
        if (lin != v->image)
            free(lin);
        v->shading_grad = (float *)malloc(sizeof(float) * 2 * (size_t)v->w * v->h);
        orc_gradients_and_hessian(v->shading, v->w, v->h, v->shading_grad, NULL);
    }
}

/* StereoView::set_scale, stereo_view.cc:24-46 */
static void
oview_set_scale(OView *v, int scale)
{
    double const sigma = 0.12 * pow(2.0, scale) + 0.2;
    float *blur = blur_gaussian(v->image, v->w, v->h, v->c, (float)sigma);
    float *grey = desaturate(blur, v->w, v->h, v->c);
    free(v->grad);
    free(v->hess);
    v->grad = (float *)malloc(sizeof(float) * 2 * (size_t)v->w * v->h);
    v->hess = (float *)malloc(sizeof(float) * 3 * (size_t)v->w * v->h);
    orc_gradients_and_hessian(grey, v->w, v->h, v->grad, v->hess);
    free(blur);
    free(grey);
}

static void
oview_free(OView *v)
{
    free(v->image); free(v->grad); free(v->hess);
    free(v->shading); free(v->shading_grad);
}

/* ====================================================================== */
/* surface topology (lib/surface.cc)                                       */
/* ====================================================================== */

typedef struct {
    orc_surface s;
    float *depth;     /* init depth image, W*H */
    int cap_nodes, cap_patches;
} OSurf;

static int
node_ok(const OSurf *S, int idx, int idy)
{
    if (idx < 0 || idy < 0 || idx > S->s.npx || idy > S->s.npy)
        return 0;
    return S->s.node_valid[idy * (S->s.npx + 1) + idx];
}

static int
patch_ok(const OSurf *S, int idx, int idy)
{
    if (idx < 0 || idy < 0 || idx >= S->s.npx || idy >= S->s.npy)
        return 0;
    return S->s.patch_valid[idy * S->s.npx + idx];
}

static void
surf_alloc_grid(OSurf *S, int npx, int npy)
{
    int const nn = (npx + 1) * (npy + 1), np = npx * npy;
    S->s.npx = npx;
    S->s.npy = npy;
    S->s.nodes = (double *)calloc((size_t)nn * 4, sizeof(double));
    S->s.node_valid = (uint8_t *)calloc((size_t)nn, 1);
    S->s.patch_valid = (uint8_t *)calloc((size_t)np, 1);
    S->s.patch_vis = (uint32_t *)calloc((size_t)np, sizeof(uint32_t));
}

static void
surf_free_grid(OSurf *S)
{
    free(S->s.nodes); free(S->s.node_valid);
    free(S->s.patch_valid); free(S->s.patch_vis);
}

/* surface.cc:630-651 */
static int
surf_fill_holes(OSurf *S)
{
    int filled = 0;
    for (int x = 0; x < S->s.npx; ++x)
        for (int y = 0; y < S->s.npy; ++y)
        {
            if (patch_ok(S, x, y))
                continue;
            if (node_ok(S, x, y) && node_ok(S, x + 1, y) && node_ok(S, x, y + 1)
                && node_ok(S, x + 1, y + 1))
            {
                S->s.patch_valid[y * S->s.npx + x] = 1;
                filled += 1;
            }
        }
    return filled;
}

/* surface.cc:762-869: a node survives iff one of its incident patches does */
static void
surf_remove_nodes_without_patch(OSurf *S)
{
    int const stride = S->s.npx + 1;
    for (int i = 0; i < stride * (S->s.npy + 1); ++i)
    {
        if (!S->s.node_valid[i])
            continue;
        int const idx = i % stride, idy = i / stride;
        if (!patch_ok(S, idx - 1, idy - 1) && !patch_ok(S, idx, idy - 1)
            && !patch_ok(S, idx - 1, idy) && !patch_ok(S, idx, idy))
            S->s.node_valid[i] = 0;
    }
}

/* surface.cc:667-760 */
static void
surf_initialize_node_from_depth(OSurf *S, int idx, int idy)
{
    int const ps = S->s.patchsize, W = S->s.width, H = S->s.height;
    int const stride = S->s.npx + 1;
    int const x = idx * ps + S->s.start_x;
    int const y = idy * ps + S->s.start_y;
    if (S->s.node_valid[idy * stride + idx])
        return;
    int const ws = ps / 2;
    int const cap = 4 * (ws > 0 ? ws * ws : 1) + 4;
    double *all = (double *)malloc(sizeof(double) * cap);
    int nall = 0;
    double avg[4];
    int cnt[4] = { 0, 0, 0, 0 };
    int num_non_zeros = 4;
    for (int q = 0; q < 4; ++q)
    {
        int const i0 = (q & 1) ? 0 : -ws, i1 = (q & 1) ? ws : 0;
        int const j0 = (q & 2) ? 0 : -ws, j1 = (q & 2) ? ws : 0;
        double mn = 0.0;
        for (int i = i0; i < i1; ++i)
            for (int j = j0; j < j1; ++j)
                if (x + i >= 0 && x + i < W && y + j >= 0 && y + j < H
                    && S->depth[(size_t)(y + j) * W + (x + i)] > 0.0)
                {
                    double const d = S->depth[(size_t)(y + j) * W + (x + i)];
                    if (cnt[q] == 0 || d < mn)
                        mn = d;
                    cnt[q] += 1;
                    all[nall++] = d;
                }
        if (cnt[q] == 0)
        {
            avg[q] = 0.0;
            num_non_zeros -= 1;
        }
        else
            avg[q] = mn;
    }
    if (num_non_zeros == 0 || nall < 2)
    {
        free(all);
        return;
    }
    /* std::nth_element(all, n/2): the value a full sort puts at n/2 */
    for (int a = 1; a < nall; ++a)
    {
        double const key = all[a];
        int b = a - 1;
        while (b >= 0 && all[b] > key)
        {
            all[b + 1] = all[b];
            b -= 1;
        }
        all[b + 1] = key;
    }
    double *node = S->s.nodes + 4 * (size_t)(idy * stride + idx);
    node[0] = all[nall / 2];
    node[1] = node[2] = node[3] = 0.0;
    if (num_non_zeros == 4)
    {
        node[1] = ((avg[1] + avg[3]) - (avg[0] + avg[2])) / 2.0;
        node[2] = ((avg[2] + avg[3]) - (avg[0] + avg[1])) / 2.0;
        node[3] = ((avg[3] - avg[2]) - (avg[1] - avg[0]));
    }
    else
    {
        if ((avg[1] == 0 || avg[0] == 0) && avg[3] != 0 && avg[2] != 0)
            node[1] = (avg[3] - avg[2]);
        else if ((avg[2] == 0 || avg[3] == 0) && avg[1] != 0 && avg[0] != 0)
            node[1] = (avg[1] - avg[0]);
        if ((avg[0] == 0 || avg[2] == 0) && avg[3] != 0 && avg[1] != 0)
            node[2] = (avg[3] - avg[1]);
        else if ((avg[1] == 0 || avg[2] == 0) && avg[0] != 0 && avg[2] != 0)
            node[2] = (avg[2] - avg[0]);
    }
    S->s.node_valid[idy * stride + idx] = 1;
    free(all);
}

/* surface.cc:140-152 */
static void
surf_fill_patches_from_depth(OSurf *S)
{
    for (int i = 0; i < S->s.npx + 1; ++i)
        for (int j = 0; j < S->s.npy + 1; ++j)
            surf_initialize_node_from_depth(S, i, j);
    surf_fill_holes(S);
    surf_remove_nodes_without_patch(S);
}

/* surface.cc:90-130 */
static void
surf_depth_from_bundle(OSurf *S, const orc_bundle *b, const OView *v,
    float flen)
{
    int const W = S->s.width, H = S->s.height;
    double const fwidth2 = (double)W / 2.0, fheight2 = (double)H / 2.0;
    double const fnorm = (double)(W > H ? W : H);
    for (int j = 0; j < b->num_features; ++j)
        for (int k = b->ref_offsets[j]; k < b->ref_offsets[j + 1]; ++k)
            if (b->ref_views[k] == v->view_id)
            {
                const float *fp = b->positions + 3 * (size_t)j;
                float proj[3];
                for (int r = 0; r < 3; ++r)
                {
                    float s = 0.0f;
                    for (int c = 0; c < 3; ++c)
                        s += v->rot[3 * r + c] * fp[c];
                    proj[r] = s + v->trans[r];
                }
                float const depth = proj[2];
                proj[0] = proj[0] * flen / proj[2];
                proj[1] = proj[1] * flen / proj[2];
                float const ix = (float)(proj[0] * fnorm + fwidth2);
                float const iy = (float)(proj[1] * fnorm + fheight2);
                int const x = (int)floorf(ix), y = (int)floorf(iy);
                if (x >= 0 && x < W && y >= 0 && y < H)
                    S->depth[(size_t)y * W + x] = depth;
                break;
            }
}

/* surface.cc:19-53 */
static void
surf_create(OSurf *S, const orc_bundle *bundle, const OView *main, float flen,
    int scale, const float *init_depth)
{
    memset(S, 0, sizeof(*S));
    int const W = main->w, H = main->h;
    S->s.width = W; S->s.height = H;
    S->s.scale = scale;
    S->s.patchsize = 1 << scale;
    int const npx = (W - 2) / S->s.patchsize - 1;
    int const npy = (H - 2) / S->s.patchsize - 1;
    surf_alloc_grid(S, npx, npy);
    S->s.start_x = (W - npx * S->s.patchsize) / 2;
    S->s.start_y = (H - npy * S->s.patchsize) / 2;
    S->depth = (float *)calloc((size_t)W * H, sizeof(float));
    if (init_depth == NULL)
        surf_depth_from_bundle(S, bundle, main, flen);
    else
        for (size_t p = 0; p < (size_t)W * H; ++p)
            if (init_depth[p] > 0.0)
                S->depth[p] = init_depth[p];
    surf_fill_patches_from_depth(S);
}

static void
patch_nodes16(const OSurf *S, int patch_id, double *n16, int *ids)
{
    int const stride = S->s.npx + 1;
    int const idx = patch_id % S->s.npx, idy = patch_id / S->s.npx;
    ids[0] = idy * stride + idx;
    ids[1] = ids[0] + 1;
    ids[2] = ids[0] + stride;
    ids[3] = ids[2] + 1;
    for (int n = 0; n < 4; ++n)
        memcpy(n16 + 4 * n, S->s.nodes + 4 * (size_t)ids[n], 4 * sizeof(double));
}

/* surface.cc:983-1107 */
static void
surf_subdivide(OSurf *S)
{
    int const old_npx = S->s.npx, old_npy = S->s.npy;
    int const old_stride = old_npx + 1;
    S->s.scale -= 1;
    S->s.patchsize = 1 << S->s.scale;
    int new_npx = (S->s.width - 2) / S->s.patchsize;
    int new_npy = (S->s.height - 2) / S->s.patchsize;
    int offset_x = new_npx - old_npx * 2;
    int offset_y = new_npy - old_npy * 2;
    if (offset_x >= 2)
    {
        new_npx = old_npx * 2 + 2;
        S->s.start_x = (S->s.width - new_npx * S->s.patchsize) / 2;
        offset_x = 1;
    }
    else
    {
        offset_x = 0;
        new_npx = old_npx * 2;
    }
    if (offset_y >= 2)
    {
        new_npy = old_npy * 2 + 2;
        S->s.start_y = (S->s.height - new_npy * S->s.patchsize) / 2;
        offset_y = 1;
    }
    else
    {
        offset_y = 0;
        new_npy = old_npy * 2;
    }
    int const new_stride = new_npx + 1;
    int const nn = new_stride * (new_npy + 1);
    double *nodes = (double *)calloc((size_t)nn * 4, sizeof(double));
    uint8_t *valid = (uint8_t *)calloc((size_t)nn, 1);

    static const double pos[5][2] = { { 0.5, 0.0 }, { 0.0, 0.5 }, { 0.5, 0.5 },
        { 1.0, 0.5 }, { 0.5, 1.0 } };
    static const int off[5][2] = { { 1, 0 }, { 0, 1 }, { 1, 1 }, { 2, 1 }, { 1, 2 } };
    for (int p = 0; p < old_npx * old_npy; ++p)
    {
        if (!S->s.patch_valid[p])
            continue;
        int const idx = p % old_npx, idy = p / old_npx;
        int const nidx = 2 * idx + offset_x, nidy = 2 * idy + offset_y;
        double n16[16], coeffs[16];
        int ids[4];
        patch_nodes16(S, p, n16, ids);
        orc_bicubic_coeffs(n16, coeffs);
        for (int k = 0; k < 5; ++k)
        {
            int const id = (nidx + off[k][0]) + new_stride * (nidy + off[k][1]);
            double *nd = nodes + 4 * (size_t)id;
            nd[0] = orc_bicubic_eval(coeffs, 0, pos[k][0], pos[k][1]);
            nd[1] = orc_bicubic_eval(coeffs, 1, pos[k][0], pos[k][1]) / 2;
            nd[2] = orc_bicubic_eval(coeffs, 2, pos[k][0], pos[k][1]) / 2;
            nd[3] = orc_bicubic_eval(coeffs, 3, pos[k][0], pos[k][1]) / 4;
            valid[id] = 1;
        }
    }
    for (int i = 0; i < old_stride * (old_npy + 1); ++i)
    {
        if (!S->s.node_valid[i])
            continue;
        int const idx = i % old_stride, idy = i / old_stride;
        int const id = (2 * idx + offset_x) + new_stride * (2 * idy + offset_y);
        double *src = S->s.nodes + 4 * (size_t)i;
        src[1] /= 2;
        src[2] /= 2;
        src[3] /= 4;
        memcpy(nodes + 4 * (size_t)id, src, 4 * sizeof(double));
        valid[id] = 1;
    }
    surf_free_grid(S);
    S->s.npx = new_npx;
    S->s.npy = new_npy;
    S->s.nodes = nodes;
    S->s.node_valid = valid;
    S->s.patch_valid = (uint8_t *)calloc((size_t)new_npx * new_npy, 1);
    S->s.patch_vis = (uint32_t *)calloc((size_t)new_npx * new_npy, sizeof(uint32_t));
    surf_fill_holes(S);
    surf_remove_nodes_without_patch(S);
}

/* surface.cc:887-927 */
static void
surf_remove_isolated_patches(OSurf *S)
{
    for (int x = 0; x < S->s.npx; ++x)
        for (int y = 0; y < S->s.npy; ++y)
        {
            if (!patch_ok(S, x, y))
                continue;
            int valid = 0;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    if ((dx != 0 || dy != 0) && patch_ok(S, x + dx, y + dy))
                        valid += 1;
            if (valid < 3)
                S->s.patch_valid[y * S->s.npx + x] = 0;
        }
    surf_remove_nodes_without_patch(S);
}

/* surface.cc:472-628 */
static int
surf_expand(OSurf *S)
{
    int const stride = S->s.npx + 1;
    int const nn = stride * (S->s.npy + 1);
    double *newf = (double *)calloc((size_t)nn, sizeof(double));
    uint8_t *has_new = (uint8_t *)calloc((size_t)nn, 1);
    static const int noff[8][2] = { { -1, -1 }, { 0, -1 }, { 1, -1 }, { -1, 0 },
        { 1, 0 }, { -1, 1 }, { 0, 1 }, { 1, 1 } };
    for (int iter = 0; iter < 2; ++iter)
    {
        for (int id = 0; id < nn; ++id)
        {
            if (S->s.node_valid[id] && !has_new[id])
                continue;
            int const idx = id % stride, idy = id / stride;
            const double *nb[8];
            for (int k = 0; k < 8; ++k)
                nb[k] = node_ok(S, idx + noff[k][0], idy + noff[k][1])
                    ? S->s.nodes + 4 * (size_t)((idy + noff[k][1]) * stride
                        + idx + noff[k][0]) : NULL;
            double cand[8];
            int ncand = 0;
#define F(k) (nb[k][0])
#define DX(k) (nb[k][1])
#define DY(k) (nb[k][2])
            if (nb[0] && nb[1] && nb[3])
                cand[ncand++] = ((F(3) + DX(3) / 2.0) + (F(1) + DY(1) / 2.0)) / 2.0;
            if (nb[1] && nb[2] && nb[4])
                cand[ncand++] = ((F(4) - DX(4) / 2.0) + (F(1) + DY(1) / 2.0)) / 2.0;
            if (nb[3] && nb[5] && nb[6])
                cand[ncand++] = ((F(3) + DX(3) / 2.0) + (F(6) - DY(6) / 2.0)) / 2.0;
            if (nb[4] && nb[6] && nb[7])
                cand[ncand++] = ((F(4) - DX(4) / 2.0) + (F(6) - DY(6) / 2.0)) / 2.0;
            if (nb[0] && nb[1] && nb[2])
                cand[ncand++] = ((F(0) + DY(0) / 2.0) + (F(1) + DY(1) / 2.0)
                    + (F(2) + DY(2) / 2.0)) / 3.0;
            if (nb[0] && nb[3] && nb[5])
                cand[ncand++] = ((F(0) + DX(0) / 2.0) + (F(3) + DX(3) / 2.0)
                    + (F(5) + DX(5) / 2.0)) / 3.0;
            if (nb[5] && nb[6] && nb[7])
                cand[ncand++] = ((F(5) - DY(5) / 2.0) + (F(6) - DY(6) / 2.0)
                    + (F(7) - DY(7) / 2.0)) / 3.0;
            if (nb[2] && nb[4] && nb[7])
                cand[ncand++] = ((F(2) - DX(2) / 2.0) + (F(4) - DX(4) / 2.0)
                    + (F(7) - DX(7) / 2.0)) / 3.0;
#undef F
#undef DX
#undef DY
            /* check_swap_nodes, surface.cc:472-480 */
            for (int c = 0; c < ncand; ++c)
                if (!has_new[id] || cand[c] * 0.9 > newf[id])
                {
                    newf[id] = cand[c];
                    has_new[id] = 1;
                }
        }
        for (int id = 0; id < nn; ++id)
            if (has_new[id])
            {
                double *nd = S->s.nodes + 4 * (size_t)id;
                nd[0] = newf[id];
                nd[1] = nd[2] = nd[3] = 0.0;
                S->s.node_valid[id] = 1;
            }
    }
    free(newf);
    free(has_new);
    int const filled = surf_fill_holes(S);
    surf_remove_nodes_without_patch(S);
    return filled;
}

/* ====================================================================== */
/* DepthOptimizer                                                          */
/* ====================================================================== */

typedef struct {
    const orc_opt_options *opts;
    OView *main;
    OView *subs;
    int n_subs;
    double *Mi, *ti;
    OSurf surf;
    const float *sgm_depth;   /* filtered, full resolution, or NULL */
    double lighting[16];
    int has_lighting;
    orc_opt_log *log;
    float flen, inv_flen;
} OOpt;

static void
make_views(const OOpt *O, orc_views *V, orc_subview *sv)
{
    for (int j = 0; j < O->n_subs; ++j)
    {
        sv[j].width = O->subs[j].w;
        sv[j].height = O->subs[j].h;
        sv[j].grad = O->subs[j].grad;
        sv[j].hess = O->subs[j].hess;
    }
    V->width = O->main->w;
    V->height = O->main->h;
    V->flen = O->flen;
    V->inv_flen = O->inv_flen;
    V->grad = O->main->grad;
    V->shading = O->main->shading;
    V->shading_grad = O->main->shading_grad;
    V->n_subs = O->n_subs;
    V->subs = sv;
    V->M = O->Mi;
    V->t = O->ti;
}

static int
count_patches(const OSurf *S)
{
    int n = 0;
    for (int p = 0; p < S->s.npx * S->s.npy; ++p)
        n += S->s.patch_valid[p] ? 1 : 0;
    return n;
}

/* depth_optimizer.cc:747-790 */
static double
opt_mse_for_patch(const OOpt *O, int patch_id)
{
    const OSurf *S = &O->surf;
    int const size = S->s.patchsize;
    double n16[16];
    int ids[4];
    patch_nodes16(S, patch_id, n16, ids);
    int const n = size * size;
    double *pix = (double *)malloc(sizeof(double) * 2 * n);
    double *dep = (double *)malloc(sizeof(double) * n);
    double *dd = (double *)malloc(sizeof(double) * 2 * n);
    int const px0 = S->s.start_x + (patch_id % S->s.npx) * size;
    int const py0 = S->s.start_y + (patch_id / S->s.npx) * size;
    orc_patch_values_at_pixels(n16, px0, py0, size, 1, pix, dep, dd, NULL, NULL);
    double error = 0.0, counter = 0.0;
    int const W = O->main->w;
    for (int i = 0; i < n; ++i)
    {
        size_t const mp = (size_t)pix[2 * i + 1] * W + (size_t)pix[2 * i];
        double const gm0 = O->main->grad[2 * mp], gm1 = O->main->grad[2 * mp + 1];
        for (int j = 0; j < O->n_subs; ++j)
        {
            if (!(S->s.patch_vis[patch_id] & (1u << j)))
                continue;
            const OView *sv = &O->subs[j];
            orc_corr C;
            orc_corr_update(&C, O->Mi + 9 * j, O->ti + 3 * j, pix[2 * i] + 0.5,
                pix[2 * i + 1] + 0.5, dep[i], dd[2 * i], dd[2 * i + 1]);
            double proj[2], jac[4];
            orc_corr_fill(&C, proj);
            orc_corr_fill_jacobian(&C, jac);
            proj[0] -= 0.5;
            proj[1] -= 0.5;
            double const g0 = orc_linear_at_f32(sv->grad, sv->w, sv->h, 2,
                (float)proj[0], (float)proj[1], 0);
            double const g1 = orc_linear_at_f32(sv->grad, sv->w, sv->h, 2,
                (float)proj[0], (float)proj[1], 1);
            double const d0 = gm0 - (jac[0] * g0 + jac[1] * g1);
            double const d1 = gm1 - (jac[2] * g0 + jac[3] * g1);
            error += sqrt(d0 * d0 + d1 * d1);
            counter += 1.0;
        }
    }
    free(pix); free(dep); free(dd);
    if (counter == 0.0)
        return 1.0;
    return error / counter;
}

/* depth_optimizer.cc:792-912 */
static double
opt_ncc_for_patch(const OOpt *O, int patch_id, int sub_id)
{
    const OSurf *S = &O->surf;
    const OView *mv = O->main, *sv = &O->subs[sub_id];
    int const size = S->s.patchsize;
    double n16[16];
    int ids[4];
    patch_nodes16(S, patch_id, n16, ids);
    int const px0 = S->s.start_x + (patch_id % S->s.npx) * size;
    int const py0 = S->s.start_y + (patch_id / S->s.npx) * size;
    int const cap = size * size + 4 + 8 * (size + 2) * 4;
    double *pix = (double *)malloc(sizeof(double) * 2 * cap);
    double *dep = (double *)malloc(sizeof(double) * cap);
    int n = orc_patch_values_at_pixels(n16, px0, py0, size, 1, pix, dep, NULL,
        NULL, NULL);
    double const cx[4] = { px0, px0 + size, px0, px0 + size };
    double const cy[4] = { py0, py0, py0 + size, py0 + size };
    double const cdep[4] = { n16[0], n16[4], n16[8], n16[12] };
    double const minx = cx[0], miny = cy[0], maxx = cx[3], maxy = cy[3];
    if (minx > 1 && maxx < mv->w - 2 && miny > 1 && maxy < mv->h - 2)
    {
        static const int sx[4] = { -1, 1, -1, 1 }, sy[4] = { -1, -1, 1, 1 };
        for (int k = 0; k < 4; ++k)
        {
            pix[2 * n] = cx[k] + sx[k];
            pix[2 * n + 1] = cy[k] + sy[k];
            dep[n++] = cdep[k];
        }
    }
    /* the list grows while it is walked (:823-857) */
    for (int i = 0; i < n; ++i)
    {
        double const x = pix[2 * i], y = pix[2 * i + 1], d = dep[i];
#define PUSH(ax, ay) do { if (n < cap) { pix[2 * n] = (ax); \
    pix[2 * n + 1] = (ay); dep[n++] = d; } } while (0)
        if (miny > 2 && y == miny) { PUSH(x, y - 2); PUSH(x, y - 1); }
        if (maxy < mv->h - 3 && y == maxy) { PUSH(x, y + 2); PUSH(x, y + 1); }
        if (minx > 2 && x == minx) { PUSH(x - 2, y); PUSH(x - 1, y); }
        if (maxx < mv->w - 3 && x == maxx) { PUSH(x + 2, y); PUSH(x + 1, y); }
#undef PUSH
    }
    double *v0 = (double *)malloc(sizeof(double) * 3 * n);
    double *v1 = (double *)malloc(sizeof(double) * 3 * n);
    double means0[3] = { 0, 0, 0 }, means1[3] = { 0, 0, 0 }, counter[3] = { 0, 0, 0 };
    double result = 0.0;
    int early = 0;
    for (int i = 0; i < n && !early; ++i)
    {
        orc_corr C;
        orc_corr_update(&C, O->Mi + 9 * sub_id, O->ti + 3 * sub_id,
            pix[2 * i] + 0.5, pix[2 * i + 1] + 0.5, dep[i], 0, 0);
        double proj[2];
        orc_corr_fill(&C, proj);
        proj[0] -= 0.5;
        proj[1] -= 0.5;
        if (proj[0] < 1 || proj[0] > sv->w - 2 || proj[1] < 1
            || proj[1] > sv->h - 2)
        {
            result = -1;
            early = 1;
            break;
        }
        for (int c = 0; c < 3; ++c)
        {
            int const mc = c < mv->c ? c : mv->c - 1;
            int const sc = c < sv->c ? c : sv->c - 1;
            double const cm = mv->image[((size_t)pix[2 * i + 1] * mv->w
                + (size_t)pix[2 * i]) * mv->c + mc];
            double const cs = orc_linear_at_f32(sv->image, sv->w, sv->h, sv->c,
                (float)proj[0], (float)proj[1], sc);
            counter[c] += 1.0;
            means0[c] += (cm - means0[c]) / counter[c];
            means1[c] += (cs - means1[c]) / counter[c];
            v0[i * 3 + c] = cm;
            v1[i * 3 + c] = cs;
        }
    }
    if (!early)
    {
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c)
            {
                v0[i * 3 + c] -= means0[c];
                v1[i * 3 + c] -= means1[c];
            }
        double const norm0 = sqrt(orc_vec_dot(v0, v0, 3 * (size_t)n));
        double const norm1 = sqrt(orc_vec_dot(v1, v1, 3 * (size_t)n));
        if (norm0 + norm1 < 0.001 * n)
            result = 1;
        else
            result = orc_vec_dot(v0, v1, 3 * (size_t)n) / (norm0 * norm1);
    }
    free(pix); free(dep); free(v0); free(v1);
    return result;
}

/* depth_optimizer.cc:433-604 */
static void
opt_create_subview_surfaces(OOpt *O)
{
    OSurf *S = &O->surf;
    int const W = O->main->w, H = O->main->h;
    int const np = S->s.npx * S->s.npy;
    memset(S->s.patch_vis, 0, sizeof(uint32_t) * np);

    float **cache = (float **)malloc(sizeof(float *) * O->n_subs);
    for (int j = 0; j < O->n_subs; ++j)
    {
        size_t const n = (size_t)(O->subs[j].w + 1) * (O->subs[j].h + 1);
        cache[j] = (float *)malloc(sizeof(float) * n);
        for (size_t i = 0; i < n; ++i)
            cache[j][i] = 10000.0f;
    }
    float *depth = (float *)malloc(sizeof(float) * (size_t)W * H);
    orc_depth_map(&S->s, depth);
    size_t cap = 2 * (size_t)W * H, cnt = 0;
    double *px = (double *)malloc(sizeof(double) * 2 * cap);
    double *pd = (double *)malloc(sizeof(double) * cap);
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < H; ++y)
        {
            if (depth[(size_t)y * W + x] != 0)
            {
                px[2 * cnt] = x; px[2 * cnt + 1] = y;
                pd[cnt++] = depth[(size_t)y * W + x];
            }
            if (O->opts->use_sgm && O->sgm_depth[(size_t)y * W + x] != 0)
            {
                px[2 * cnt] = x; px[2 * cnt + 1] = y;
                pd[cnt++] = O->sgm_depth[(size_t)y * W + x];
            }
        }
    /* first pass: minimal depth per neighbour pixel (:471-500) */
    for (int j = 0; j < O->n_subs; ++j)
    {
        double const sw = O->subs[j].w, sh = O->subs[j].h;
        int const cw = O->subs[j].w + 1;
        for (size_t i = 0; i < cnt; ++i)
        {
            orc_corr C;
            orc_corr_update(&C, O->Mi + 9 * j, O->ti + 3 * j, px[2 * i] + 0.5,
                px[2 * i + 1] + 0.5, pd[i], 0, 0);
            double proj[2];
            orc_corr_fill(&C, proj);
            proj[0] -= 0.5;
            proj[1] -= 0.5;
            double const cutoffset = 3.0;
            if (proj[0] < cutoffset || proj[0] >= sw - cutoffset
                || proj[1] < cutoffset || proj[1] >= sh - cutoffset)
                continue;
            int const cx = (int)proj[0], cy = (int)proj[1];
            for (int x = -1; x < 2; ++x)
                for (int y = -1; y < 2; ++y)
                    if (C.d < cache[j][(size_t)(cy + y) * cw + (cx + x)])
                        cache[j][(size_t)(cy + y) * cw + (cx + x)] = (float)C.d;
        }
    }
    free(px); free(pd); free(depth);

    /* second pass (:502-585) */
    int const size = S->s.patchsize;
    int const n = size * size;
    /* (patches are independent here: each writes its own patch_vis word) */
#if defined(_OPENMP)
#pragma omp parallel num_threads(orc_get_threads())
#endif
    {
    double *pix = (double *)malloc(sizeof(double) * 2 * n);
    double *dep = (double *)malloc(sizeof(double) * n);
    double *dd = (double *)malloc(sizeof(double) * 2 * n);
#if defined(_OPENMP)
#pragma omp for schedule(dynamic, 32)
#endif
    for (int p = 0; p < np; ++p)
    {
        if (!S->s.patch_valid[p])
            continue;
        double n16[16];
        int ids[4];
        patch_nodes16(S, p, n16, ids);
        int const px0 = S->s.start_x + (p % S->s.npx) * size;
        int const py0 = S->s.start_y + (p / S->s.npx) * size;
        orc_patch_values_at_pixels(n16, px0, py0, size, 1, pix, dep, dd, NULL, NULL);
        for (int j = 0; j < O->n_subs; ++j)
        {
            double const sw = O->subs[j].w, sh = O->subs[j].h;
            int const cw = O->subs[j].w + 1;
            int success = 1;
            for (int i = 0; i < n && success; i++)
            {
                orc_corr C;
                orc_corr_update(&C, O->Mi + 9 * j, O->ti + 3 * j,
                    pix[2 * i] + 0.5, pix[2 * i + 1] + 0.5, dep[i], 0, 0);
                double proj[2];
                orc_corr_fill(&C, proj);
                proj[0] -= 0.5;
                proj[1] -= 0.5;
                double const cutoffset = 0.03 * (sw > sh ? sw : sh);
                if (proj[0] < cutoffset || proj[0] >= sw - cutoffset
                    || proj[1] < cutoffset || proj[1] >= sh - cutoffset)
                {
                    success = 0;
                    break;
                }
                int const cx = (int)proj[0], cy = (int)proj[1];
                for (int x = -1; x < 2; ++x)
                    for (int y = -1; y < 2; ++y)
                        if (C.d * 0.95 > cache[j][(size_t)(cy + y) * cw + (cx + x)])
                            success = 0;
            }
            if (!success)
                continue;
            double mx = 0.0;
            for (int i = 0; i < n; ++i)
            {
                orc_corr C;
                orc_corr_update(&C, O->Mi + 9 * j, O->ti + 3 * j,
                    pix[2 * i] + 0.5, pix[2 * i + 1] + 0.5, dep[i], dd[2 * i],
                    dd[2 * i + 1]);
                double jac[4];
                orc_corr_fill_jacobian(&C, jac);
                double S0 = (sqrt(POW2(jac[0] - jac[3]) + POW2(jac[1] + jac[2]))
                    + sqrt(POW2(jac[0] + jac[3]) + POW2(jac[1] - jac[2]))) / 2.0;
                double S1 = fabs(S0 - sqrt(POW2(jac[0] - jac[3])
                    + POW2(jac[1] + jac[2])));
                double const sigma0 = POW2(S0 > S1 ? S0 : S1);
                double const sigma1 = POW2(S0 < S1 ? S0 : S1);
                double const ratio = sigma0 / sigma1;
                mx = mx > ratio ? mx : ratio;   /* std::max(max, ratio) */
            }
            if (mx > 8.0)
                continue;
            if (!O->opts->use_sgm && opt_ncc_for_patch(O, p, j) < 0)
                continue;
            S->s.patch_vis[p] |= (1u << j);
        }
    }
    free(pix); free(dep); free(dd);
    }
    int invalid = 0;
    for (int p = 0; p < np; ++p)
        if (S->s.patch_valid[p] && S->s.patch_vis[p] == 0)
        {
            S->s.patch_valid[p] = 0;
            invalid += 1;
        }
    if (invalid > 0)
        surf_remove_nodes_without_patch(S);
    for (int j = 0; j < O->n_subs; ++j)
        free(cache[j]);
    free(cache);
}

/* depth_optimizer.cc:360-431 */
static int
opt_cut_boundaries(OOpt *O)
{
    OSurf *S = &O->surf;
    int deleted = 0;
    int const np = S->s.npx * S->s.npy;
    int const size = S->s.patchsize;
    const float *invproj = O->main->Kinv;
    for (int p = 0; p < np; ++p)
    {
        if (!S->s.patch_valid[p])
            continue;
        double n16[16];
        int ids[4];
        patch_nodes16(S, p, n16, ids);
        double const depths[4] = { n16[0], n16[4], n16[8], n16[12] };
        /* std::multimap order: first minimum, last maximum */
        int imin = 0, imax = 0;
        for (int i = 1; i < 4; ++i)
        {
            if (depths[i] < depths[imin])
                imin = i;
            if (depths[i] >= depths[imax])
                imax = i;
        }
        double dd_factor = 5.0;
        if (imin + imax == 3)
            dd_factor *= 1.41421356237309504880; /* MATH_SQRT2 */
        float const px = (float)(S->s.start_x + (p % S->s.npx) * size) + 0.5f;
        float const py = (float)(S->s.start_y + (p / S->s.npx) * size) + 0.5f;
        float v[3];
        for (int r = 0; r < 3; ++r)
        {
            float s = 0.0f;
            s += invproj[3 * r + 0] * px;
            s += invproj[3 * r + 1] * py;
            s += invproj[3 * r + 2] * 1.0f;
            v[r] = s;
        }
        float const vnorm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        double const threshold = dd_factor * depths[imin] * invproj[0] * size
            / vnorm;
        double const dist = depths[imax] - depths[imin];
        if (dist > threshold)
        {
            S->s.patch_valid[p] = 0;
            deleted += 1;
        }
    }
    int const stride = S->s.npx + 1;
    /* mse_for_patch reads nothing this loop changes (node validity only
     * changes in remove_nodes_without_patch below): evaluated up front, in
     * parallel, for the patches the loop will visit */
    double *errors = (double *)malloc(sizeof(double) * (size_t)np);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 32) num_threads(orc_get_threads())
#endif
    for (int p = 0; p < np; ++p)
        errors[p] = S->s.patch_valid[p] ? opt_mse_for_patch(O, p) : 0.0;
    for (int p = 0; p < np; ++p)
    {
        if (!S->s.patch_valid[p])
            continue;
        int const idx = p % S->s.npx, idy = p / S->s.npx;
        int const ids[4] = { idy * stride + idx, idy * stride + idx + 1,
            (idy + 1) * stride + idx, (idy + 1) * stride + idx + 1 };
        double const error = errors[p];
        for (int node = 0; node < 4; ++node)
        {
            int const nx = ids[node] % stride, ny = ids[node] / stride;
            int num_invalid = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    if ((dx != 0 || dy != 0) && !node_ok(S, nx + dx, ny + dy))
                        num_invalid += 1;
            if (num_invalid > 1 && error > 0.05)
            {
                S->s.patch_valid[p] = 0;
                deleted += 1;
                break;
            }
        }
    }
    free(errors);
    surf_remove_nodes_without_patch(S);
    return deleted;
}

/* ---------------------------------------------------------------------- */
/* Test access to the topology tests above on caller-provided state         */
/* (the device versions, smvs_topology_*, are checked against these).       */
/* ---------------------------------------------------------------------- */
static void
topo_setup(OOpt *O, OView *mainv, OView *subv, orc_opt_options *opts,
    orc_surface *s, const orc_topo_view *main_view, const orc_topo_view *subs,
    int n_subs, const double *Mi, const double *ti, const float *sgm_depth,
    int use_sgm)
{
    memset(O, 0, sizeof(*O));
    memset(opts, 0, sizeof(*opts));
    opts->use_sgm = use_sgm;
    O->opts = opts;
    memset(mainv, 0, sizeof(*mainv));
    mainv->w = main_view->w; mainv->h = main_view->h; mainv->c = main_view->c;
    mainv->image = (float *)main_view->image;
    mainv->grad = (float *)main_view->grad;
    fill_calibration(main_view->flen, mainv->w, mainv->h, mainv->K, mainv->Kinv);
    for (int j = 0; j < n_subs; ++j)
    {
        memset(&subv[j], 0, sizeof(OView));
        subv[j].w = subs[j].w; subv[j].h = subs[j].h; subv[j].c = subs[j].c;
        subv[j].image = (float *)subs[j].image;
        subv[j].grad = (float *)subs[j].grad;
    }
    O->main = mainv;
    O->subs = subv;
    O->n_subs = n_subs;
    O->Mi = (double *)Mi;
    O->ti = (double *)ti;
    O->surf.s = *s;
    O->sgm_depth = sgm_depth;
}

void
orc_topology_subviews(orc_surface *s, const orc_topo_view *main_view,
    const orc_topo_view *subs, int n_subs, const double *Mi, const double *ti,
    const float *sgm_depth, int use_sgm)
{
    OOpt O; OView mainv; OView subv[32]; orc_opt_options opts;
    topo_setup(&O, &mainv, subv, &opts, s, main_view, subs, n_subs, Mi, ti,
        sgm_depth, use_sgm);
    opt_create_subview_surfaces(&O);
}

void
orc_topology_patch_mse(orc_surface *s, const orc_topo_view *main_view,
    const orc_topo_view *subs, int n_subs, const double *Mi, const double *ti,
    double *mse_out)
{
    OOpt O; OView mainv; OView subv[32]; orc_opt_options opts;
    topo_setup(&O, &mainv, subv, &opts, s, main_view, subs, n_subs, Mi, ti,
        NULL, 0);
    int const np = s->npx * s->npy;
    for (int p = 0; p < np; ++p)
        mse_out[p] = s->patch_valid[p] ? opt_mse_for_patch(&O, p) : -1.0;
}

int
orc_topology_cut_boundaries(orc_surface *s, const orc_topo_view *main_view,
    const orc_topo_view *subs, int n_subs, const double *Mi, const double *ti)
{
    OOpt O; OView mainv; OView subv[32]; orc_opt_options opts;
    topo_setup(&O, &mainv, subv, &opts, s, main_view, subs, n_subs, Mi, ti,
        NULL, 0);
    /* the loops of depth_optimizer.cc:186-190, 323-337 */
    int total = 0, deleted = 11;
    while (deleted > 10)
    {
        deleted = opt_cut_boundaries(&O);
        total += deleted;
    }
    return total;
}

static double
wall_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void
log_push(orc_opt_log *log, int scale, int iter, int steps, int patches,
    int cg_iterations, double loop_seconds, long long active_patch_steps)
{
    if (log == NULL || log->count >= ORC_OPT_LOG_MAX)
        return;
    int const i = log->count++;
    log->loop_seconds[i] = loop_seconds;
    log->active_patch_steps[i] = active_patch_steps;
    log->scale[i] = scale;
    log->iter[i] = iter;
    log->newton_steps[i] = steps;
    log->valid_patches[i] = patches;
    log->cg_iterations[i] = cg_iterations;
}

/* Debug aid: with ORC_DUMP_DIR set, the surface state at the stages of a
 * Newton batch goes to <dir>/s<scale>_i<iter>_<tag>.bin (ints scale, npx, npy;
 * nodes, node_valid, patch_valid, patch_vis) for tools/compare_dumps.py. */
static void
opt_dump(const OOpt *O, int iter, const char *tag)
{
    const char *dir = getenv("ORC_DUMP_DIR");
    if (dir == NULL)
        return;
    const orc_surface *s = &O->surf.s;
    char path[512];
    snprintf(path, sizeof(path), "%s/s%d_i%d_%s.bin", dir, s->scale, iter, tag);
    FILE *f = fopen(path, "wb");
    if (f == NULL)
        return;
    int const hdr[3] = { s->scale, s->npx, s->npy };
    size_t const nn = (size_t)(s->npx + 1) * (s->npy + 1);
    size_t const np = (size_t)s->npx * s->npy;
    fwrite(hdr, sizeof(int), 3, f);
    fwrite(s->nodes, sizeof(double), 4 * nn, f);
    fwrite(s->node_valid, 1, nn, f);
    fwrite(s->patch_valid, 1, np, f);
    fwrite(s->patch_vis, sizeof(uint32_t), np, f);
    fclose(f);
}

/* Measurement control (bench.py's cpu_baseline, not part of the restated
 * algorithm): with n > 0 the Newton loop of the FIRST batch of every scale runs
 * with n OpenMP threads whatever orc_set_threads says for the rest -- one
 * optimize() then times the reference's one-thread-per-view loop on a bounded
 * sample (a batch of every scale) while the other batches and everything
 * between them use all cores.  The results do not depend on the thread count. */
static int g_first_batch_loop_threads = 0;
void orc_set_first_batch_loop_threads(int n) { g_first_batch_loop_threads = n < 0 ? 0 : n; }

/* depth_optimizer.cc:164-358 */
static void
opt_run_newton_iterations(OOpt *O, int num_iters)
{
    OSurf *S = &O->surf;
    int finished = 0;
    for (int iter = 0; iter < num_iters; ++iter)
    {
        int const num_valid_patches = count_patches(S);
        if (iter == 0)
        {
            opt_create_subview_surfaces(O);
            int deleted = 0x7fffffff;
            while (deleted > 10)
                deleted = opt_cut_boundaries(O);
        }
        int const stride = S->s.npx + 1;
        int const nn = stride * (S->s.npy + 1);
        uint8_t *active = (uint8_t *)calloc((size_t)nn, 1);
        size_t num_initial = 0;
        for (int i = 0; i < nn; ++i)
            if (S->s.node_valid[i])
            {
                active[i] = 1;
                num_initial += 1;
            }
        size_t num_active = num_initial;
        unsigned newton_step = 0;
        int cg_total = 0;
        double *H9 = (double *)malloc(sizeof(double) * 144 * (size_t)nn);
        uint8_t *present = (uint8_t *)malloc(9 * (size_t)nn);
        double *g = (double *)malloc(sizeof(double) * 4 * (size_t)nn);
        double *P = (double *)malloc(sizeof(double) * 16 * (size_t)nn);
        double *x = (double *)malloc(sizeof(double) * 4 * (size_t)nn);
        orc_subview *sv = (orc_subview *)malloc(sizeof(orc_subview) * O->n_subs);
        orc_views V;
        make_views(O, &V, sv);
        orc_gn_options gopts = { O->opts->regularization,
            O->opts->light_surf_regularization };
        int const threads_outside = orc_get_threads();
        if (g_first_batch_loop_threads > 0 && iter == 0)
            orc_set_threads(g_first_batch_loop_threads);
        double const t_loop = wall_seconds();
        long long patch_steps = 0;
        for (; newton_step < 200 && num_active > num_initial / 20;)
        {
            newton_step += 1;
            patch_steps += orc_gn_construct(&V, &S->s, &gopts,
                O->has_lighting ? O->lighting : NULL, active, H9, present, g, P);
            double const gnorm = sqrt(orc_vec_dot(g, g, 4 * (size_t)nn));
            for (int i = 0; i < 4 * nn; ++i)
                g[i] = -g[i];
            int its = 0;
            orc_cg_solve(nn, stride, H9, present, P, g, x, 200, gnorm * 0.01,
                1e-3, &its);
            cg_total += its;
            if (isnan(x[0]))
                break;
            double mean = 0.0;
            int const r = orc_update_and_reactivate(&V, &S->s, x, active,
                O->opts->full_optimization, &mean);
            if (O->opts->full_optimization)
            {
                if (mean < 0.01)
                    break;
                else
                    continue;
            }
            num_active = (size_t)r;
        }
        double const loop_seconds = wall_seconds() - t_loop;
        orc_set_threads(threads_outside);
        free(H9); free(present); free(g); free(P); free(x); free(sv);
        free(active);
        log_push(O->log, S->s.scale, iter, (int)newton_step, num_valid_patches,
            cg_total, loop_seconds, patch_steps);
        opt_dump(O, iter, "newton");
        if (finished)
            break;
        int deleted = 0x7fffffff;
        while (deleted > 10)
            deleted = opt_cut_boundaries(O);
        opt_dump(O, iter, "cut");
        if (!O->opts->use_sgm)
        {
            surf_expand(S);
            opt_create_subview_surfaces(O);
            deleted = 0x7fffffff;
            while (deleted > 10)
                deleted = opt_cut_boundaries(O);
        }
        surf_remove_isolated_patches(S);
        int const num_valid_new = count_patches(S);
        int const mn = num_valid_new < num_valid_patches ? num_valid_new
            : num_valid_patches;
        int const mx = num_valid_new > num_valid_patches ? num_valid_new
            : num_valid_patches;
        double const change = 1.0 - (double)mn / (double)mx;
        if (iter > 0 && (num_valid_new <= num_valid_patches
            || change < 0.05 * S->s.scale))
            finished = 1;
    }
}

/* depth_optimizer.cc:53-162 */
int
orc_optimize(const orc_view_input *main_in, const orc_view_input *subs_in,
    int n_subs, const orc_bundle *bundle, const float *sgm_depth,
    const orc_opt_options *opts, float *depth_out, float *normals_out,
    orc_opt_log *log)
{
    if (n_subs < 1 || n_subs > 32)
        return -1;
    if (log != NULL)
        log->count = 0;
    OOpt O;
    memset(&O, 0, sizeof(O));
    O.opts = opts;
    O.n_subs = n_subs;
    O.log = log;
    OView mainv;
    oview_init(&mainv, main_in, opts->use_shading ? (opts->gamma_correction ? 2 : 1) : 0);
    O.main = &mainv;
    O.subs = (OView *)malloc(sizeof(OView) * n_subs);
    for (int j = 0; j < n_subs; ++j)
        oview_init(&O.subs[j], &subs_in[j], 0);
    O.flen = mainv.K[0];
    O.inv_flen = mainv.Kinv[0];
    /* prepare_correspondences (:679-699) */
    O.Mi = (double *)malloc(sizeof(double) * 9 * n_subs);
    O.ti = (double *)malloc(sizeof(double) * 3 * n_subs);
    for (int j = 0; j < n_subs; ++j)
    {
        float M[9], t[3];
        orc_fill_reprojection(mainv.Kinv, mainv.rot, mainv.trans, O.subs[j].K,
            O.subs[j].rot, O.subs[j].trans, M, t);
        for (int k = 0; k < 9; ++k)
            O.Mi[9 * j + k] = M[k];
        for (int k = 0; k < 3; ++k)
            O.ti[3 * j + k] = t[k];
    }

    /* create_initial_surface (:35-51) */
    int const init_scale = (int)fmax(ceil(log2(mainv.w * mainv.h / 1.7e6) / 2)
        + 4, 4.0);
    float *filtered = NULL;
    if (opts->use_sgm)
    {
        if (sgm_depth == NULL)
            return -2;
        filtered = (float *)malloc(sizeof(float) * (size_t)mainv.w * mainv.h);
        orc_bilateral_upsample(sgm_depth, opts->sgm_width, opts->sgm_height,
            mainv.image, mainv.w, mainv.h, mainv.c, 5.0f, 5, filtered);
        surf_create(&O.surf, bundle, &mainv, main_in->flen, init_scale, filtered);
        O.sgm_depth = filtered;
    }
    else
        surf_create(&O.surf, bundle, &mainv, main_in->flen, init_scale + 1, NULL);

    oview_set_scale(&mainv, O.surf.s.scale);
    for (int j = 0; j < n_subs; ++j)
        oview_set_scale(&O.subs[j], O.surf.s.scale);
    opt_run_newton_iterations(&O, opts->num_iterations);

    while (O.surf.s.scale > opts->min_scale && O.surf.s.scale > 0)
    {
        surf_subdivide(&O.surf);
        oview_set_scale(&mainv, O.surf.s.scale);
        for (int j = 0; j < n_subs; ++j)
            oview_set_scale(&O.subs[j], O.surf.s.scale);
        surf_fill_patches_from_depth(&O.surf);
        if (opts->use_shading && O.surf.s.scale < 4)
        {
            size_t const npix = (size_t)mainv.w * mainv.h;
            float *normals = (float *)malloc(sizeof(float) * 3 * npix);
            orc_normal_map(&O.surf.s, O.inv_flen, normals);
            double A[256], b[16];
            orc_light_accumulate(normals, mainv.shading, (int)npix, A, b);
            orc_light_solve(A, b, O.lighting);
            O.has_lighting = 1;
            free(normals);
        }
        opt_run_newton_iterations(&O, opts->num_iterations);
    }

    if (depth_out != NULL)
        orc_depth_map(&O.surf.s, depth_out);
    if (normals_out != NULL)
        orc_normal_map(&O.surf.s, O.inv_flen, normals_out);
    if (log != NULL)
    {
        log->final_scale = O.surf.s.scale;
        log->final_patches = count_patches(&O.surf);
        memcpy(log->lighting, O.lighting, sizeof(O.lighting));
        log->has_lighting = O.has_lighting;
    }

    surf_free_grid(&O.surf);
    free(O.surf.depth);
    free(filtered);
    free(O.Mi); free(O.ti);
    for (int j = 0; j < n_subs; ++j)
        oview_free(&O.subs[j]);
    free(O.subs);
    oview_free(&mainv);
    return 0;
}

/* mve::image::rescale_half_size<uint8_t> [MVE-unverified] (sgm_stereo.cc:31-39):
 * output ((w+1)/2, (h+1)/2), mean of the 2x2 block with 0.25 weights through
 * the u8 interpolate (+0.5f, truncate); odd edges replicate. */
void
orc_rescale_half_size_u8(const uint8_t *in, int w, int h, uint8_t *out)
{
    int const ow = (w + 1) >> 1, oh = (h + 1) >> 1;
    for (int y = 0; y < oh; ++y)
    {
        int const y0 = 2 * y, y1 = (2 * y + 1 < h) ? 2 * y + 1 : h - 1;
        for (int x = 0; x < ow; ++x)
        {
            int const x0 = 2 * x, x1 = (2 * x + 1 < w) ? 2 * x + 1 : w - 1;
            float const v = (float)in[(size_t)y0 * w + x0] * 0.25f
                + (float)in[(size_t)y0 * w + x1] * 0.25f
                + (float)in[(size_t)y1 * w + x0] * 0.25f
                + (float)in[(size_t)y1 * w + x1] * 0.25f;
            out[(size_t)y * ow + x] = (uint8_t)(v + 0.5f);
        }
    }
}

/* StereoView::StereoView + set_scale (stereo_view.cc:16-46) for one u8 image:
 * the gradient (2 ch) and Hessian (3 ch) planes of `scale`. */
void
orc_scale_planes(const uint8_t *bytes, int w, int h, int c, int scale,
    float *grad2, float *hess3)
{
    orc_view_input in;
    memset(&in, 0, sizeof(in));
    in.width = w; in.height = h; in.channels = c; in.bytes = bytes;
    in.flen = 1.0f;
    OView v;
    oview_init(&v, &in, 0);
    oview_set_scale(&v, scale);
    memcpy(grad2, v.grad, sizeof(float) * 2 * (size_t)w * h);
    if (hess3 != NULL)
        memcpy(hess3, v.hess, sizeof(float) * 3 * (size_t)w * h);
    oview_free(&v);
}

/* ---------------------------------------------------------------------- */
/* Surface operations on their own (lib/surface.cc), for the CPU parity   */
/* test of the host mirror's Surface class: create (from the bundle or    */
/* an initial depth map) and then a script of                             */
/*   1 expand, 2 subdivide_patches, 3 fill_patches_from_depth,            */
/*   4 remove_isolated_patches, 5 delete every `delete_every`-th valid    */
/*   patch + remove_nodes_without_patch.                                  */
/* Outputs: info = { scale, npx, npy, start_x, start_y }, and the node /  */
/* validity arrays (caller-sized for the finest scale the script reaches).*/
/* ---------------------------------------------------------------------- */
int
orc_surface_script(const orc_view_input *main_in, const orc_bundle *bundle,
    const float *init_depth, int init_scale, const int *ops, int n_ops,
    int delete_every, int *info, double *nodes_out, uint8_t *node_valid_out,
    uint8_t *patch_valid_out)
{
    OView mainv;
    oview_init(&mainv, main_in, 0);
    OSurf S;
    surf_create(&S, bundle, &mainv, main_in->flen, init_scale, init_depth);
    for (int k = 0; k < n_ops; ++k)
        switch (ops[k])
        {
        case 1: surf_expand(&S); break;
        case 2: surf_subdivide(&S); break;
        case 3: surf_fill_patches_from_depth(&S); break;
        case 4: surf_remove_isolated_patches(&S); break;
        case 5:
        {
            int seen = 0;
            for (int p = 0; p < S.s.npx * S.s.npy; ++p)
                if (S.s.patch_valid[p] && (++seen % delete_every) == 0)
                    S.s.patch_valid[p] = 0;
            surf_remove_nodes_without_patch(&S);
            break;
        }
        default:
            oview_free(&mainv);
            return -1;
        }
    int const nn = (S.s.npx + 1) * (S.s.npy + 1), np = S.s.npx * S.s.npy;
    info[0] = S.s.scale; info[1] = S.s.npx; info[2] = S.s.npy;
    info[3] = S.s.start_x; info[4] = S.s.start_y;
    memcpy(nodes_out, S.s.nodes, sizeof(double) * 4 * (size_t)nn);
    memcpy(node_valid_out, S.s.node_valid, (size_t)nn);
    memcpy(patch_valid_out, S.s.patch_valid, (size_t)np);
    surf_free_grid(&S);
    free(S.depth);
    oview_free(&mainv);
    return 0;
}
