/*
 * smvs_oracle_front.c -- CPU restatement of the callers either side of the
 * depth optimiser: the SGM front end (SGMStereo::reconstruct, the depth range
 * from the bundle, the two-neighbour merge of app/smvsrecon.cc) and the
 * consumer of the depth / normal maps (MeshGenerator::cut_depth_maps).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference holds no test or
 * golden vector for any of this and cannot be built here (needs MVE).
 */
#include "smvs_oracle.h"
#include "smvs_oracle_opt.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* CameraInfo::fill_calibration / fill_inverse_calibration [MVE-unverified]:
 * ppoint = (0.5, 0.5), paspect = 1. */
static void
front_calibration(float flen, int w, int h, float *K, float *Kinv)
{
    float const dim = (float)(w > h ? w : h);
    float const ax = flen * dim, ay = flen * dim;
    if (K != NULL)
    {
        float const K_[9] = { ax, 0, (float)w * 0.5f, 0, ay, (float)h * 0.5f,
            0, 0, 1 };
        memcpy(K, K_, sizeof(K_));
    }
    if (Kinv != NULL)
    {
        float const Ki[9] = { 1.0f / ax, 0, -(float)w * 0.5f / ax, 0,
            1.0f / ay, -(float)h * 0.5f / ay, 0, 0, 1 };
        memcpy(Kinv, Ki, sizeof(Ki));
    }
}

/* StereoView::get_byte_image, stereo_view.cc:86-95: desaturate<uint8_t>
 * (DESATURATE_LUMINANCE) [MVE-unverified: 0.21 R + 0.72 G + 0.07 B, +0.5f,
 * truncate] */
static uint8_t *
front_byte_image(const orc_view_input *in)
{
    size_t const n = (size_t)in->width * in->height;
    uint8_t *out = (uint8_t *)malloc(n);
    int const c = in->channels;
    for (size_t p = 0; p < n; ++p)
        out[p] = c >= 3 ? (uint8_t)((float)in->bytes[p * c] * 0.21f
            + (float)in->bytes[p * c + 1] * 0.72f
            + (float)in->bytes[p * c + 2] * 0.07f + 0.5f) : in->bytes[p * c];
    return out;
}

/* SGMStereo::SGMStereo, sgm_stereo.cc:27-44: `scale` half-sizings */
static uint8_t *
front_sgm_image(const orc_view_input *in, int scale, int *ow, int *oh)
{
    uint8_t *img = front_byte_image(in);
    int w = in->width, h = in->height;
    for (int i = 0; i < scale; ++i)
    {
        int const nw = (w + 1) >> 1, nh = (h + 1) >> 1;
        uint8_t *half = (uint8_t *)malloc((size_t)nw * nh);
        orc_rescale_half_size_u8(img, w, h, half);
        free(img);
        img = half;
        w = nw;
        h = nh;
    }
    *ow = w;
    *oh = h;
    return img;
}

static int
cmp_float(const void *a, const void *b)
{
    float const x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* SGMStereo::fill_depth_range_for_view, sgm_stereo.cc:669-720 */
void
orc_sgm_depth_range(const orc_bundle *bundle, const orc_view_input *view,
    float *range)
{
    int const width = view->width, height = view->height;
    float *values = (float *)malloc(sizeof(float)
        * (size_t)(bundle->num_features > 0 ? bundle->num_features : 1));
    size_t count = 0;
    float const flen = view->flen;
    double const fwidth2 = (double)width / 2.0;
    double const fheight2 = (double)height / 2.0;
    double const fnorm = (double)(width > height ? width : height);
    for (int j = 0; j < bundle->num_features; ++j)
        for (int k = bundle->ref_offsets[j]; k < bundle->ref_offsets[j + 1]; ++k)
            if (bundle->ref_views[k] == view->view_id)
            {
                const float *fpos = bundle->positions + 3 * (size_t)j;
                /* rot * fpos + trans: inner product from 0.0f, then + */
                float proj[3];
                for (int r = 0; r < 3; ++r)
                {
                    float s = 0.0f;
                    for (int c = 0; c < 3; ++c)
                        s += view->rot[3 * r + c] * fpos[c];
                    proj[r] = s + view->trans[r];
                }
                float const depth = proj[2];
                proj[0] = proj[0] * flen / proj[2];
                proj[1] = proj[1] * flen / proj[2];
                float const ix = (float)(proj[0] * fnorm + fwidth2);
                float const iy = (float)(proj[1] * fnorm + fheight2);
                int const x = (int)floorf(ix);
                int const y = (int)floorf(iy);
                if (x >= 0 && x < width && y >= 0 && y < height)
                    values[count++] = depth;
                break;
            }
    qsort(values, count, sizeof(float), cmp_float);
    if (count < 2)
    {
        range[0] = 0.3f;
        range[1] = 1.1f;
    }
    else
    {
        range[0] = values[0] * 0.7f;
        range[1] = (float)(values[(count * 99) / 100] * 5.0);
    }
    free(values);
}

/* SGMStereo::run_sgm, sgm_stereo.cc:98-124, on two SGM-scale u8 images */
static void
front_run_sgm(const uint8_t *main_img, int w, int h, const uint8_t *nbr,
    int nw, int nh, const float *M, const float *t, float min_depth,
    float max_depth, int num_steps, uint16_t p1, uint16_t p2, float *depth)
{
    size_t const vol = (size_t)w * h * num_steps;
    float *depths = (float *)malloc(sizeof(float) * num_steps);
    uint16_t *cost = (uint16_t *)malloc(sizeof(uint16_t) * vol);
    uint16_t *sgm = (uint16_t *)malloc(sizeof(uint16_t) * vol);
    orc_sgm_depths(min_depth, max_depth, num_steps, depths);
    orc_sgm_cost_volume(main_img, w, h, nbr, nw, nh, M, t, depths, num_steps,
        cost);
    orc_sgm_aggregate(cost, w, h, num_steps, p1, p2, sgm);
    orc_sgm_depth_from_volume(sgm, main_img, w, h, depths, num_steps, depth,
        NULL);
    free(depths);
    free(cost);
    free(sgm);
}

/* SGMStereo::reconstruct, sgm_stereo.cc:46-96.  d_main receives the checked
 * depth of the main view at SGM resolution (*ow x *oh). */
static float *
front_reconstruct(const orc_view_input *main_in, const orc_view_input *nbr_in,
    const orc_bundle *bundle, int sgm_scale, float min_depth, float max_depth,
    int num_steps, uint16_t p1, uint16_t p2, int *ow, int *oh)
{
    float range[2] = { min_depth, max_depth };
    int w, h, nw, nh;
    uint8_t *mi = front_sgm_image(main_in, sgm_scale, &w, &h);
    uint8_t *ni = front_sgm_image(nbr_in, sgm_scale, &nw, &nh);

    float Km[9], Kmi[9], Kn[9], Kni[9];
    front_calibration(main_in->flen, w, h, Km, Kmi);
    front_calibration(nbr_in->flen, nw, nh, Kn, Kni);
    float M12[9], t12[3], M21[9], t21[3];
    orc_fill_reprojection(Kmi, main_in->rot, main_in->trans, Kn, nbr_in->rot,
        nbr_in->trans, M12, t12);
    orc_fill_reprojection(Kni, nbr_in->rot, nbr_in->trans, Km, main_in->rot,
        main_in->trans, M21, t21);

    if (bundle != NULL && max_depth == 0.0f)
        orc_sgm_depth_range(bundle, main_in, range);
    float *d_main = (float *)malloc(sizeof(float) * (size_t)w * h);
    front_run_sgm(mi, w, h, ni, nw, nh, M12, t12, range[0], range[1],
        num_steps, p1, p2, d_main);
    if (bundle != NULL && max_depth == 0.0f)
        orc_sgm_depth_range(bundle, nbr_in, range);
    float *d_neig = (float *)malloc(sizeof(float) * (size_t)nw * nh);
    front_run_sgm(ni, nw, nh, mi, w, h, M21, t21, range[0], range[1],
        num_steps, p1, p2, d_neig);

    /* :64-91; the reprojection at the depth maps' sizes is M12, t12 */
    orc_sgm_lr_check(d_main, w, h, d_neig, nw, nh, M12, t12);
    free(d_neig);
    free(mi);
    free(ni);
    *ow = w;
    *oh = h;
    return d_main;
}

/* mve::image::depthmap_convert_conventions<float> [MVE-unverified]: the ray
 * length |invproj (x + 0.5, y + 0.5, 1)| is a float norm widened to double
 * (`double len = px.norm()`), and the pixel is scaled in double and rounded
 * once: dm *= (to_mve ? len : 1.0 / len) */
static void
front_convert_conventions(float *dm, int w, int h, const float *invproj,
    int to_mve)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            float const px = (float)x + 0.5f, py = (float)y + 0.5f;
            float v[3];
            for (int r = 0; r < 3; ++r)
                v[r] = invproj[3 * r] * px + invproj[3 * r + 1] * py
                    + invproj[3 * r + 2];
            float const len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            double const len_d = (double)len;
            float *d = &dm[(size_t)y * w + x];
            *d = (float)((double)*d * (to_mve ? len_d : 1.0 / len_d));
        }
}

/* reconstruct_sgm_depth_for_view, app/smvsrecon.cc:346-384, followed by the
 * StereoView::write_depth_to_view / get_sgm_depth round trip
 * (stereo_view.h:108-130): depth_out is what the optimiser reads. n_neighbors
 * is 1 or more (the first two are used, app/smvsrecon.cc:360-365). */
int
orc_sgm_depth_for_view(const orc_view_input *main_in,
    const orc_view_input *neighbors, int n_neighbors, const orc_bundle *bundle,
    int sgm_scale, float min_depth, float max_depth, int num_steps,
    int penalty1, int penalty2, int roundtrip, float *depth_out, int *out_w,
    int *out_h)
{
    if (n_neighbors < 1)
        return -1;
    int w, h;
    float *d1 = front_reconstruct(main_in, &neighbors[0], bundle, sgm_scale,
        min_depth, max_depth, num_steps, (uint16_t)penalty1,
        (uint16_t)penalty2, &w, &h);
    if (n_neighbors > 1)
    {
        int w2, h2;
        float *d2 = front_reconstruct(main_in, &neighbors[1], bundle,
            sgm_scale, min_depth, max_depth, num_steps, (uint16_t)penalty1,
            (uint16_t)penalty2, &w2, &h2);
        for (size_t p = 0; p < (size_t)w * h; ++p)
        {
            if (d2[p] == 0.0f)
                continue;
            if (d1[p] == 0.0f)
            {
                d1[p] = d2[p];
                continue;
            }
            d1[p] = (d1[p] + d2[p]) * 0.5f;
        }
        free(d2);
    }
    if (roundtrip)
    {
        float Kinv[9];
        front_calibration(main_in->flen, w, h, NULL, Kinv);
        front_convert_conventions(d1, w, h, Kinv, 1);
        front_convert_conventions(d1, w, h, Kinv, 0);
    }
    if (depth_out != NULL)
        memcpy(depth_out, d1, sizeof(float) * (size_t)w * h);
    if (out_w != NULL)
        *out_w = w;
    if (out_h != NULL)
        *out_h = h;
    free(d1);
    return 0;
}

/* ---------------------------------------------------------------------- */
/* MeshGenerator::cut_depth_maps, mesh_generator.cc:24-158                  */
/* ---------------------------------------------------------------------- */

/* MeshGenerator::ViewProjection, mesh_generator.cc:300-342 (all float).
 * math::Matrix3f * Matrix3f / * Vec3f: inner products from 0.0f left to right
 * [MVE-unverified]. */
typedef struct {
    float KR[9];
    float t[3];
} view_proj;

static void
view_proj_init(view_proj *vp, const orc_view_input *cam, int w, int h)
{
    float K[9];
    front_calibration(cam->flen, w, h, K, NULL);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
        {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k)
                s += K[3 * r + k] * cam->rot[3 * k + c];
            vp->KR[3 * r + c] = s;
        }
    /* CameraInfo::fill_camera_pos: -R^T t [MVE-unverified] */
    float pos[3];
    for (int r = 0; r < 3; ++r)
    {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += -cam->rot[3 * k + r] * cam->trans[k];
        pos[r] = s;
    }
    for (int r = 0; r < 3; ++r)
    {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += vp->KR[3 * r + k] * pos[k];
        vp->t[r] = s;
    }
}

static float
dot3f(const float *a, const float *b)
{
    float s = 0.0f;
    for (int k = 0; k < 3; ++k)
        s += a[k] * b[k];
    return s;
}

static void
view_proj_get_proj(const view_proj *vp, const float *pos, float *out)
{
    out[0] = dot3f(vp->KR + 0, pos) - vp->t[0];
    out[1] = dot3f(vp->KR + 3, pos) - vp->t[1];
    out[2] = dot3f(vp->KR + 6, pos) - vp->t[2];
}

static float
view_proj_surface_power(const view_proj *vp, const float *pos,
    const float *normal)
{
    const float *KR = vp->KR;
    float const u = dot3f(KR + 0, pos) - vp->t[0];
    float const v = dot3f(KR + 3, pos) - vp->t[1];
    float const w = dot3f(KR + 6, pos) - vp->t[2];
    float const denom = w * w;
    float u_dx[3], v_dx[3];
    for (int k = 0; k < 3; ++k)
    {
        u_dx[k] = (KR[k] * w - KR[6 + k] * u) / denom;
        v_dx[k] = (KR[3 + k] * w - KR[6 + k] * v) / denom;
    }
    /* math::Vector::cross */
    float const cr[3] = { u_dx[1] * v_dx[2] - u_dx[2] * v_dx[1],
        u_dx[2] * v_dx[0] - u_dx[0] * v_dx[2],
        u_dx[0] * v_dx[1] - u_dx[1] * v_dx[0] };
    return -dot3f(normal, cr);
}

/* mve::geom::pixel_3dpos [MVE-unverified]: ray = invproj (x + 0.5, y + 0.5,
 * 1), normalised, times the (ray-length) depth */
static void
pixel_3dpos(int x, int y, float depth, const float *invproj, float *pos)
{
    float const px = (float)x + 0.5f, py = (float)y + 0.5f;
    float v[3];
    for (int r = 0; r < 3; ++r)
    {
        float s = 0.0f;
        s += invproj[3 * r] * px;
        s += invproj[3 * r + 1] * py;
        s += invproj[3 * r + 2] * 1.0f;
        v[r] = s;
    }
    float const len = sqrtf(dot3f(v, v));
    for (int r = 0; r < 3; ++r)
        pos[r] = v[r] / len * depth;
}

/* Matrix4f::mult(Vec3f, 1.0f) with the cam-to-world matrix [R^T | -R^T t]
 * (CameraInfo::fill_cam_to_world [MVE-unverified]) */
static void
cam_to_world(const orc_view_input *cam, const float *p, float *out)
{
    float c2w_t[3];
    for (int r = 0; r < 3; ++r)
    {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += -cam->rot[3 * k + r] * cam->trans[k];
        c2w_t[r] = s;
    }
    for (int r = 0; r < 3; ++r)
    {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += cam->rot[3 * k + r] * p[k];
        out[r] = s + c2w_t[r] * 1.0f;
    }
}

/* generate_mesh's normal preparation (mesh_generator.cc:189-208) followed by
 * cut_depth_maps (:24-158).
 *   depth[i]   : w[i] x h[i], MVE convention (ray length), as stored by
 *                StereoView::write_depth_to_view; overwritten with the cut map
 *   normals[i] : w[i] x h[i] x 3, camera space as DepthOptimizer writes them;
 *                overwritten with the world-space normals
 * Views with depth[i] == NULL are skipped like nullptr depth maps. */
int
orc_cut_depth_maps(int n_views, const orc_view_input *cams, const int *w,
    const int *h, float **depth, float **normals)
{
    if (n_views < 1)
        return -1;
    /* :197-208 normals to world space: rot = cam-to-world rotation = R^T,
     * input (n0, -n1, -n2) */
    for (int i = 0; i < n_views; ++i)
    {
        if (normals[i] == NULL)
            continue;
        size_t const npix = (size_t)w[i] * h[i];
        for (size_t p = 0; p < npix; ++p)
        {
            float const n[3] = { normals[i][3 * p], -normals[i][3 * p + 1],
                -normals[i][3 * p + 2] };
            for (int r = 0; r < 3; ++r)
            {
                float s = 0.0f;
                for (int k = 0; k < 3; ++k)
                    s += cams[i].rot[3 * k + r] * n[k];
                normals[i][3 * p + r] = s;
            }
        }
    }
    if (n_views < 2)
        return 0;

    /* :31-47: cutmaps = ray-length copies, depthmaps -> z-depth */
    float **cut = (float **)calloc(n_views, sizeof(float *));
    float **cut_j = (float **)calloc(n_views, sizeof(float *));
    float (*invproj)[9] = (float (*)[9])malloc(sizeof(float[9]) * n_views);
    view_proj *vps = (view_proj *)malloc(sizeof(view_proj) * n_views);
    for (int i = 0; i < n_views; ++i)
    {
        front_calibration(cams[i].flen, w[i], h[i], NULL, invproj[i]);
        view_proj_init(&vps[i], &cams[i], w[i], h[i]);
        if (depth[i] == NULL)
            continue;
        size_t const bytes = sizeof(float) * (size_t)w[i] * h[i];
        cut[i] = (float *)malloc(bytes);
        cut_j[i] = (float *)malloc(bytes);
        memcpy(cut[i], depth[i], bytes);
        memcpy(cut_j[i], depth[i], bytes);
        front_convert_conventions(depth[i], w[i], h[i], invproj[i], 0);
    }

    for (int i = 0; i < n_views; ++i)
    {
        if (depth[i] == NULL)
            continue;
        const orc_view_input *cam = &cams[i];
        for (int x = 0; x < w[i]; ++x)
            for (int y = 0; y < h[i]; ++y)
            {
                size_t const p = (size_t)y * w[i] + x;
                float const d = cut[i][p];
                if (d == 0.0f)
                    continue;
                float pc[3], pos[3];
                pixel_3dpos(x, y, d, invproj[i], pc);
                cam_to_world(cam, pc, pos);
                float const normal[3] = { normals[i][3 * p],
                    normals[i][3 * p + 1], normals[i][3 * p + 2] };
                float const surface_power = view_proj_surface_power(&vps[i],
                    pos, normal);
                if (surface_power < 0)
                    cut[i][p] = 0.0f;
                float consistency = 0;
                for (int j = 0; j < n_views; ++j)
                {
                    if (j == i || depth[j] == NULL)
                        continue;
                    float proj[3];
                    view_proj_get_proj(&vps[j], pos, proj);
                    if (proj[2] < 0)
                        continue;
                    int const xj = (int)(proj[0] / proj[2]);
                    int const yj = (int)(proj[1] / proj[2]);
                    if (xj < 0 || xj >= w[j] || yj < 0 || yj >= h[j])
                        continue;
                    size_t const pj = (size_t)yj * w[j] + xj;
                    float const dm_j = depth[j][pj];
                    if (dm_j == 0.0f)
                        continue;
                    float const surface_power_j = view_proj_surface_power(
                        &vps[j], pos, normal);
                    float pcj[3], pos_j[3];
                    pixel_3dpos(xj, yj, cut_j[j][pj], invproj[j], pcj);
                    cam_to_world(&cams[j], pcj, pos_j);
                    float const normal_j[3] = { normals[j][3 * pj],
                        normals[j][3 * pj + 1], normals[j][3 * pj + 2] };
                    float const surface_power_j_j = view_proj_surface_power(
                        &vps[j], pos_j, normal_j);
                    if (dm_j * 1.01 < proj[2])
                        continue;
                    if (dm_j * 0.997 > proj[2])
                    {
                        if (surface_power_j_j > 0.5 * surface_power)
                            consistency -= surface_power_j_j;
                        continue;
                    }
                    if (surface_power_j_j > 2.0 * surface_power
                        || surface_power_j > 2.0 * surface_power)
                    {
                        cut[i][p] = 0.0f;
                        break;
                    }
                    consistency += surface_power_j_j;
                }
                if (consistency <= 0)
                    cut[i][p] = 0.0f;
            }
    }
    for (int i = 0; i < n_views; ++i)
    {
        if (depth[i] == NULL)
            continue;
        memcpy(depth[i], cut[i], sizeof(float) * (size_t)w[i] * h[i]);
        free(cut[i]);
        free(cut_j[i]);
    }
    free(cut);
    free(cut_j);
    free(invproj);
    free(vps);
    return 0;
}

/* ---- pieces of the front end on their own (CPU parity tests of the host
 * mirror): the SGM input image (stereo_view.cc:86-95 + sgm_stereo.cc:27-44)
 * and the reprojection between two views at their image sizes
 * (sgm_stereo.cc:56-62, 154-160) ---- */
int
orc_sgm_image(const orc_view_input *in, int halvings, uint8_t *out, int *ow,
    int *oh)
{
    uint8_t *img = front_sgm_image(in, halvings, ow, oh);
    memcpy(out, img, (size_t)*ow * *oh);
    free(img);
    return 0;
}

void
orc_view_reprojection(const orc_view_input *src, const orc_view_input *dst,
    float *M, float *t)
{
    float Ks[9], Ksi[9], Kd[9], Kdi[9];
    front_calibration(src->flen, src->width, src->height, Ks, Ksi);
    front_calibration(dst->flen, dst->width, dst->height, Kd, Kdi);
    orc_fill_reprojection(Ksi, src->rot, src->trans, Kd, dst->rot, dst->trans,
        M, t);
}

/* mve::image::rescale_half_size_gaussian<uint8_t>(img, sigma2 = 0.75f)
 * [MVE-unverified, tests/golden/README.md M29] -- the filter smvsrecon
 * pre-scales its input embedding with (app/smvsrecon.cc:634-647).  Restated as
 * MVE writes it: four clamped row pointers, four clamped column offsets, and
 * sixteen math::Accum<unsigned char>::add calls per channel in row-major order
 * (a float value sum and a float weight sum), normalized() = the value sum
 * divided by the weight sum, rounded with math::round (half away from zero).
 * out: ((w + 1) / 2) * ((h + 1) / 2) * c bytes. */
void
orc_rescale_half_size_gaussian_u8(const uint8_t *in, int iw, int ih, int ic,
    uint8_t *out)
{
    int const ow = (iw + 1) >> 1, oh = (ih + 1) >> 1;
    float const sigma2 = 0.75f;
    float const w1 = expf(-0.5f / (2.0f * sigma2));
    float const w2 = expf(-2.5f / (2.0f * sigma2));
    float const w3 = expf(-4.5f / (2.0f * sigma2));
    size_t outpos = 0;
    size_t const rowstride = (size_t)iw * ic;
    for (int y = 0; y < oh; ++y)
    {
        int const y2 = y << 1;
        const uint8_t *row[4];
        row[0] = in + (size_t)(y2 - 1 > 0 ? y2 - 1 : 0) * rowstride;
        row[1] = in + (size_t)y2 * rowstride;
        row[2] = in + (size_t)(y2 + 1 < ih - 1 ? y2 + 1 : ih - 1) * rowstride;
        row[3] = in + (size_t)(y2 + 2 < ih - 1 ? y2 + 2 : ih - 1) * rowstride;
        for (int x = 0; x < ow; ++x)
        {
            int const x2 = x << 1;
            int xi[4];
            xi[0] = (x2 - 1 > 0 ? x2 - 1 : 0) * ic;
            xi[1] = x2 * ic;
            xi[2] = (x2 + 1 < iw - 1 ? x2 + 1 : iw - 1) * ic;
            xi[3] = (x2 + 2 < iw - 1 ? x2 + 2 : iw - 1) * ic;
            for (int c = 0; c < ic; ++c)
            {
                float v = 0.0f, w = 0.0f;
#define ORC_ACC(r, k, wt) do { v += (float)row[r][xi[k] + c] * (wt); w += (wt); } while (0)
                ORC_ACC(0, 0, w3); ORC_ACC(0, 1, w2); ORC_ACC(0, 2, w2); ORC_ACC(0, 3, w3);
                ORC_ACC(1, 0, w2); ORC_ACC(1, 1, w1); ORC_ACC(1, 2, w1); ORC_ACC(1, 3, w2);
                ORC_ACC(2, 0, w2); ORC_ACC(2, 1, w1); ORC_ACC(2, 2, w1); ORC_ACC(2, 3, w2);
                ORC_ACC(3, 0, w3); ORC_ACC(3, 1, w2); ORC_ACC(3, 2, w2); ORC_ACC(3, 3, w3);
#undef ORC_ACC
                float const q = v / w;
                out[outpos++] = (uint8_t)(q > 0.0f ? floorf(q + 0.5f) : ceilf(q - 0.5f));
            }
        }
    }
}
