/*
 * smvs_oracle_sgm.c -- CPU restatement of SGMStereo (lib/sgm_stereo.cc) and
 * of the joint bilateral upsample (lib/depth_optimizer.cc:957-1004).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference has no test or
 * golden vector for SGM and cannot be built here (needs MVE).
 *
 * The reference build defines __SSE4_1__ (lib/Makefile:4, -march=native), so
 * the SSE branch semantics are restated: constant penalty2
 * (sgm_stereo.cc:366-367), u16 volumes, wrapping u16 adds.
 */
#include "smvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <alloca.h>
#include <string.h>

/* sgm_stereo.cc:126-148 */
void
orc_census_filter(const uint8_t *img, int w, int h, int c, uint64_t *out)
{
    memset(out, 0, sizeof(uint64_t) * (size_t)w * h * c);
    /* (pixels are independent; threads only when orc_set_threads asked) */
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) num_threads(orc_get_threads())
#endif
    for (int x = 4; x < w - 5; ++x)
        for (int y = 3; y < h - 4; ++y)
            for (int d = 0; d < c; ++d)
            {
                uint8_t const threshold = img[((size_t)y * w + x) * c + d];
                if (threshold == 0)
                    continue;
                uint64_t census = 0;
                for (int i = x - 4; i < x + 5; ++i)
                    for (int j = y - 3; j < y + 4; ++j)
                    {
                        census *= 2;
                        if (threshold < img[((size_t)j * w + i) * c + d])
                            census += 1;
                    }
                out[((size_t)y * w + x) * c + d] = census;
            }
}

/* sgm_stereo.cc:195-203 */
void
orc_sgm_depths(float min_depth, float max_depth, int num_steps, float *depths)
{
    float inv_depth = 1.0f / max_depth;
    float const increment = (1.0f / min_depth - inv_depth) / (num_steps - 1);
    for (int i = 0; i < num_steps; ++i)
    {
        depths[i] = 1.0f / inv_depth;
        inv_depth += increment;
    }
}

/* sgm_stereo.cc:150-190.  math::Matrix3f * Vec3f is an inner product from
 * 0.0f left to right; Vec3f * float then + t component-wise. */
void
orc_sgm_warp(const uint8_t *neighbor, int nw, int nh, const float *M,
    const float *t, const float *depths, int num_steps, int w, int h,
    uint8_t *warped)
{
    memset(warped, 0, (size_t)w * h * num_steps);
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) num_threads(orc_get_threads())
#endif
    for (int x = 0; x < w; ++x)
        for (int y = 0; y < h; ++y)
        {
            float const px = 0.5f + x, py = 0.5f + y, pz = 1.f;
            float tp[3];
            for (int r = 0; r < 3; ++r)
            {
                float s = 0.0f;
                s += M[3 * r + 0] * px;
                s += M[3 * r + 1] * py;
                s += M[3 * r + 2] * pz;
                tp[r] = s;
            }
            for (int d = 0; d < num_steps; ++d)
            {
                float projected[3];
                for (int r = 0; r < 3; ++r)
                    projected[r] = tp[r] * depths[d] + t[r];
                if (projected[2] < 0)
                    continue;
                projected[0] /= projected[2];
                projected[1] /= projected[2];
                projected[0] -= 0.5f;
                projected[1] -= 0.5f;
                if (projected[0] < 0 || projected[1] < 0
                    || projected[0] > nw - 1 || projected[1] > nh - 1)
                    continue;
                warped[((size_t)y * w + x) * num_steps + d] =
                    orc_linear_at_u8(neighbor, nw, nh, 1, projected[0],
                        projected[1], 0);
            }
        }
}

/* sgm_stereo.cc:192-244 */
void
orc_sgm_cost_volume(const uint8_t *main_img, int w, int h,
    const uint8_t *neighbor, int nw, int nh, const float *M, const float *t,
    const float *depths, int num_steps, uint16_t *cost)
{
    size_t const npix = (size_t)w * h;
    uint8_t *n_warped = (uint8_t *)malloc(npix * num_steps);
    uint64_t *n_census = (uint64_t *)malloc(sizeof(uint64_t) * npix * num_steps);
    uint64_t *main_census = (uint64_t *)malloc(sizeof(uint64_t) * npix);
    orc_census_filter(main_img, w, h, 1, main_census);
    for (size_t i = 0; i < npix * num_steps; ++i)
        cost[i] = 255;
    orc_sgm_warp(neighbor, nw, nh, M, t, depths, num_steps, w, h, n_warped);
    orc_census_filter(n_warped, w, h, num_steps, n_census);
    for (size_t p = 0; p < npix; ++p)
        for (int i = 0; i < num_steps; ++i)
        {
            if (n_warped[p * num_steps + i] == 0)
                continue;
            uint8_t const count = (uint8_t)__builtin_popcountll(
                main_census[p] ^ n_census[p * num_steps + i]);
            cost[p * num_steps + i] = count;
        }
    free(n_warped);
    free(n_census);
    free(main_census);
}

static int g_literal = 0;
void orc_sgm_set_literal(int on) { g_literal = on; }

/* sgm_stereo.cc:361-406.  With P2 >= P1 the 128 x 128 SSE evaluation equals
 * min(L[d], L[d-1]+P1, L[d+1]+P1, min_k(L[k]+P2)); all in wrapping u16. */
static void
fill_path_cost(const uint16_t *cost, uint16_t *sgm, uint16_t *path,
    size_t base, size_t pbase, int D, uint16_t p1, uint16_t p2)
{
    uint16_t min_prev = 0xFFFF;
    for (int k = 0; k < D; ++k)
        if (path[pbase + k] < min_prev)
            min_prev = path[pbase + k];
    uint16_t *upd = (uint16_t *)alloca(sizeof(uint16_t) * D);
    if (g_literal)
    {
        for (int idx = 0; idx < D; ++idx)
        {
            /* cost_updates = prev + P2 everywhere, then [idx] = prev[idx],
             * [idx +- 1] = prev + P1, min over all (:372-388). */
            uint16_t m = 0xFFFF;
            for (int k = 0; k < D; ++k)
            {
                uint16_t v;
                if (k == idx)
                    v = path[pbase + k];
                else if (k == idx - 1 || k == idx + 1)
                    v = (uint16_t)(path[pbase + k] + p1);
                else
                    v = (uint16_t)(path[pbase + k] + p2);
                if (v < m)
                    m = v;
            }
            upd[idx] = m;
        }
    }
    else
    {
        /* O(D) form, identical while P2 >= P1 and nothing wraps
         * (L <= 255 + P2): tests/test_oracle_sgm.py checks both forms. */
        uint16_t const mp2 = (uint16_t)(min_prev + p2);
        for (int idx = 0; idx < D; ++idx)
        {
            uint16_t m = path[pbase + idx];
            if (idx > 0 && (uint16_t)(path[pbase + idx - 1] + p1) < m)
                m = (uint16_t)(path[pbase + idx - 1] + p1);
            if (idx < D - 1 && (uint16_t)(path[pbase + idx + 1] + p1) < m)
                m = (uint16_t)(path[pbase + idx + 1] + p1);
            if (mp2 < m)
                m = mp2;
            upd[idx] = m;
        }
    }
    for (int idx = 0; idx < D; ++idx)
    {
        uint16_t v = (uint16_t)(cost[base + idx] + upd[idx]);
        v = (uint16_t)(v - min_prev);
        path[base + idx] = v;
        sgm[base + idx] = (uint16_t)(sgm[base + idx] + v);
    }
}

/* One step of the path recurrence as the reference's two builds evaluate it
 * (Q18), on caller-provided rows: prev[D] = path costs at the predecessor
 * pixel, cost[D] = matching costs at the pixel, out[D] = new path costs.
 *
 * Scalar build, sgm_stereo.cc:310-346: penalty2 adapts to the intensity step
 * between the two pixels, max(P1 * 3 / 2, P2 / (|i1 - i2| + 1)). */
void
orc_sgm_path_step_scalar(const uint16_t *prev, const uint16_t *cost, int D,
    int i1, int i2, uint16_t p1, uint16_t p2_opt, uint16_t *out)
{
    uint16_t const diff = (uint16_t)(abs(i1 - i2) + 1);
    uint16_t const penalty1 = p1;
    int const a = penalty1 * 3 / 2, b = p2_opt / diff;
    uint16_t const penalty2 = (uint16_t)(a > b ? a : b);
    uint16_t min_prev_cost = 0xFFFF;
    for (int i = 0; i < D; ++i)
        if (prev[i] < min_prev_cost)
            min_prev_cost = prev[i];
    for (int i = 0; i < D; ++i)
    {
        uint16_t cost_update = prev[i];
        for (int j = 0; j < D; ++j)
        {
            if (i == j)
                continue;
            uint16_t const cand = abs(j - i) == 1
                ? (uint16_t)(prev[j] + penalty1) : (uint16_t)(prev[j] + penalty2);
            if (cand < cost_update)
                cost_update = cand;
        }
        out[i] = (uint16_t)(cost[i] + cost_update - min_prev_cost);
    }
}

/* SSE build, sgm_stereo.cc:361-406, evaluated literally (D reductions over D
 * planes), constant penalty2. */
void
orc_sgm_path_step_sse(const uint16_t *prev, const uint16_t *cost, int D,
    uint16_t p1, uint16_t p2, uint16_t *out)
{
    uint16_t min_prev = 0xFFFF;
    for (int k = 0; k < D; ++k)
        if (prev[k] < min_prev)
            min_prev = prev[k];
    for (int idx = 0; idx < D; ++idx)
    {
        uint16_t m = 0xFFFF;
        for (int k = 0; k < D; ++k)
        {
            uint16_t v;
            if (k == idx)
                v = prev[k];
            else if (k == idx - 1 || k == idx + 1)
                v = (uint16_t)(prev[k] + p1);
            else
                v = (uint16_t)(prev[k] + p2);
            if (v < m)
                m = v;
        }
        uint16_t r = (uint16_t)(cost[idx] + m);
        out[idx] = (uint16_t)(r - min_prev);
    }
}

/* sgm_stereo.cc:408-426 */
static void
copy_cost_and_add(const uint16_t *cost, uint16_t *sgm, uint16_t *local,
    size_t base, int D)
{
    for (int k = 0; k < D; ++k)
    {
        local[base + k] = cost[base + k];
        sgm[base + k] = (uint16_t)(sgm[base + k] + cost[base + k]);
    }
}

/* sgm_stereo.cc:429-667, SSE branch */
void
orc_sgm_aggregate(const uint16_t *cost, int width, int height, int D,
    uint16_t p1, uint16_t p2, uint16_t *sgm)
{
    size_t const total = (size_t)width * height * D;
    size_t const ys = (size_t)width;
    memset(sgm, 0, sizeof(uint16_t) * total);
    uint16_t *lv = (uint16_t *)calloc(total, sizeof(uint16_t));
    uint16_t *d1 = (uint16_t *)calloc(total, sizeof(uint16_t));
    uint16_t *d2 = (uint16_t *)calloc(total, sizeof(uint16_t));
#define BASE(x, y) (((size_t)(y) * ys + (size_t)(x)) * (size_t)D)

    /* left to right (:457-468) */
    memset(lv, 0, sizeof(uint16_t) * total);
    for (int y = 0; y < height; ++y)
        copy_cost_and_add(cost, sgm, lv, BASE(0, y), D);
    for (int x = 1; x < width; ++x)
        for (int y = 0; y < height; ++y)
            fill_path_cost(cost, sgm, lv, BASE(x, y), BASE(x - 1, y), D, p1, p2);

    /* right to left (:483-494) */
    memset(lv, 0, sizeof(uint16_t) * total);
    for (int y = 0; y < height; ++y)
        copy_cost_and_add(cost, sgm, lv, BASE(width - 1, y), D);
    for (int x = width - 2; x >= 0; --x)
        for (int y = 0; y < height; ++y)
            fill_path_cost(cost, sgm, lv, BASE(x, y), BASE(x + 1, y), D, p1, p2);

    /* top to bottom with both diagonals (:511-546) */
    memset(lv, 0, sizeof(uint16_t) * total);
    memset(d1, 0, sizeof(uint16_t) * total);
    memset(d2, 0, sizeof(uint16_t) * total);
    for (int x = 0; x < width; ++x)
    {
        copy_cost_and_add(cost, sgm, lv, BASE(x, 0), D);
        copy_cost_and_add(cost, sgm, d1, BASE(x, 0), D);
        copy_cost_and_add(cost, sgm, d2, BASE(x, 0), D);
    }
    for (int y = 0; y < height; ++y)
        copy_cost_and_add(cost, sgm, d1, BASE(0, y), D);
    for (int y = 0; y < height; ++y)
        copy_cost_and_add(cost, sgm, d2, BASE(width - 1, y), D);
    for (int y = 1; y < height; ++y)
        for (int x = 0; x < width; ++x)
        {
            if (x > 0)
                fill_path_cost(cost, sgm, d1, BASE(x, y), BASE(x - 1, y - 1),
                    D, p1, p2);
            if (x < width - 1)
                fill_path_cost(cost, sgm, d2, BASE(x, y), BASE(x + 1, y - 1),
                    D, p1, p2);
            fill_path_cost(cost, sgm, lv, BASE(x, y), BASE(x, y - 1), D, p1, p2);
        }

    /* bottom to top with both diagonals (:589-624) */
    memset(lv, 0, sizeof(uint16_t) * total);
    memset(d1, 0, sizeof(uint16_t) * total);
    memset(d2, 0, sizeof(uint16_t) * total);
    for (int x = 0; x < width; ++x)
    {
        copy_cost_and_add(cost, sgm, lv, BASE(x, height - 1), D);
        copy_cost_and_add(cost, sgm, d1, BASE(x, height - 1), D);
        copy_cost_and_add(cost, sgm, d2, BASE(x, height - 1), D);
    }
    for (int y = 0; y < height; ++y)
        copy_cost_and_add(cost, sgm, d1, BASE(0, y), D);
    for (int y = 0; y < height; ++y)
        copy_cost_and_add(cost, sgm, d2, BASE(width - 1, y), D);
    for (int y = height - 2; y >= 0; --y)
        for (int x = 0; x < width; ++x)
        {
            if (x > 0)
                fill_path_cost(cost, sgm, d1, BASE(x, y), BASE(x - 1, y + 1),
                    D, p1, p2);
            if (x < width - 1)
                fill_path_cost(cost, sgm, d2, BASE(x, y), BASE(x + 1, y + 1),
                    D, p1, p2);
            fill_path_cost(cost, sgm, lv, BASE(x, y), BASE(x, y + 1), D, p1, p2);
        }
#undef BASE
    free(lv);
    free(d1);
    free(d2);
}

/* sgm_stereo.cc:274-306 */
void
orc_sgm_depth_from_volume(const uint16_t *sgm, const uint8_t *main_img,
    int w, int h, const float *depths, int num_steps, float *depth,
    int32_t *argmin)
{
    size_t p = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x, ++p)
        {
            uint16_t min_error = 0xFFFF;
            int min_index = 0;
            for (int i = 0; i < num_steps; ++i)
            {
                uint16_t const value = sgm[p * num_steps + i];
                if (value < min_error)
                {
                    min_error = value;
                    min_index = i;
                }
            }
            if (argmin != NULL)
                argmin[p] = min_index;
            if (min_index < 2 || main_img[p] < 25)
                depth[p] = 0;
            else
                depth[p] = depths[min_index];
        }
}

/* sgm_stereo.cc:64-91.  Correspondence with M, t widened float -> double
 * (math::Matrix3d(Matrix3f)), integer pixel coordinates, truncating lookup. */
void
orc_sgm_lr_check(float *d_main, int w, int h, const float *d_neig, int nw,
    int nh, const float *Mf, const float *tf)
{
    double M[9], t[3];
    for (int i = 0; i < 9; ++i)
        M[i] = Mf[i];
    for (int i = 0; i < 3; ++i)
        t[i] = tf[i];
    int const cut = (int)(0.03 * (nw > nh ? nw : nh));
    for (int x = 0; x < w; ++x)
        for (int y = 0; y < h; ++y)
        {
            float *dm = &d_main[(size_t)y * w + x];
            if (*dm == 0)
                continue;
            orc_corr c;
            orc_corr_update(&c, M, t, x, y, *dm, 0, 0);
            double coords[2];
            orc_corr_fill(&c, coords);
            if (coords[0] < cut || coords[0] >= nw - cut
                || coords[1] < cut || coords[1] >= nh - cut)
            {
                *dm = 0;
                continue;
            }
            float const cdepth = (float)c.d;
            float const ndepth = d_neig[(size_t)(int)coords[1] * nw
                + (size_t)(int)coords[0]];
            float const ratio = fminf(cdepth, ndepth) / fmaxf(cdepth, ndepth);
            if (ndepth == 0 || ratio < 0.8)
                *dm = 0;
        }
}

/* depth_optimizer.cc:957-1004; math::gaussian(x, s) = exp(-x^2 / (2 s^2)),
 * gaussian_2d(x, y, sx, sy) = exp(-(x^2/(2 sx^2) + y^2/(2 sy^2))), in float
 * [MVE-unverified]; Accum<float>: v += value * weight, w += weight. */
void
orc_bilateral_upsample(const float *dm, int dm_w, int dm_h, const float *ci,
    int w, int h, int channels, float sigma, int kernel_size, float *out)
{
    memset(out, 0, sizeof(float) * (size_t)w * h);
    float const scale_x = (float)dm_w / (float)w;
    float const scale_y = (float)dm_h / (float)h;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) num_threads(orc_get_threads())
#endif
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            float acc_v = 0.0f, acc_w = 0.0f;
            for (int ky = -kernel_size; ky <= kernel_size; ++ky)
                for (int kx = -kernel_size; kx <= kernel_size; ++kx)
                {
                    int ci_x = x + kx, ci_y = y + ky;
                    ci_x = ci_x < 0 ? 0 : (ci_x > w - 1 ? w - 1 : ci_x);
                    ci_y = ci_y < 0 ? 0 : (ci_y > h - 1 ? h - 1 : ci_y);
                    float fx = scale_x * ci_x, fy = scale_y * ci_y;
                    fx = fx < 0.f ? 0.f : (fx > (float)dm_w - 1.f
                        ? (float)dm_w - 1.f : fx);
                    fy = fy < 0.f ? 0.f : (fy > (float)dm_h - 1.f
                        ? (float)dm_h - 1.f : fy);
                    int const dm_x = (int)fx;
                    int const dm_y = (int)fy;
                    float const dv = dm[(size_t)dm_y * dm_w + dm_x];
                    if (dv == 0.0f)
                        continue;
                    float weight = 1.0f;
                    weight *= expf(-((float)kx * (float)kx
                        / (2.0f * sigma * sigma) + (float)ky * (float)ky
                        / (2.0f * sigma * sigma)));
                    for (int c = 0; c < channels; ++c)
                    {
                        float const diff =
                            ci[((size_t)ci_y * w + ci_x) * channels + c]
                            - ci[((size_t)y * w + x) * channels + c];
                        weight *= expf(-(diff * diff) / (2.0f * 0.1f * 0.1f));
                    }
                    acc_v += dv * weight;
                    acc_w += weight;
                }
            if (acc_w > 0)
                out[(size_t)y * w + x] = acc_v / acc_w;
        }
}
