/*
 * smvs_oracle.c -- CPU restatement of the smvs depth-optimisation hot path.
 * TEST INFRASTRUCTURE ONLY (see smvs_oracle.h for the pinning status; parts
 * of this file are "parity unpinned").
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -msse4.1 -mpopcnt (oracle/Makefile).
 * -ffp-contract=off keeps mul/add separate as in the reference's x86 build.
 */
#include "smvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define POW2(x) ((x) * (x))
#define R_FACTOR 1e-4 /* gauss_newton_step.cc:17 */

/* ====================================================================== */
/* bicubic patch                                                           */
/* ====================================================================== */

/*
 * The reference hard-codes the 16x16 Hermite -> monomial matrix
 * (bicubic_patch.cc:20-38).  It is the Kronecker form of the 1-D cubic
 * Hermite conversion L (value0, value1, slope0, slope1 -> 1, t, t^2, t^3);
 * we generate it instead of restating the table:  a[4*j+i] multiplies
 * x^i y^j and the input vector is (f x4 | dx x4 | dy x4 | dxy x4) with the
 * node order n00, n10, n01, n11.
 */
static double g_coeff_matrix[256];
static int g_coeff_ready = 0;

static void
build_coeff_matrix(void)
{
    /* 1-D: p(t) = sum_i t^i * (L[i][0] v0 + L[i][1] v1 + L[i][2] s0 + L[i][3] s1) */
    static const double L[4][4] = {
        { 1, 0, 0, 0 }, { 0, 0, 1, 0 }, { -3, 3, -2, -1 }, { 2, -2, 1, 1 } };
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
        {
            double *row = g_coeff_matrix + 16 * (4 * j + i);
            /* column of the input vector: kind k (f,dx,dy,dxy), node (a,b)
             * with node index = 2*b + a (n00,n10,n01,n11). In x the input
             * is a value (f, dy) or a slope (dx, dxy); same in y. */
            for (int k = 0; k < 4; ++k)
                for (int b = 0; b < 2; ++b)
                    for (int a = 0; a < 2; ++a)
                    {
                        int const xs = (k == 1 || k == 3) ? 2 + a : a;
                        int const ys = (k == 2 || k == 3) ? 2 + b : b;
                        row[4 * k + 2 * b + a] = L[i][xs] * L[j][ys];
                    }
        }
    g_coeff_ready = 1;
}

static const double *
coeff_matrix(void)
{
    if (!g_coeff_ready)
        build_coeff_matrix();
    return g_coeff_matrix;
}

/* bicubic_patch.cc:56-86 */
void
orc_bicubic_coeffs(const double *nodes, double *coeffs)
{
    const double *A = coeff_matrix();
    double x[16];
    for (int n = 0; n < 4; ++n)
        for (int k = 0; k < 4; ++k)
            x[4 * k + n] = nodes[4 * n + k];
    double a[16];
    for (int r = 0; r < 16; ++r)
    {
        /* math::Matrix * Vector: inner product left to right from 0 */
        double s = 0.0;
        for (int c = 0; c < 16; ++c)
            s += A[16 * r + c] * x[c];
        a[r] = s;
    }
    for (int k = 0, j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i, ++k)
            coeffs[i * 4 + j] = a[k];
}

/* bicubic_patch.cc:40-53 */
static void
exponentials(double x, double y, double *ex, double *ey)
{
    ex[0] = 1.0; ex[1] = x; ex[2] = x * x; ex[3] = ex[2] * x;
    ey[0] = 1.0; ey[1] = y; ey[2] = y * y; ey[3] = ey[2] * y;
}

/* bicubic_patch.cc:121-187 */
double
orc_bicubic_eval(const double *c, int kind, double x, double y)
{
    double ex[4], ey[4];
    exponentials(x, y, ex, ey);
    double result = 0.0;
    switch (kind)
    {
    case 0:
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                result += c[i * 4 + j] * ex[i] * ey[j];
        break;
    case 1:
        for (int i = 1; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                result += c[i * 4 + j] * i * ex[i - 1] * ey[j];
        break;
    case 2:
        for (int i = 0; i < 4; ++i)
            for (int j = 1; j < 4; ++j)
                result += c[i * 4 + j] * ex[i] * j * ey[j - 1];
        break;
    case 3:
        for (int i = 1; i < 4; ++i)
            for (int j = 1; j < 4; ++j)
                result += c[i * 4 + j] * i * ex[i - 1] * j * ey[j - 1];
        break;
    case 4:
        for (int i = 2; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                result += c[i * 4 + j] * i * (i - 1) * ex[i - 2] * ey[j];
        break;
    case 5:
        for (int i = 0; i < 4; ++i)
            for (int j = 2; j < 4; ++j)
                result += c[i * 4 + j] * ex[i] * j * (j - 1) * ey[j - 2];
        break;
    }
    return result;
}

/* bicubic_patch.cc:258-300 */
static void
node_deriv(const double *x, const double *y, int node_offset, double *d)
{
    const double *A = coeff_matrix();
    double *d_f = d, *d_dx = d + 4, *d_dy = d + 8, *d_dxy = d + 12;
    double *d_dxx = d + 16, *d_dyy = d + 20;
    for (int i = 0; i < 24; ++i)
        d[i] = 0.0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 4; ++c)
            {
                int const o = 4 * c + node_offset;
                double const m = A[16 * (j * 4 + i) + o];
                d_f[c] += m * x[i] * y[j];
                if (i > 0)
                    d_dx[c] += m * i * x[i - 1] * y[j];
                if (j > 0)
                    d_dy[c] += m * x[i] * j * y[j - 1];
                if (i > 0 && j > 0)
                    d_dxy[c] += m * i * x[i - 1] * j * y[j - 1];
                if (i > 1)
                    d_dxx[c] += m * i * (i - 1) * x[i - 2] * y[j];
                if (j > 1)
                    d_dyy[c] += m * x[i] * j * (j - 1) * y[j - 2];
            }
}

/* bicubic_patch.cc:302-316 */
void
orc_node_derivatives(double x, double y, double *dn)
{
    double ex[4], ey[4];
    exponentials(x, y, ex, ey);
    for (int n = 0; n < 4; ++n)
        node_deriv(ex, ey, n, dn + 24 * n);
}

/* bicubic_patch.cc:318-339 */
void
orc_node_derivatives_for_patchsize(double x, double y, double patchsize,
    double *dn)
{
    orc_node_derivatives(x, y, dn);
    double const patch_to_pixel = 1.0 / patchsize;
    for (int n = 0; n < 4; ++n)
    {
        for (int i = 4; i < 24; ++i)
            dn[24 * n + i] *= patch_to_pixel;
        for (int i = 12; i < 24; ++i)
            dn[24 * n + i] *= patch_to_pixel;
    }
}

/* surface.cc:929-955 */
void
orc_node_derivatives_for_pixel(int pixel_id, int patchsize, double *dn)
{
    int const i = pixel_id % patchsize;
    int const j = pixel_id / patchsize;
    double const x = ((double)i + 0.5) / patchsize;
    double const y = ((double)j + 0.5) / patchsize;
    orc_node_derivatives(x, y, dn);
    double const patch_to_pixel = 1.0 / patchsize;
    for (int n = 0; n < 4; ++n)
    {
        for (int k = 4; k < 24; ++k)
            dn[24 * n + k] *= patch_to_pixel;
        for (int k = 12; k < 24; ++k)
            dn[24 * n + k] *= patch_to_pixel;
    }
}

/* surface_patch.cc:57-120 */
int
orc_patch_values_at_pixels(const double *nodes, int pixel_x, int pixel_y,
    int size, int subsample, double *pixels, double *depths, double *first,
    double *second, int *pids)
{
    double coeffs[16];
    orc_bicubic_coeffs(nodes, coeffs);
    int id = 0;
    for (int pid = 0; pid < size * size;)
    {
        int const i = pid % size;
        int const j = pid / size;
        double x = (double)i;
        double y = (double)j;
        if (pixels != NULL)
        {
            pixels[2 * id + 0] = x + (double)pixel_x;
            pixels[2 * id + 1] = y + (double)pixel_y;
        }
        if (pids != NULL)
            pids[id] = pid;
        x += 0.5;
        y += 0.5;
        x /= size;
        y /= size;
        if (depths != NULL)
            depths[id] = orc_bicubic_eval(coeffs, 0, x, y);
        if (first != NULL)
        {
            first[2 * id + 0] = orc_bicubic_eval(coeffs, 1, x, y);
            first[2 * id + 1] = orc_bicubic_eval(coeffs, 2, x, y);
            first[2 * id + 0] /= (double)size;
            first[2 * id + 1] /= (double)size;
        }
        if (second != NULL)
        {
            second[3 * id + 0] = orc_bicubic_eval(coeffs, 3, x, y);
            second[3 * id + 1] = orc_bicubic_eval(coeffs, 4, x, y);
            second[3 * id + 2] = orc_bicubic_eval(coeffs, 5, x, y);
            double const s2 = (double)(size * size);
            second[3 * id + 0] /= s2;
            second[3 * id + 1] /= s2;
            second[3 * id + 2] /= s2;
        }
        id += 1;
        if (subsample > 1)
        {
            pid += subsample;
            if ((pid / size) % subsample == 1)
                pid += size * (subsample - 1);
        }
        else
            pid += 1;
    }
    return id;
}

/* ====================================================================== */
/* correspondence                                                          */
/* ====================================================================== */

/* correspondence.cc:20-44 */
void
orc_corr_update(orc_corr *c, const double *M, const double *t,
    double u, double v, double w, double w_dx, double w_dy)
{
    c->t[0] = t[0]; c->t[1] = t[1]; c->t[2] = t[2];
    c->w = w;
    c->p_prime[0] = M[0]; c->p_prime[1] = M[1];
    c->q_prime[0] = M[3]; c->q_prime[1] = M[4];
    c->r_prime[0] = M[6]; c->r_prime[1] = M[7];
    c->w_prime[0] = w_dx; c->w_prime[1] = w_dy;
    c->p = M[0] * u + M[1] * v + M[2];
    c->q = M[3] * u + M[4] * v + M[5];
    c->r = M[6] * u + M[7] * v + M[8];
    c->a = w * c->p + t[0];
    c->b = w * c->q + t[1];
    c->d = w * c->r + t[2];
    c->d2 = c->d * c->d;
}

/* correspondence.cc:46-51 */
void
orc_corr_fill(const orc_corr *c, double *corr)
{
    corr[0] = c->a / c->d;
    corr[1] = c->b / c->d;
}

/* correspondence.cc:88-100 */
void
orc_corr_fill_jacobian(const orc_corr *c, double *jac)
{
    jac[0] = (c->w_prime[0] * c->p + c->w * c->p_prime[0]) / c->d;
    jac[2] = (c->w_prime[1] * c->p + c->w * c->p_prime[1]) / c->d;
    jac[0] -= c->a * (c->w_prime[0] * c->r + c->w * c->r_prime[0]) / c->d2;
    jac[2] -= c->a * (c->w_prime[1] * c->r + c->w * c->r_prime[1]) / c->d2;
    jac[1] = (c->w_prime[0] * c->q + c->w * c->q_prime[0]) / c->d;
    jac[3] = (c->w_prime[1] * c->q + c->w * c->q_prime[1]) / c->d;
    jac[1] -= c->b * (c->w_prime[0] * c->r + c->w * c->r_prime[0]) / c->d2;
    jac[3] -= c->b * (c->w_prime[1] * c->r + c->w * c->r_prime[1]) / c->d2;
}

/* correspondence.cc:74-86 */
void
orc_corr_fill_derivative(const orc_corr *c, const double *dn, double *c_dn)
{
    double const du_w = (c->p * c->d - c->r * c->a) / c->d2;
    double const dv_w = (c->q * c->d - c->r * c->b) / c->d2;
    for (int n = 0; n < 4; ++n)
        for (int i = 0; i < 4; ++i)
        {
            c_dn[2 * (n * 4 + i) + 0] = du_w * dn[n * 24 + i];
            c_dn[2 * (n * 4 + i) + 1] = dv_w * dn[n * 24 + i];
        }
}

/* correspondence.cc:102-187 */
void
orc_corr_fill_jacobian_derivative_grad(const orc_corr *c, const double *grad,
    const double *dn, double *jac_dn)
{
    double const d = c->d, d2 = c->d2, w = c->w;
    double const p = c->p, q = c->q, r = c->r;
    const double *t = c->t;
    double const d4 = d2 * d2;
    double const d_prime = 2.0 * d * r;

    double du_a_temp[2], du_a_prime[2], du_b_prime[2], du_c[2];
    double dv_a_temp[2], dv_a_prime[2], dv_b_prime[2], dv_c[2];
    for (int k = 0; k < 2; ++k)
    {
        du_a_temp[k] = w * (c->p_prime[k] * r - p * c->r_prime[k]);
        du_a_prime[k] = 2.0 * du_a_temp[k];
        du_b_prime[k] = (c->p_prime[k] * t[2] - c->r_prime[k] * t[0]);
    }
    double const du_c_prime = (p * t[2] - r * t[0]);
    du_c[0] = c->w_prime[0] * du_c_prime;
    du_c[1] = c->w_prime[1] * du_c_prime;
    for (int k = 0; k < 2; ++k)
    {
        dv_a_temp[k] = w * (c->q_prime[k] * r - q * c->r_prime[k]);
        dv_a_prime[k] = 2.0 * dv_a_temp[k];
        dv_b_prime[k] = (c->q_prime[k] * t[2] - c->r_prime[k] * t[1]);
    }
    double const dv_c_prime = (q * t[2] - r * t[1]);
    dv_c[0] = c->w_prime[0] * dv_c_prime;
    dv_c[1] = c->w_prime[1] * dv_c_prime;

    double du_A[2], dv_A[2];
    for (int k = 0; k < 2; ++k)
    {
        double const du_a_b_c = w * (du_a_temp[k] + du_b_prime[k]) + du_c[k];
        double const dv_a_b_c = w * (dv_a_temp[k] + dv_b_prime[k]) + dv_c[k];
        double const du_ap_bp_d = (du_a_prime[k] + du_b_prime[k]) / d2;
        double const dv_ap_bp_d = (dv_a_prime[k] + dv_b_prime[k]) / d2;
        double const du_abc_dp = du_a_b_c * d_prime / d4;
        double const dv_abc_dp = dv_a_b_c * d_prime / d4;
        du_A[k] = du_ap_bp_d - du_abc_dp;
        dv_A[k] = dv_ap_bp_d - dv_abc_dp;
    }
    double const du_c_prime_d = du_c_prime / d2;
    double const dv_c_prime_d = dv_c_prime / d2;

    for (int n = 0; n < 4; ++n)
        for (int i = 0; i < 4; ++i)
        {
            int const offset = n * 24;
            double du_dn[2], dv_dn[2];
            du_dn[0] = du_A[0] * dn[offset + 0 + i];
            du_dn[1] = du_A[1] * dn[offset + 0 + i];
            dv_dn[0] = dv_A[0] * dn[offset + 0 + i];
            dv_dn[1] = dv_A[1] * dn[offset + 0 + i];
            jac_dn[2 * (n * 4 + i) + 0] =
                (du_dn[0] + du_c_prime_d * dn[offset + 4 + i]) * grad[0]
                + (dv_dn[0] + dv_c_prime_d * dn[offset + 4 + i]) * grad[1];
            jac_dn[2 * (n * 4 + i) + 1] =
                (du_dn[1] + du_c_prime_d * dn[offset + 8 + i]) * grad[0]
                + (dv_dn[1] + dv_c_prime_d * dn[offset + 8 + i]) * grad[1];
        }
}

/* ====================================================================== */
/* surface derivative                                                      */
/* ====================================================================== */

/* surface_derivative.cc:17-28 (Vector::normalize divides by the norm) */
void
orc_fill_normal(double x, double y, double inv_flen, double w,
    double dx, double dy, double *n)
{
    double normal[3];
    normal[0] = dx;
    normal[1] = -dy;
    normal[2] = x * dx + y * dy + w;
    normal[2] *= inv_flen;
    double s = 0.0;
    for (int i = 0; i < 3; ++i)
        s += normal[i] * normal[i];
    double const len = sqrt(s);
    for (int i = 0; i < 3; ++i)
        n[i] = normal[i] / len;
}

/* surface_derivative.cc:31-65 */
void
orc_normal_derivative(const double *d_node, double x, double y, double f,
    double w, double dx, double dy, double *deriv)
{
    double const f_sqr_inv = 1.0 / (f * f);
    double const a = w + x * dx + y * dy;
    double const t = dx * dx + dy * dy + a * a * f_sqr_inv;
    double const n = sqrt(t);
    for (int node = 0; node < 4; ++node)
    {
        const double *dn = d_node + 24 * node;
        for (int i = 0; i < 4; ++i)
        {
            double const w_prime = dn[0 + i];
            double const dx_prime = dn[4 + i];
            double const dy_prime = dn[8 + i];
            double const a_prime = w_prime + x * dx_prime + y * dy_prime;
            double const t_prime_2 = (dx * dx_prime) + (dy * dy_prime)
                + f_sqr_inv * a * a_prime;
            double const n_prime = t_prime_2 / n;
            double const nx_prime = (dx_prime * n - dx * n_prime) / t;
            double const ny_prime = (-dy_prime * n + dy * n_prime) / t;
            double const nz_prime = (a_prime * n - a * n_prime) / (t * f);
            deriv[0 + node * 4 + i] = nx_prime;
            deriv[16 + node * 4 + i] = ny_prime;
            deriv[32 + node * 4 + i] = nz_prime;
        }
    }
}

/* surface_derivative.cc:69-107 */
void
orc_normal_divergence(double x, double y, double f, double w,
    double dx, double dy, double dxy, double dxx, double dyy, double *div)
{
    double const a = (w + x * dx + y * dy);
    double const ax = 2.0 * dx + x * dxx + y * dxy;
    double const ay = 2.0 * dy + y * dyy + x * dxy;
    double t = a / f;
    t = t * t;
    t += POW2(dx) + POW2(dy);
    double const n = sqrt(t);

    double nx = dx * dxx + dy * dxy;
    nx += (1.0 / (f * f)) * (w + x * dx + y * dy)
        * (dx + dx + x * dxx + y * dxy);
    nx /= n;
    double ny = dx * dxy + dy * dyy;
    ny += (1.0 / (f * f)) * (w + x * dx + y * dy)
        * (dy + dy + x * dxy + y * dyy);
    ny /= n;

    double const xx = (dxx * n - dx * nx) / t;
    double const yy = (dyy * n - dy * ny) / t;
    double const xy = (dxy * n - dx * ny) / t;
    double const yx = (dxy * n - dy * nx) / t;
    double const zx = (ax * n - a * nx) / (t * f);
    double const zy = (ay * n - a * ny) / (t * f);
    div[0] = xx; div[1] = -yx; div[2] = zx;
    div[3] = xy; div[4] = -yy; div[5] = zy;
}

/* surface_derivative.cc:109-190 */
void
orc_normal_divergence_deriv(const double *d_node, double x, double y,
    double f, double w, double dx, double dy, double dxy, double dxx,
    double dyy, double *full_deriv)
{
    double const f_sqr_inv = 1.0 / (f * f);
    double const a = w + x * dx + y * dy;
    double const ax = 2.0 * dx + x * dxx + y * dxy;
    double const ay = 2.0 * dy + y * dyy + x * dxy;
    double const a_f2 = a * f_sqr_inv;
    double const t = dx * dx + dy * dy + a * a_f2;
    double const n = sqrt(t);
    double const b = dx * dxx + dy * dxy + a_f2 * (2.0 * dx + x * dxx + y * dxy);
    double const c = dx * dxy + dy * dyy + a_f2 * (2.0 * dy + x * dxy + y * dyy);
    double const nx = b / n;
    double const ny = c / n;

    for (int node = 0; node < 4; ++node)
    {
        const double *dn = d_node + 24 * node;
        for (int i = 0; i < 4; ++i)
        {
            double const w_prime = dn[0 + i];
            double const dx_prime = dn[4 + i];
            double const dy_prime = dn[8 + i];
            double const dxy_prime = dn[12 + i];
            double const dxx_prime = dn[16 + i];
            double const dyy_prime = dn[20 + i];

            double const a_prime = w_prime + x * dx_prime + y * dy_prime;
            double const ax_prime = 2.0 * dx_prime + x * dxx_prime + y * dxy_prime;
            double const ay_prime = 2.0 * dy_prime + y * dyy_prime + x * dxy_prime;
            double const t_prime_2 = (dx * dx_prime) + (dy * dy_prime)
                + f_sqr_inv * a * a_prime;
            double const n_prime = t_prime_2 / n;
            double const b_prime = (dx_prime * dxx + dx * dxx_prime)
                + (dy_prime * dxy + dy * dxy_prime)
                + f_sqr_inv * (a_prime * ax + a * ax_prime);
            double const c_prime = (dx_prime * dxy + dx * dxy_prime)
                + (dy_prime * dyy + dy * dyy_prime)
                + f_sqr_inv * (a_prime * ay + a * ay_prime);
            double const nx_prime = (b_prime * n - b * n_prime) / t;
            double const ny_prime = (c_prime * n - c * n_prime) / t;

            double const xx_prime = ((dxx_prime * n + dxx * n_prime
                - dx_prime * nx - dx * nx_prime) * t
                - (dxx * n - dx * nx) * t_prime_2 * 2.0) / (t * t);
            double const yy_prime = ((dyy_prime * n + dyy * n_prime
                - dy_prime * ny - dy * ny_prime) * t
                - (dyy * n - dy * ny) * t_prime_2 * 2.0) / (t * t);
            double const xy_prime = ((dxy_prime * n + dxy * n_prime
                - dx_prime * ny - dx * ny_prime) * t
                - (dxy * n - dx * ny) * t_prime_2 * 2.0) / (t * t);
            double const yx_prime = ((dxy_prime * n + dxy * n_prime
                - dy_prime * nx - dy * nx_prime) * t
                - (dxy * n - dy * nx) * t_prime_2 * 2.0) / (t * t);
            double const zx_prime = ((ax_prime * n + ax * n_prime
                - a_prime * nx - a * nx_prime) * t
                - (ax * n - a * nx) * t_prime_2 * 2.0) / (t * t * f);
            double const zy_prime = ((ay_prime * n + ay * n_prime
                - a_prime * ny - a * ny_prime) * t
                - (ay * n - a * ny) * t_prime_2 * 2.0) / (t * t * f);

            full_deriv[0 + node * 4 + i] = xx_prime;
            full_deriv[16 + node * 4 + i] = -yx_prime;
            full_deriv[32 + node * 4 + i] = zx_prime;
            full_deriv[48 + node * 4 + i] = xy_prime;
            full_deriv[64 + node * 4 + i] = -yy_prime;
            full_deriv[80 + node * 4 + i] = zy_prime;
        }
    }
}

/* ====================================================================== */
/* spherical harmonics                                                     */
/* ====================================================================== */

/* spherical_harmonics.h:53-73, 133-151 */
void
orc_sh_evaluate_4_band(const double *n, double *sh)
{
    sh[0] = 1.0;
    sh[1] = n[1];
    sh[2] = n[2];
    sh[3] = n[0];
    sh[4] = n[0] * n[1];
    sh[5] = n[1] * n[2];
    sh[6] = -POW2(n[0]) - POW2(n[1]) + 2.0 * POW2(n[2]);
    sh[7] = n[0] * n[2];
    sh[8] = n[0] * n[0] - n[1] * n[1];

    double const x2 = POW2(n[0]);
    double const y2 = POW2(n[1]);
    double const z2 = POW2(n[2]);
    sh[9] = (3.0 * x2 - y2) * n[1];
    sh[10] = n[0] * n[1] * n[2];
    sh[11] = (4.0 * z2 - x2 - y2) * n[1];
    sh[12] = (2.0 * z2 - 3.0 * x2 - 3.0 * y2) * n[2];
    sh[13] = (4.0 * z2 - x2 - y2) * n[0];
    sh[14] = (x2 - y2) * n[2];
    sh[15] = (x2 - 3.0 * y2) * n[0];
}

/* spherical_harmonics.h:79-127, 157-201 */
void
orc_sh_derivative_4_band(const double *n, double *d)
{
    d[0] = 0.0; d[1] = 0.0; d[2] = 0.0;
    d[3] = 0.0; d[4] = 1.0; d[5] = 0.0;
    d[6] = 0.0; d[7] = 0.0; d[8] = 1.0;
    d[9] = 1.0; d[10] = 0.0; d[11] = 0.0;
    d[12] = n[1]; d[13] = n[0]; d[14] = 0.0;
    d[15] = 0.0; d[16] = n[2]; d[17] = n[1];
    d[18] = -2.0 * n[0]; d[19] = -2.0 * n[1]; d[20] = 4.0 * n[2];
    d[21] = n[2]; d[22] = 0.0; d[23] = n[0];
    d[24] = 2.0 * n[0]; d[25] = -2.0 * n[1]; d[26] = 0.0;

    double const x2 = POW2(n[0]);
    double const y2 = POW2(n[1]);
    double const z2 = POW2(n[2]);
    d[27] = 6.0 * n[0] * n[1]; d[28] = 3.0 * (x2 - y2); d[29] = 0.0;
    d[30] = n[1] * n[2]; d[31] = n[0] * n[2]; d[32] = n[0] * n[1];
    d[33] = -2.0 * n[0] * n[1]; d[34] = 4.0 * z2 - x2 - 3.0 * y2;
    d[35] = 8.0 * n[1] * n[2];
    d[36] = -6.0 * n[0] * n[2]; d[37] = -6.0 * n[1] * n[2];
    d[38] = 6.0 * z2 - 3.0 * (x2 + y2);
    d[39] = 4.0 * z2 - 3.0 * x2 - y2; d[40] = -2.0 * n[0] * n[1];
    d[41] = 8.0 * n[0] * n[2];
    d[42] = 2.0 * n[0] * n[2]; d[43] = -2.0 * n[1] * n[2]; d[44] = x2 - y2;
    d[45] = 3.0 * (x2 - y2); d[46] = -6.0 * n[0] * n[1]; d[47] = 0.0;
}

/* ====================================================================== */
/* dense algebra                                                           */
/* ====================================================================== */

/* ldl_decomposition.h:43-92 (early return on a zero pivot leaves A as is) */
void
orc_ldl_inverse(double *A, int const size)
{
    double *L = (double *)calloc((size_t)size * size, sizeof(double));
    double *D = (double *)calloc((size_t)size, sizeof(double));

    for (int j = 0; j < size; ++j)
    {
        D[j] = A[j * size + j];
        L[j * size + j] = 1.0;
        for (int k = 0; k < j; ++k)
            D[j] -= (L[j * size + k] * L[j * size + k]) * D[k];
        if (D[j] == 0.0)
        {
            free(L);
            free(D);
            return;
        }
        for (int i = j + 1; i < size; ++i)
        {
            L[i * size + j] = A[i * size + j];
            for (int k = 0; k < j; ++k)
                L[i * size + j] -= L[i * size + k] * D[k] * L[j * size + k];
            L[i * size + j] /= D[j];
        }
    }
    for (int i = 0; i < size; ++i)
        for (int j = i + 1; j < size; ++j)
        {
            double sum = 0.0;
            for (int k = i; k < j; ++k)
                sum -= L[j * size + k] * L[k * size + i];
            L[j * size + i] = sum;
        }
    for (int i = 0; i < size; ++i)
        D[i] = 1.0 / D[i];

    /* combine_ldl, ldl_decomposition.h:20-36 */
    for (int i = 0; i < size * size; ++i)
        A[i] = 0.0;
    for (int r = 0; r < size; ++r)
        for (int c1 = 0; c1 < size; ++c1)
            for (int c2 = 0; c2 < size; ++c2)
                A[c1 * size + c2] += L[r * size + c2] * L[r * size + c1] * D[r];
    free(L);
    free(D);
}

/* sse_vector.cc:19-41: _mm_dp_pd(a, b, 0xFF) = a0*b0 + a1*b1 per pair,
 * pairs summed sequentially, odd tail added last. */
double
orc_vec_dot(const double *a, const double *b, size_t n)
{
    double ret = 0.0;
    size_t const dim = n / 2;
    for (size_t i = 0; i < dim; ++i)
        ret += a[2 * i] * b[2 * i] + a[2 * i + 1] * b[2 * i + 1];
    for (size_t i = n % 2; i > 0; --i)
        ret += a[n - i] * b[n - i];
    return ret;
}

/* sse_vector.cc:139-213: multiply_add (sign > 0) / multiply_sub: a +- b * f,
 * one multiplication and one addition per element in every branch. */
void
orc_vec_multiply_add(const double *a, const double *b, double factor,
    int sign, double *out, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        out[i] = sign > 0 ? a[i] + b[i] * factor : a[i] - b[i] * factor;
}

/* mve::Image<float>::linear_at [MVE-unverified: recalled semantics] */
float
orc_linear_at_f32(const float *img, int w, int h, int c, float x, float y,
    int ch)
{
    x = fmaxf(0.0f, fminf((float)(w - 1), x));
    y = fmaxf(0.0f, fminf((float)(h - 1), y));
    int const floor_x = (int)x;
    int const floor_y = (int)y;
    int const floor_xp1 = floor_x + 1 < w - 1 ? floor_x + 1 : w - 1;
    int const floor_yp1 = floor_y + 1 < h - 1 ? floor_y + 1 : h - 1;
    float const w1 = x - (float)floor_x;
    float const w0 = 1.0f - w1;
    float const w3 = y - (float)floor_y;
    float const w2 = 1.0f - w3;
    size_t const rowstride = (size_t)w * c;
    size_t const row1 = floor_y * rowstride;
    size_t const row2 = floor_yp1 * rowstride;
    size_t const col1 = (size_t)floor_x * c;
    size_t const col2 = (size_t)floor_xp1 * c;
    float const v1 = img[row1 + col1 + ch];
    float const v2 = img[row1 + col2 + ch];
    float const v3 = img[row2 + col1 + ch];
    float const v4 = img[row2 + col2 + ch];
    return v1 * (w0 * w2) + v2 * (w1 * w2) + v3 * (w0 * w3) + v4 * (w1 * w3);
}

/* uint8 specialisation: interpolate in float, +0.5f, truncate */
uint8_t
orc_linear_at_u8(const uint8_t *img, int w, int h, int c, float x, float y,
    int ch)
{
    x = fmaxf(0.0f, fminf((float)(w - 1), x));
    y = fmaxf(0.0f, fminf((float)(h - 1), y));
    int const floor_x = (int)x;
    int const floor_y = (int)y;
    int const floor_xp1 = floor_x + 1 < w - 1 ? floor_x + 1 : w - 1;
    int const floor_yp1 = floor_y + 1 < h - 1 ? floor_y + 1 : h - 1;
    float const w1 = x - (float)floor_x;
    float const w0 = 1.0f - w1;
    float const w3 = y - (float)floor_y;
    float const w2 = 1.0f - w3;
    size_t const rowstride = (size_t)w * c;
    size_t const row1 = floor_y * rowstride;
    size_t const row2 = floor_yp1 * rowstride;
    size_t const col1 = (size_t)floor_x * c;
    size_t const col2 = (size_t)floor_xp1 * c;
    float const v1 = (float)img[row1 + col1 + ch];
    float const v2 = (float)img[row1 + col2 + ch];
    float const v3 = (float)img[row2 + col1 + ch];
    float const v4 = (float)img[row2 + col2 + ch];
    return (uint8_t)(v1 * (w0 * w2) + v2 * (w1 * w2) + v3 * (w0 * w3)
        + v4 * (w1 * w3) + 0.5f);
}

/* ====================================================================== */
/* Gauss-Newton step                                                       */
/* ====================================================================== */

static void
patch_node_ids(const orc_surface *s, int patch_id, int *ids)
{
    /* surface.cc:283-298 */
    int const idx = patch_id % s->npx;
    int const idy = patch_id / s->npx;
    int const stride = s->npx + 1;
    ids[0] = idy * stride + idx;
    ids[1] = idy * stride + idx + 1;
    ids[2] = (idy + 1) * stride + idx;
    ids[3] = (idy + 1) * stride + idx + 1;
}

static void
patch_nodes(const orc_surface *s, int patch_id, double *nodes16)
{
    int ids[4];
    patch_node_ids(s, patch_id, ids);
    for (int n = 0; n < 4; ++n)
        memcpy(nodes16 + 4 * n, s->nodes + 4 * (size_t)ids[n],
            4 * sizeof(double));
}

static void
patch_pixel_origin(const orc_surface *s, int patch_id, int *px, int *py)
{
    /* surface.cc:185-197 */
    *px = s->start_x + (patch_id % s->npx) * s->patchsize;
    *py = s->start_y + (patch_id / s->npx) * s->patchsize;
}

#define ORC_MAX_SUBS 32

/* Which restatement of gauss_newton_step.cc:252-383 orc_gn_patch runs:
 *   0  the SSE4.1 branch (:252-333) with SSE2 intrinsics, two residual
 *      components per register like the reference (default: the reference
 *      build defines __SSE4_1__, lib/Makefile:4);
 *   1  the same branch with the two lanes written out as scalars (must be
 *      bit-identical to 0: every lane operation is the same IEEE operation);
 *   2  the reference's scalar fallback (:335-383): same sums in a different
 *      association, agrees with 0 to rounding (tests/test_oracle_core.py). */
static int g_k2_mode = 0;
void orc_set_k2_mode(int mode) { g_k2_mode = mode; }

/* OpenMP threads of the per-patch loops (1 = the reference's behaviour per
 * view: it parallelises over views, app/smvsrecon.cc:658-733, not inside). */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

static void
k2_photometric(int num_subs, double (*j_grad_subs)[2],
    double (*jac_entries)[16][2], const double *grad_main, double *gradient,
    double *hessian_entries)
{
    if (g_k2_mode == 2)
    {
        /* :335-383 */
        double weight[2], subweight[2], diff[2], subdiff[2], jace[2];
        for (int j = 0; j < num_subs; ++j)
        {
            diff[0] = j_grad_subs[j][0] - grad_main[0];
            diff[1] = j_grad_subs[j][1] - grad_main[1];
            weight[0] = 1.0 / (R_FACTOR + fabs(diff[0]));
            weight[1] = 1.0 / (R_FACTOR + fabs(diff[1]));
            for (int col = 0; col < 16; ++col)
            {
                gradient[col] += (diff[0] * weight[0] * jac_entries[j][col][0]
                    + diff[1] * weight[1] * jac_entries[j][col][1]);
                for (int col2 = col; col2 < 16; ++col2)
                    hessian_entries[col * 16 + col2] +=
                        (jac_entries[j][col][0] * weight[0]
                            * jac_entries[j][col2][0]
                        + jac_entries[j][col][1] * weight[1]
                            * jac_entries[j][col2][1]);
            }
            for (int j2 = j + 1; j2 < num_subs; ++j2)
            {
                subdiff[0] = j_grad_subs[j][0] - j_grad_subs[j2][0];
                subdiff[1] = j_grad_subs[j][1] - j_grad_subs[j2][1];
                subweight[0] = 1.0 / (R_FACTOR + fabs(subdiff[0]));
                subweight[1] = 1.0 / (R_FACTOR + fabs(subdiff[1]));
                for (int col = 0; col < 16; ++col)
                {
                    jace[0] = (jac_entries[j][col][0]
                        - jac_entries[j2][col][0]) * subweight[0];
                    jace[1] = (jac_entries[j][col][1]
                        - jac_entries[j2][col][1]) * subweight[1];
                    gradient[col] += (jace[0] * subdiff[0]
                        + jace[1] * subdiff[1]);
                    for (int col2 = col; col2 < 16; ++col2)
                        hessian_entries[col * 16 + col2] +=
                            (jace[0] * (jac_entries[j][col2][0]
                                - jac_entries[j2][col2][0])
                            + jace[1] * (jac_entries[j][col2][1]
                                - jac_entries[j2][col2][1]));
                }
            }
        }
        return;
    }
#if defined(__SSE2__)
    if (g_k2_mode == 0)
    {
        /* :252-333: the two residual components live in the two lanes of a
         * register and are only summed at the end (:323-332) */
        __m128d reg_grad[16];
        __m128d reg_hessian[256];
        __m128d const reg_grad_main = _mm_set_pd(grad_main[1], grad_main[0]);
        __m128d const sign_mask = _mm_set1_pd(-0.);
        __m128d const reg_rfactor = _mm_set1_pd(R_FACTOR);
        for (int col = 0; col < 16; ++col)
            reg_grad[col] = _mm_setzero_pd();
        for (int col = 0; col < 16; ++col)
            for (int col2 = col; col2 < 16; ++col2)
                reg_hessian[col * 16 + col2] = _mm_setzero_pd();
        for (int j = 0; j < num_subs; ++j)
        {
            __m128d const reg_jgrad_sub = _mm_loadu_pd(j_grad_subs[j]);
            __m128d const reg_diff = _mm_sub_pd(reg_jgrad_sub, reg_grad_main);
            __m128d const reg_weight = _mm_add_pd(
                _mm_andnot_pd(sign_mask, reg_diff), reg_rfactor);
            for (int col = 0; col < 16; ++col)
            {
                __m128d const jcol = _mm_loadu_pd(jac_entries[j][col]);
                reg_grad[col] = _mm_add_pd(reg_grad[col],
                    _mm_div_pd(_mm_mul_pd(reg_diff, jcol), reg_weight));
                for (int col2 = col; col2 < 16; ++col2)
                    reg_hessian[col * 16 + col2] = _mm_add_pd(
                        reg_hessian[col * 16 + col2], _mm_mul_pd(jcol,
                            _mm_div_pd(_mm_loadu_pd(jac_entries[j][col2]),
                                reg_weight)));
            }
            for (int j2 = j + 1; j2 < num_subs; ++j2)
            {
                __m128d const reg_subdiff = _mm_sub_pd(reg_jgrad_sub,
                    _mm_loadu_pd(j_grad_subs[j2]));
                __m128d const reg_subweight = _mm_add_pd(
                    _mm_andnot_pd(sign_mask, reg_subdiff), reg_rfactor);
                for (int col = 0; col < 16; ++col)
                {
                    __m128d const reg_jace = _mm_div_pd(_mm_sub_pd(
                        _mm_loadu_pd(jac_entries[j][col]),
                        _mm_loadu_pd(jac_entries[j2][col])), reg_subweight);
                    reg_grad[col] = _mm_add_pd(reg_grad[col],
                        _mm_mul_pd(reg_jace, reg_subdiff));
                    for (int col2 = col; col2 < 16; ++col2)
                        reg_hessian[col * 16 + col2] = _mm_add_pd(
                            reg_hessian[col * 16 + col2], _mm_mul_pd(reg_jace,
                                _mm_sub_pd(_mm_loadu_pd(jac_entries[j][col2]),
                                    _mm_loadu_pd(jac_entries[j2][col2]))));
                }
            }
        }
        for (int col = 0; col < 16; ++col)
        {
            double lanes[2];
            _mm_storeu_pd(lanes, reg_grad[col]);
            gradient[col] += lanes[0] + lanes[1];
        }
        for (int col = 0; col < 16; ++col)
            for (int col2 = col; col2 < 16; ++col2)
            {
                double lanes[2];
                _mm_storeu_pd(lanes, reg_hessian[col * 16 + col2]);
                hessian_entries[col * 16 + col2] += lanes[0] + lanes[1];
            }
        return;
    }
#endif
    {
        double reg_grad[16][2];
        double reg_hessian[256][2];
        memset(reg_grad, 0, sizeof(reg_grad));
        memset(reg_hessian, 0, sizeof(reg_hessian));
        for (int j = 0; j < num_subs; ++j)
        {
            double reg_diff[2], reg_weight[2];
            for (int k = 0; k < 2; ++k)
            {
                reg_diff[k] = j_grad_subs[j][k] - grad_main[k];
                reg_weight[k] = fabs(reg_diff[k]) + R_FACTOR;
            }
            for (int col = 0; col < 16; ++col)
                for (int k = 0; k < 2; ++k)
                {
                    double const jcol = jac_entries[j][col][k];
                    reg_grad[col][k] = reg_grad[col][k]
                        + (reg_diff[k] * jcol) / reg_weight[k];
                    for (int col2 = col; col2 < 16; ++col2)
                        reg_hessian[col * 16 + col2][k] =
                            reg_hessian[col * 16 + col2][k]
                            + jcol * (jac_entries[j][col2][k] / reg_weight[k]);
                }
            for (int j2 = j + 1; j2 < num_subs; ++j2)
            {
                double reg_subdiff[2], reg_subweight[2];
                for (int k = 0; k < 2; ++k)
                {
                    reg_subdiff[k] = j_grad_subs[j][k] - j_grad_subs[j2][k];
                    reg_subweight[k] = fabs(reg_subdiff[k]) + R_FACTOR;
                }
                for (int col = 0; col < 16; ++col)
                    for (int k = 0; k < 2; ++k)
                    {
                        double const jace = (jac_entries[j][col][k]
                            - jac_entries[j2][col][k]) / reg_subweight[k];
                        reg_grad[col][k] = reg_grad[col][k]
                            + jace * reg_subdiff[k];
                        for (int col2 = col; col2 < 16; ++col2)
                            reg_hessian[col * 16 + col2][k] =
                                reg_hessian[col * 16 + col2][k]
                                + jace * (jac_entries[j][col2][k]
                                    - jac_entries[j2][col2][k]);
                    }
            }
        }
        for (int col = 0; col < 16; ++col)
            gradient[col] += reg_grad[col][0] + reg_grad[col][1];
        for (int col = 0; col < 16; ++col)
            for (int col2 = col; col2 < 16; ++col2)
                hessian_entries[col * 16 + col2] +=
                    reg_hessian[col * 16 + col2][0]
                    + reg_hessian[col * 16 + col2][1];
    }
}

/* gauss_newton_step.cc:145-518 */
void
orc_gn_patch(const orc_views *views, const orc_surface *surf,
    const orc_gn_options *opts, const double *lighting, int patch_id,
    const double *node_derivatives, double *gradient, double *hessian_entries)
{
    int const scale = surf->scale;
    int const size = surf->patchsize;
    int sub_ids[ORC_MAX_SUBS];
    int num_subs = 0;
    for (int j = 0; j < views->n_subs && j < ORC_MAX_SUBS; ++j)
        if (surf->patch_vis[patch_id] & (1u << j))
            sub_ids[num_subs++] = j;

    int sampling = 4; /* :157-161 */
    if (scale < 5)
        sampling = 2;
    if (scale < 3)
        sampling = 1;

    int const max_px = size * size;
    double *pixels = (double *)malloc(sizeof(double) * 2 * max_px);
    double *depths = (double *)malloc(sizeof(double) * max_px);
    double *dd = (double *)malloc(sizeof(double) * 2 * max_px);
    double *dd2 = (double *)malloc(sizeof(double) * 3 * max_px);
    int *pids = (int *)malloc(sizeof(int) * max_px);

    double nodes16[16];
    patch_nodes(surf, patch_id, nodes16);
    int px0, py0;
    patch_pixel_origin(surf, patch_id, &px0, &py0);
    int const npix = orc_patch_values_at_pixels(nodes16, px0, py0, size,
        sampling, pixels, depths, dd, dd2, pids);

    double j_grad_subs[ORC_MAX_SUBS][2];
    double jac_entries[ORC_MAX_SUBS][16][2];
    double full_surface_div[6];
    double full_surface_div_deriv[96];
    double normal_deriv[48];
    double const flen = (double)views->flen;
    int const W = views->width;
    int const H = views->height;

    for (int i = 0; i < npix; ++i)
    {
        int const pxi = (int)pixels[2 * i + 0];
        int const pyi = (int)pixels[2 * i + 1];
        double grad_main[2];
        grad_main[0] = views->grad[((size_t)pyi * W + pxi) * 2 + 0];
        grad_main[1] = views->grad[((size_t)pyi * W + pxi) * 2 + 1];
        const double *dn00 = node_derivatives + (size_t)pids[i] * 96;

        for (int j = 0; j < num_subs; ++j)
        {
            int const sub_id = sub_ids[j];
            const orc_subview *sv = &views->subs[sub_id];
            orc_corr C;
            orc_corr_update(&C, views->M + 9 * sub_id, views->t + 3 * sub_id,
                pixels[2 * i] + 0.5, pixels[2 * i + 1] + 0.5, depths[i],
                dd[2 * i], dd[2 * i + 1]);
            double proj[2], jac[4];
            orc_corr_fill(&C, proj);
            orc_corr_fill_jacobian(&C, jac);
            proj[0] -= 0.5;
            proj[1] -= 0.5;

            double grad_sub[2], hess_sub[4];
            grad_sub[0] = orc_linear_at_f32(sv->grad, sv->width, sv->height,
                2, (float)proj[0], (float)proj[1], 0);
            grad_sub[1] = orc_linear_at_f32(sv->grad, sv->width, sv->height,
                2, (float)proj[0], (float)proj[1], 1);
            hess_sub[0] = orc_linear_at_f32(sv->hess, sv->width, sv->height,
                3, (float)proj[0], (float)proj[1], 0);
            hess_sub[1] = orc_linear_at_f32(sv->hess, sv->width, sv->height,
                3, (float)proj[0], (float)proj[1], 1);
            hess_sub[2] = hess_sub[1];
            hess_sub[3] = orc_linear_at_f32(sv->hess, sv->width, sv->height,
                3, (float)proj[0], (float)proj[1], 2);

            /* j_grad_subs[j] = jac * grad_sub (:200) */
            j_grad_subs[j][0] = jac[0] * grad_sub[0] + jac[1] * grad_sub[1];
            j_grad_subs[j][1] = jac[2] * grad_sub[0] + jac[3] * grad_sub[1];

            double c_dn[32], jac_dn[32];
            orc_corr_fill_derivative(&C, dn00, c_dn);
            orc_corr_fill_jacobian_derivative_grad(&C, grad_sub, dn00, jac_dn);

            /* jac_hess = jac * hess_sub (:205) */
            double jac_hess[4];
            jac_hess[0] = jac[0] * hess_sub[0] + jac[1] * hess_sub[2];
            jac_hess[1] = jac[0] * hess_sub[1] + jac[1] * hess_sub[3];
            jac_hess[2] = jac[2] * hess_sub[0] + jac[3] * hess_sub[2];
            jac_hess[3] = jac[2] * hess_sub[1] + jac[3] * hess_sub[3];
            for (int col = 0; col < 16; ++col)
            {
                double const v0 = jac_hess[0] * c_dn[2 * col]
                    + jac_hess[1] * c_dn[2 * col + 1];
                double const v1 = jac_hess[2] * c_dn[2 * col]
                    + jac_hess[3] * c_dn[2 * col + 1];
                jac_entries[j][col][0] = jac_dn[2 * col] + v0;
                jac_entries[j][col][1] = jac_dn[2 * col + 1] + v1;
            }
        }

        double basic_regularizer_weight = 0.0;
        if (opts->regularization > 0.0) /* :210-240 */
        {
            double const abs_sum = fabs(grad_main[0]) + fabs(grad_main[1]);
            basic_regularizer_weight = opts->regularization * 0.005
                / (0.03 > abs_sum ? 0.03 : abs_sum);
            double const x = pixels[2 * i] + 0.5 - (double)W / 2.0;
            double const y = pixels[2 * i + 1] + 0.5 - (double)H / 2.0;
            orc_normal_divergence(x, y, flen, depths[i], dd[2 * i],
                dd[2 * i + 1], dd2[3 * i], dd2[3 * i + 1], dd2[3 * i + 2],
                full_surface_div);
            orc_normal_divergence_deriv(dn00, x, y, flen, depths[i],
                dd[2 * i], dd[2 * i + 1], dd2[3 * i], dd2[3 * i + 1],
                dd2[3 * i + 2], full_surface_div_deriv);
            orc_normal_derivative(dn00, x, y, flen, depths[i], dd[2 * i],
                dd[2 * i + 1], normal_deriv);
        }

        /* ---- fill_gradient_and_hessian_entries (:246-518), photometric
         * part (:252-383) ---- */
        k2_photometric(num_subs, j_grad_subs, jac_entries, grad_main, gradient,
            hessian_entries);

        if (opts->regularization <= 0.0)
            continue;

        size_t const num_diffs = ((size_t)num_subs * (num_subs + 1)) / 2;
        basic_regularizer_weight *= num_diffs;
        if (lighting == NULL || opts->light_surf_regularization > 0.0)
        {
            double geom_weight = 1.0;
            if (lighting != NULL)
                geom_weight *= opts->light_surf_regularization / 100;
            for (int v = 0; v < 6; ++v)
            {
                double const weight = geom_weight
                    / (R_FACTOR + fabs(full_surface_div[v]));
                for (int col = 0; col < 16; ++col)
                {
                    gradient[col] += full_surface_div_deriv[16 * v + col]
                        * full_surface_div[v]
                        * basic_regularizer_weight * weight;
                    for (int col2 = col; col2 < 16; ++col2)
                        hessian_entries[col * 16 + col2] +=
                            full_surface_div_deriv[v * 16 + col]
                            * full_surface_div_deriv[v * 16 + col2]
                            * basic_regularizer_weight * weight;
                }
            }
            if (lighting == NULL)
                continue;
        }

        /* shading based energy term (:420-517) */
        double normal[3];
        double const x = pixels[2 * i] + 0.5 - (double)W / 2.0;
        double const y = pixels[2 * i + 1] + 0.5 - (double)H / 2.0;
        orc_fill_normal(x, y, (double)views->inv_flen, depths[i], dd[2 * i],
            dd[2 * i + 1], normal);
        double sh_deriv[48];
        orc_sh_derivative_4_band(normal, sh_deriv);
        double sh_basis[16];
        orc_sh_evaluate_4_band(normal, sh_basis);
        double shading = 0.0;
        for (int l = 0; l < 16; ++l)
            shading += lighting[l] * sh_basis[l];

        double lig[2];
        lig[0] = views->shading_grad[((size_t)pyi * W + pxi) * 2 + 0];
        lig[1] = views->shading_grad[((size_t)pyi * W + pxi) * 2 + 1];
        double const linear_image_value = views->shading[(size_t)pyi * W + pxi];
        double const shading_weight = 0.001 * num_diffs
            / (R_FACTOR + (fabs(lig[0]) + fabs(lig[1])));
        if (sqrt(lig[0] * lig[0] + lig[1] * lig[1]) < 1e-10)
            continue;
        if (POW2(shading) < 1e-10 || POW2(linear_image_value) < 1e-10)
            continue;

        double shading_grad[2] = { 0.0, 0.0 };
        for (int l = 1; l < 16; ++l)
        {
            shading_grad[0] += lighting[l] * (
                sh_deriv[l * 3 + 0] * full_surface_div[0]
                + sh_deriv[l * 3 + 1] * full_surface_div[1]
                + sh_deriv[l * 3 + 2] * full_surface_div[2]);
            shading_grad[1] += lighting[l] * (
                sh_deriv[l * 3 + 0] * full_surface_div[3]
                + sh_deriv[l * 3 + 1] * full_surface_div[4]
                + sh_deriv[l * 3 + 2] * full_surface_div[5]);
        }
        double render_grad[2];
        render_grad[0] = shading_grad[0] / shading;
        render_grad[1] = shading_grad[1] / shading;
        double const inv_liv = 1.0 / linear_image_value;
        lig[0] *= inv_liv;
        lig[1] *= inv_liv;
        double shading_error[2];
        shading_error[0] = render_grad[0] - lig[0];
        shading_error[1] = render_grad[1] - lig[1];

        double shading_deriv[16];
        double shading_grad_deriv[16][2];
        double render_deriv[16][2];
        for (int col = 0; col < 16; ++col)
        {
            shading_deriv[col] = 0;
            for (int l = 1; l < 16; ++l)
                shading_deriv[col] += lighting[l] * (
                    sh_deriv[l * 3 + 0] * normal_deriv[0 + col]
                    + sh_deriv[l * 3 + 1] * normal_deriv[16 + col]
                    + sh_deriv[l * 3 + 2] * normal_deriv[32 + col]);
        }
        for (int col = 0; col < 16; ++col)
        {
            shading_grad_deriv[col][0] = 0.0;
            shading_grad_deriv[col][1] = 0.0;
            for (int l = 1; l < 16; ++l)
            {
                shading_grad_deriv[col][0] += lighting[l] * (
                    sh_deriv[l * 3 + 0] * full_surface_div_deriv[16 * 0 + col]
                    + sh_deriv[l * 3 + 1] * full_surface_div_deriv[16 * 1 + col]
                    + sh_deriv[l * 3 + 2] * full_surface_div_deriv[16 * 2 + col]);
                shading_grad_deriv[col][1] += lighting[l] * (
                    sh_deriv[l * 3 + 0] * full_surface_div_deriv[16 * 3 + col]
                    + sh_deriv[l * 3 + 1] * full_surface_div_deriv[16 * 4 + col]
                    + sh_deriv[l * 3 + 2] * full_surface_div_deriv[16 * 5 + col]);
            }
        }
        for (int col = 0; col < 16; ++col)
            for (int k = 0; k < 2; ++k)
                render_deriv[col][k] = (shading_grad_deriv[col][k] * shading
                    - shading_grad[k] * shading_deriv[col]) / POW2(shading);

        double weight[2];
        weight[0] = 1.0 / (R_FACTOR + fabs(shading_error[0]));
        weight[1] = 1.0 / (R_FACTOR + fabs(shading_error[1]));
        weight[0] *= shading_weight;
        weight[1] *= shading_weight;
        for (int col = 0; col < 16; ++col)
        {
            gradient[col] += shading_error[0] * render_deriv[col][0] * weight[0]
                + shading_error[1] * render_deriv[col][1] * weight[1];
            for (int col2 = col; col2 < 16; ++col2)
                hessian_entries[col * 16 + col2] +=
                    render_deriv[col][0] * render_deriv[col2][0] * weight[0]
                    + render_deriv[col][1] * render_deriv[col2][1] * weight[1];
        }
    }

    free(pixels);
    free(depths);
    free(dd);
    free(dd2);
    free(pids);
}

/* slot of block (row node r, col node c) in the 9-point stencil of r */
static int
stencil_slot(int stride, int r, int c)
{
    int const ry = r / stride, rx = r % stride;
    int const cy = c / stride, cx = c % stride;
    return (cy - ry + 1) * 3 + (cx - rx + 1);
}

/* gauss_newton_step.cc:33-143 */
int
orc_gn_construct(const orc_views *views, const orc_surface *surf,
    const orc_gn_options *opts, const double *lighting,
    const uint8_t *active_nodes, double *H9, uint8_t *present9, double *g,
    double *P)
{
    int const stride = surf->npx + 1;
    int const num_nodes = stride * (surf->npy + 1);
    int const num_patches = surf->npx * surf->npy;
    int const ppp = surf->patchsize * surf->patchsize;

    double *node_derivatives = (double *)malloc(sizeof(double) * 96 * ppp);
    for (int i = 0; i < ppp; ++i)
        orc_node_derivatives_for_pixel(i, surf->patchsize,
            node_derivatives + 96 * (size_t)i);

    memset(g, 0, sizeof(double) * 4 * num_nodes);
    memset(H9, 0, sizeof(double) * 144 * (size_t)num_nodes);
    memset(present9, 0, 9 * (size_t)num_nodes);
    memset(P, 0, sizeof(double) * 16 * (size_t)num_nodes);

    /* The per-patch systems are independent: chunks of patches are evaluated
     * by g_threads OpenMP threads, then scattered sequentially in ascending
     * patch order -- the order the reference's patch loop feeds its std::map
     * (:64-122), so the sums are the same with any thread count. */
    enum { CHUNK = 4096 };
    double *chunk_g = (double *)malloc(sizeof(double) * 16 * CHUNK);
    double *chunk_h = (double *)malloc(sizeof(double) * 256 * CHUNK);
    uint8_t *chunk_live = (uint8_t *)malloc(CHUNK);
    int evaluated = 0;
    for (int base = 0; base < num_patches; base += CHUNK)
    {
        int const count = num_patches - base < CHUNK ? num_patches - base : CHUNK;
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
#endif
        for (int k = 0; k < count; ++k)
        {
            int const patch_id = base + k;
            chunk_live[k] = 0;
            if (!surf->patch_valid[patch_id])
                continue;
            int node_ids[4];
            patch_node_ids(surf, patch_id, node_ids);
            if (active_nodes[node_ids[0]] == 0 && active_nodes[node_ids[1]] == 0
                && active_nodes[node_ids[2]] == 0
                && active_nodes[node_ids[3]] == 0)
                continue;
            chunk_live[k] = 1;
            double *sub_gradient = chunk_g + 16 * (size_t)k;
            double *sub_hessian = chunk_h + 256 * (size_t)k;
            memset(sub_gradient, 0, sizeof(double) * 16);
            memset(sub_hessian, 0, sizeof(double) * 256);
            orc_gn_patch(views, surf, opts, lighting, patch_id,
                node_derivatives, sub_gradient, sub_hessian);
        }
        for (int k = 0; k < count; ++k)
        {
            if (!chunk_live[k])
                continue;
            int const patch_id = base + k;
            int node_ids[4];
            patch_node_ids(surf, patch_id, node_ids);
            evaluated += 1;
            const double *sub_gradient = chunk_g + 16 * (size_t)k;
            const double *sub_hessian = chunk_h + 256 * (size_t)k;

            for (int node = 0; node < 4; ++node) /* :89-96 */
            {
                if (active_nodes[node_ids[node]] == 0)
                    continue;
                for (int value = 0; value < 4; ++value)
                    g[node_ids[node] * 4 + value] +=
                        sub_gradient[node * 4 + value];
            }
            for (int node1 = 0; node1 < 16; ++node1) /* :99-121 */
            {
                if (active_nodes[node_ids[node1 / 4]] == 0)
                    continue;
                for (int node2 = node1; node2 < 16; ++node2)
                {
                    if (active_nodes[node_ids[node2 / 4]] == 0)
                        continue;
                    int const n1 = node_ids[node1 / 4];
                    int const n2 = node_ids[node2 / 4];
                    int const ox = node1 % 4;
                    int const oy = node2 % 4;
                    /* block_id1 -> (row 4*n2, col 4*n1), values[ox + 4*oy] */
                    {
                        int const s = stencil_slot(stride, n2, n1);
                        present9[n2 * 9 + s] = 1;
                        H9[((size_t)n2 * 9 + s) * 16 + ox + 4 * oy] +=
                            sub_hessian[node1 * 16 + node2];
                    }
                    if (node1 != node2)
                    {
                        /* block_id2 -> (row 4*n1, col 4*n2), values[ox*4 + oy] */
                        int const s = stencil_slot(stride, n1, n2);
                        present9[n1 * 9 + s] = 1;
                        H9[((size_t)n1 * 9 + s) * 16 + ox * 4 + oy] +=
                            sub_hessian[node1 * 16 + node2];
                    }
                }
            }
        }
    }
    free(chunk_g);
    free(chunk_h);
    free(chunk_live);

    /* preconditioner: diagonal blocks, inverted (:131-142,
     * block_sparse_matrix.h:300-316: kept un-inverted on NaN) */
    for (int n = 0; n < num_nodes; ++n)
    {
        if (!present9[n * 9 + 4])
            continue;
        double b[16];
        memcpy(b, H9 + ((size_t)n * 9 + 4) * 16, sizeof(b));
        orc_ldl_inverse(b, 4);
        int nancheck = 0;
        for (int i = 0; i < 16; ++i)
            if (isnan(b[i]))
                nancheck = 1;
        if (nancheck)
            memcpy(P + 16 * (size_t)n, H9 + ((size_t)n * 9 + 4) * 16,
                sizeof(b));
        else
            memcpy(P + 16 * (size_t)n, b, sizeof(b));
    }
    free(node_derivatives);
    return evaluated;
}

/* block_sparse_matrix.h:276-298: for every block column i (ascending), for
 * every block in it (ascending row): ret[row..] += B * rhs[col..]. */
void
orc_block_spmv(int num_nodes, int stride, const double *H9,
    const uint8_t *present9, const double *x, double *y)
{
    memset(y, 0, sizeof(double) * 4 * (size_t)num_nodes);
    for (int i = 0; i < num_nodes; ++i)
    {
        int const ix = i % stride, iy = i / stride;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
            {
                int const rx = ix + dx, ry = iy + dy;
                if (rx < 0 || rx >= stride || ry < 0)
                    continue;
                int const r = ry * stride + rx;
                if (r >= num_nodes)
                    continue;
                int const s = stencil_slot(stride, r, i);
                if (!present9[r * 9 + s])
                    continue;
                const double *v = H9 + ((size_t)r * 9 + s) * 16;
                int block_id = 0;
                for (int br = 0; br < 4; ++br)
                    for (int bc = 0; bc < 4; ++bc)
                        y[4 * r + br] += v[block_id++] * x[4 * i + bc];
            }
    }
}

static void
precond_apply(int num_nodes, const double *P, const uint8_t *present9,
    const double *r, double *z)
{
    /* precond is a block-diagonal BlockSparseMatrix: multiply() zero-fills
     * and adds B*r per present block */
    for (int n = 0; n < num_nodes; ++n)
    {
        for (int br = 0; br < 4; ++br)
            z[4 * n + br] = 0.0;
        if (!present9[n * 9 + 4])
            continue;
        const double *v = P + 16 * (size_t)n;
        int block_id = 0;
        for (int br = 0; br < 4; ++br)
            for (int bc = 0; bc < 4; ++bc)
                z[4 * n + br] += v[block_id++] * r[4 * n + bc];
    }
}

/* conjugate_gradient.h:72-202 */
int
orc_cg_solve(int num_nodes, int stride, const double *H9,
    const uint8_t *present9, const double *P, const double *b, double *x,
    int max_iterations, double error_tolerance, double q_tolerance,
    int *num_iterations)
{
    size_t const n = 4 * (size_t)num_nodes;
    double *r = (double *)malloc(sizeof(double) * n);
    double *d = (double *)malloc(sizeof(double) * n);
    double *z = (double *)malloc(sizeof(double) * n);
    double *Ad = (double *)malloc(sizeof(double) * n);
    double *tmp = (double *)malloc(sizeof(double) * n);
    int info = 1;

    memset(x, 0, sizeof(double) * n);
    memcpy(r, b, sizeof(double) * n);
    precond_apply(num_nodes, P, present9, r, z);
    double r_dot_r = orc_vec_dot(z, r, n);
    memcpy(d, z, sizeof(double) * n);

    for (size_t i = 0; i < n; ++i)
        tmp[i] = b[i] + r[i];
    double Q0 = -1.0 * orc_vec_dot(x, tmp, n);

    int it;
    for (it = 1; it < max_iterations; it += 1)
    {
        orc_block_spmv(num_nodes, stride, H9, present9, d, Ad);
        double const alpha = r_dot_r / orc_vec_dot(d, Ad, n);
        for (size_t i = 0; i < n; ++i)
            x[i] = x[i] + alpha * d[i];
        for (size_t i = 0; i < n; ++i)
            r[i] = r[i] - alpha * Ad[i];
        double new_r_dot_r = orc_vec_dot(r, r, n);
        if (new_r_dot_r < error_tolerance)
        {
            info = 0;
            break;
        }
        for (size_t i = 0; i < n; ++i)
            tmp[i] = b[i] + r[i];
        double const Q1 = -1.0 * orc_vec_dot(x, tmp, n);
        double const zeta = it * (Q1 - Q0) / Q1;
        if (zeta < q_tolerance)
        {
            info = 0;
            break;
        }
        Q0 = Q1;
        precond_apply(num_nodes, P, present9, r, z);
        new_r_dot_r = orc_vec_dot(z, r, n);
        double const beta = new_r_dot_r / r_dot_r;
        for (size_t i = 0; i < n; ++i)
            d[i] = z[i] + beta * d[i];
        r_dot_r = new_r_dot_r;
    }
    *num_iterations = it;
    free(r); free(d); free(z); free(Ad); free(tmp);
    return info;
}

/* depth_optimizer.cc:647-677 for the patches touched by active_nodes;
 * proj holds, per (patch, sub j, pixel i), one projection (the reference
 * stores it 4x, once per node id). Returns number of entries. */
static size_t
node_reprojections(const orc_views *views, const orc_surface *surf,
    const uint8_t *active_nodes, double *proj /* may be NULL: count only */)
{
    int const num_patches = surf->npx * surf->npy;
    int const size = surf->patchsize;
    double *pixels = (double *)malloc(sizeof(double) * 2 * size * size);
    double *depths = (double *)malloc(sizeof(double) * size * size);
    size_t count = 0;
    for (int patch_id = 0; patch_id < num_patches; ++patch_id)
    {
        if (!surf->patch_valid[patch_id])
            continue;
        int ids[4];
        patch_node_ids(surf, patch_id, ids);
        if ((active_nodes[ids[0]] + active_nodes[ids[1]]
            + active_nodes[ids[2]] + active_nodes[ids[3]]) == 0)
            continue;
        double nodes16[16];
        patch_nodes(surf, patch_id, nodes16);
        int px0, py0;
        patch_pixel_origin(surf, patch_id, &px0, &py0);
        int const npix = orc_patch_values_at_pixels(nodes16, px0, py0, size,
            1, pixels, depths, NULL, NULL, NULL);
        for (int j = 0; j < views->n_subs; ++j)
        {
            if (!(surf->patch_vis[patch_id] & (1u << j)))
                continue;
            for (int i = 0; i < npix; ++i)
            {
                if (proj != NULL)
                {
                    orc_corr C;
                    orc_corr_update(&C, views->M + 9 * j, views->t + 3 * j,
                        pixels[2 * i], pixels[2 * i + 1], depths[i], 0, 0);
                    orc_corr_fill(&C, proj + 2 * count);
                }
                count += 1;
            }
        }
    }
    free(pixels);
    free(depths);
    return count;
}

/* depth_optimizer.cc:271-303 + surface.cc:957-981 */
int
orc_update_and_reactivate(const orc_views *views, orc_surface *surf,
    const double *delta, uint8_t *active_nodes, int full_optimization,
    double *mean_delta)
{
    int const stride = surf->npx + 1;
    int const num_nodes = stride * (surf->npy + 1);
    int const num_patches = surf->npx * surf->npy;
    size_t const count = node_reprojections(views, surf, active_nodes, NULL);
    double *proj1 = (double *)malloc(sizeof(double) * 2 * (count + 1));
    double *proj2 = (double *)malloc(sizeof(double) * 2 * (count + 1));
    node_reprojections(views, surf, active_nodes, proj1);

    for (int i = 0; i < num_nodes; ++i) /* update_nodes */
    {
        if (!surf->node_valid[i])
            continue;
        surf->nodes[4 * i + 0] += delta[4 * i + 0];
        surf->nodes[4 * i + 1] += delta[4 * i + 1];
        surf->nodes[4 * i + 2] += delta[4 * i + 2];
        surf->nodes[4 * i + 3] += delta[4 * i + 3];
    }
    node_reprojections(views, surf, active_nodes, proj2);

    int result = -1;
    if (full_optimization)
    {
        /* every entry appears 4x in the reference's list (:673-674) */
        double sum_diff = 0;
        for (size_t p = 0; p < count; ++p)
            for (int rep = 0; rep < 4; ++rep)
            {
                double const dx = proj1[2 * p] - proj2[2 * p];
                double const dy = proj1[2 * p + 1] - proj2[2 * p + 1];
                sum_diff += sqrt(dx * dx + dy * dy);
            }
        if (mean_delta != NULL)
            *mean_delta = sum_diff / (double)(4 * count);
    }
    else
    {
        uint8_t *new_active = (uint8_t *)calloc((size_t)num_nodes, 1);
        size_t p = 0;
        int const size = surf->patchsize;
        for (int patch_id = 0; patch_id < num_patches; ++patch_id)
        {
            if (!surf->patch_valid[patch_id])
                continue;
            int ids[4];
            patch_node_ids(surf, patch_id, ids);
            if ((active_nodes[ids[0]] + active_nodes[ids[1]]
                + active_nodes[ids[2]] + active_nodes[ids[3]]) == 0)
                continue;
            for (int j = 0; j < views->n_subs; ++j)
            {
                if (!(surf->patch_vis[patch_id] & (1u << j)))
                    continue;
                for (int i = 0; i < size * size; ++i, ++p)
                {
                    double const dx = proj1[2 * p] - proj2[2 * p];
                    double const dy = proj1[2 * p + 1] - proj2[2 * p + 1];
                    double const diff = sqrt(dx * dx + dy * dy);
                    if (diff > 0.15)
                        for (int n = 0; n < 4; ++n)
                            new_active[ids[n]] = 1;
                }
            }
        }
        memcpy(active_nodes, new_active, (size_t)num_nodes);
        free(new_active);
        result = 0;
        for (int i = 0; i < num_nodes; ++i)
            if (active_nodes[i] == 1)
                result += 1;
    }
    free(proj1);
    free(proj2);
    return result;
}

/* surface.cc:155-168 + surface_patch.cc:15-28 (float image, zero filled) */
void
orc_depth_map(const orc_surface *surf, float *depth)
{
    int const W = surf->width, H = surf->height;
    int const size = surf->patchsize;
    memset(depth, 0, sizeof(float) * (size_t)W * H);
    double *pixels = (double *)malloc(sizeof(double) * 2 * size * size);
    double *depths = (double *)malloc(sizeof(double) * size * size);
    for (int p = 0; p < surf->npx * surf->npy; ++p)
    {
        if (!surf->patch_valid[p])
            continue;
        double nodes16[16];
        patch_nodes(surf, p, nodes16);
        int px0, py0;
        patch_pixel_origin(surf, p, &px0, &py0);
        int const n = orc_patch_values_at_pixels(nodes16, px0, py0, size, 1,
            pixels, depths, NULL, NULL, NULL);
        for (int i = 0; i < n; ++i)
            depth[(size_t)pixels[2 * i + 1] * W + (size_t)pixels[2 * i]] =
                (float)depths[i];
    }
    free(pixels);
    free(depths);
}

/* surface.cc:170-183 + surface_patch.cc:30-55 */
void
orc_normal_map(const orc_surface *surf, float inv_flen, float *normals)
{
    int const W = surf->width, H = surf->height;
    int const size = surf->patchsize;
    memset(normals, 0, sizeof(float) * 3 * (size_t)W * H);
    double *pixels = (double *)malloc(sizeof(double) * 2 * size * size);
    double *depths = (double *)malloc(sizeof(double) * size * size);
    double *dd = (double *)malloc(sizeof(double) * 2 * size * size);
    for (int p = 0; p < surf->npx * surf->npy; ++p)
    {
        if (!surf->patch_valid[p])
            continue;
        double nodes16[16];
        patch_nodes(surf, p, nodes16);
        int px0, py0;
        patch_pixel_origin(surf, p, &px0, &py0);
        int const n = orc_patch_values_at_pixels(nodes16, px0, py0, size, 1,
            pixels, depths, dd, NULL, NULL);
        for (int i = 0; i < n; ++i)
        {
            double normal[3];
            double const x = pixels[2 * i] + 0.5 - (double)W / 2.0;
            double const y = pixels[2 * i + 1] + 0.5 - (double)H / 2.0;
            orc_fill_normal(x, y, (double)inv_flen, depths[i], dd[2 * i],
                dd[2 * i + 1], normal);
            size_t const o = ((size_t)pixels[2 * i + 1] * W
                + (size_t)pixels[2 * i]) * 3;
            for (int c = 0; c < 3; ++c)
                normals[o + c] = (float)normal[c];
        }
    }
    free(pixels);
    free(depths);
    free(dd);
}

/* light_optimizer.cc:32-49 */
void
orc_light_accumulate(const float *normals, const float *image,
    int num_pixels, double *A, double *b)
{
    memset(A, 0, sizeof(double) * 256);
    memset(b, 0, sizeof(double) * 16);
    for (int p = 0; p < num_pixels; ++p)
    {
        double normal[3] = { normals[3 * (size_t)p], normals[3 * (size_t)p + 1],
            normals[3 * (size_t)p + 2] };
        double const len = sqrt(normal[0] * normal[0] + normal[1] * normal[1]
            + normal[2] * normal[2]);
        if (fabs(len - 1.0) > 1e-6 || image[p] < 0.05f)
            continue;
        double sh[16];
        orc_sh_evaluate_4_band(normal, sh);
        for (int i = 0; i < 16; ++i)
        {
            b[i] += sh[i] * image[p];
            for (int j = 0; j < 16; ++j)
                A[j * 16 + i] += sh[i] * sh[j];
        }
    }
}

/* Pseudo inverse of the symmetric PSD normal matrix through a cyclic Jacobi
 * eigen-decomposition; eigenvalues below 1e-12 * max are treated as zero.
 * (math::matrix_pseudo_inverse is MVE code, not on disk: [MVE-unverified].) */
void
orc_light_solve(const double *A256, const double *b16, double *params)
{
    double A[256], V[256];
    memcpy(A, A256, sizeof(A));
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            V[i * 16 + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep)
    {
        double off = 0.0;
        for (int i = 0; i < 16; ++i)
            for (int j = i + 1; j < 16; ++j)
                off += A[i * 16 + j] * A[i * 16 + j];
        if (off < 1e-300)
            break;
        for (int p = 0; p < 16; ++p)
            for (int q = p + 1; q < 16; ++q)
            {
                double const apq = A[p * 16 + q];
                if (apq == 0.0)
                    continue;
                double const theta = (A[q * 16 + q] - A[p * 16 + p])
                    / (2.0 * apq);
                double const tt = (theta >= 0 ? 1.0 : -1.0)
                    / (fabs(theta) + sqrt(theta * theta + 1.0));
                double const c = 1.0 / sqrt(tt * tt + 1.0);
                double const s = tt * c;
                for (int k = 0; k < 16; ++k)
                {
                    double const akp = A[k * 16 + p], akq = A[k * 16 + q];
                    A[k * 16 + p] = c * akp - s * akq;
                    A[k * 16 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 16; ++k)
                {
                    double const apk = A[p * 16 + k], aqk = A[q * 16 + k];
                    A[p * 16 + k] = c * apk - s * aqk;
                    A[q * 16 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 16; ++k)
                {
                    double const vkp = V[k * 16 + p], vkq = V[k * 16 + q];
                    V[k * 16 + p] = c * vkp - s * vkq;
                    V[k * 16 + q] = s * vkp + c * vkq;
                }
            }
    }
    double maxev = 0.0;
    for (int i = 0; i < 16; ++i)
        if (fabs(A[i * 16 + i]) > maxev)
            maxev = fabs(A[i * 16 + i]);
    for (int i = 0; i < 16; ++i)
        params[i] = 0.0;
    for (int k = 0; k < 16; ++k)
    {
        double const ev = A[k * 16 + k];
        if (fabs(ev) <= 1e-12 * maxev || ev == 0.0)
            continue;
        double proj = 0.0;
        for (int i = 0; i < 16; ++i)
            proj += V[i * 16 + k] * b16[i];
        proj /= ev;
        for (int i = 0; i < 16; ++i)
            params[i] += V[i * 16 + k] * proj;
    }
}
