// Gauss-Newton normal equations on gfx950.
//
// Replaces GaussNewtonStep::construct (reference: lib/gauss_newton_step.cc:33-143),
// jacobian_entries_for_patch (:145-244) and fill_gradient_and_hessian_entries
// (:246-518).
//
// Formulation.  For one pixel every Jacobian row the reference builds is a
// linear combination of six rows of the bicubic basis table dn (the table
// of lib/bicubic_patch.cc:302-316 / lib/surface.cc:929-955):
//   photometric rows  jac_entries[j][col] = P_j * dn_w[col] + Q_j * dn_wx|wy[col]
//                     (correspondence.cc:74-86, 102-187, gauss_newton_step.cc:202-207),
//   regulariser rows  full_surface_div_deriv[v][col] = sum_a E[v][a] dn_a[col]
//                     (surface_derivative.cc:109-190),
//   shading rows      render_deriv[col] = sum_a rho[a] dn_a[col]
//                     (gauss_newton_step.cc:468-499).
// So the per-pixel contribution to the patch's 16x16 system is
//   D6^T M6 D6  and  D6^T v6      with D6 = dn rows (w, w_x, w_y, w_xy, w_xx, w_yy)
// and a 6x6 symmetric M6 / 6-vector v6 that carry all the view, IRLS-weight
// and lighting dependence.  Phase 1 computes (M6, v6) with one lane per
// sampled pixel (FP64 VALU); phase 2 contracts sum_pix D6^T (M6 D6) on the
// matrix cores, K = 6 rows per pixel: three v_mfma_f64_4x4x4 per row (four
// independent 4 x 4 node blocks each) cover the ten node blocks (bi <= bj) of
// the symmetric 16 x 16 patch system; a 16x16x4 would compute all sixteen.
// One wavefront owns 64 sampled pixels = 4 patches at the fine scales.
#include "common.h"

namespace smvs_hip {

#define R_FACTOR 1e-4  // gauss_newton_step.cc:17

typedef double double4_t __attribute__((ext_vector_type(4)));

struct PatchKernelArgs {
    const double *nodes;
    const uint8_t *patch_valid;
    const uint32_t *patch_vis;
    const uint8_t *active;
    const int *live_list;        // compacted ids of the patches to evaluate
    const double *hermite_tab;   // [ps][12]
    const float2 *main_grad;
    const float *main_shading;
    const float2 *main_shading_grad;
    const SubPlanes *subs;
    const DeviceCameras *cams;
    const double *lighting;      // [16] or nullptr
    double *Hp;
    double *gp;
    PatchLayout layout;          // of Hp / gp (common.h)
    int W, H;
    int npx, npy, stride;
    int ps, start_x, start_y;
    int sampling, spr, P;        // samples per row, samples per patch
    int n_subs;
    int num_patches;
    double flen, inv_flen;
    double inv_f, f2inv;         // 1 / flen, 1 / flen^2 (double)
    double reg, light_reg;
    int use_lighting;
    int *status;
    int check_stop;   // leave at once when status[I_STOP] / [I_STEP_ABORT] is set
};

// 1 / x to ~1 ulp: hardware estimate (2^-26 or better) + two Newton steps.
__device__ __forceinline__ double
fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// IRLS weight 1 / x: one Newton step (relative error <= 2^-46), the weights
// only scale residual rows.
__device__ __forceinline__ double
weight_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

// a / d, correctly rounded for normal-range operands: quotient from the
// reciprocal plus one residual correction (the core of the IEEE sequence
// without the scaling steps).
__device__ __forceinline__ double
div_corrected(double a, double d, double inv_d)
{
    double const q = a * inv_d;
    double const r = __builtin_fma(-q, d, a);
    return __builtin_fma(r, inv_d, q);
}

// mve::Image<float>::linear_at in the reference's float operation order
// (no FMA contraction) on the packed device planes. [MVE-unverified]
struct Taps {
    unsigned o00, o10, o01, o11;
    float w00, w10, w01, w11;
};

__device__ __forceinline__ Taps
make_taps(float x, float y, int w, int h)
{
#pragma clang fp contract(off)
    Taps t;
    x = fmaxf(0.0f, fminf((float)(w - 1), x));
    y = fmaxf(0.0f, fminf((float)(h - 1), y));
    int const fx = (int)x;
    int const fy = (int)y;
    int const fx1 = min(fx + 1, w - 1);
    int const fy1 = min(fy + 1, h - 1);
    float const w1 = x - (float)fx;
    float const w0 = 1.0f - w1;
    float const w3 = y - (float)fy;
    float const w2 = 1.0f - w3;
    t.o00 = (unsigned)(fy * w + fx);
    t.o10 = (unsigned)(fy * w + fx1);
    t.o01 = (unsigned)(fy1 * w + fx);
    t.o11 = (unsigned)(fy1 * w + fx1);
    t.w00 = w0 * w2;
    t.w10 = w1 * w2;
    t.w01 = w0 * w3;
    t.w11 = w1 * w3;
    return t;
}

__device__ __forceinline__ float
tap_mix(float v1, float v2, float v3, float v4, Taps const &t)
{
#pragma clang fp contract(off)
    return v1 * t.w00 + v2 * t.w10 + v3 * t.w01 + v4 * t.w11;
}

// upper-triangular packed index of a symmetric 6x6
__device__ __forceinline__ constexpr int
sym6(int a, int b)
{
    return a <= b ? a * 6 - (a * (a - 1)) / 2 + (b - a)
                  : b * 6 - (b * (b - 1)) / 2 + (a - b);
}

// Per-pixel quantities of the normal divergence (surface_derivative.cc:69-107)
// shared by its value and its derivatives.
struct DivState {
    double x, y, f, f2inv;
    double w, wx, wy, wxy, wxx, wyy;
    double a, ax, ay, t, n, b, c, nx, ny;
    double inv_n, inv_t, inv_t2, inv_t2f, inv_f;
};

// The 6 x 6 Jacobian E[v][a] = d div[v] / d (w, w_x, w_y, w_xy, w_xx, w_yy):
// surface_derivative.cc:128-188 with the basis-table row replaced by the six
// unit directions, each written out.  (A generic directional derivative
// called with unit vectors costs 1.8x the instructions: strict IEEE
// arithmetic cannot drop a product with a literal zero, 0 * x being NaN for
// an infinite x.)  The x-divergences (v = 0, 1, 2) do not
// depend on w_yy, the y-divergences (v = 3, 4, 5) not on w_xx.
__device__ __forceinline__ constexpr bool
div_depends_on(int v, int a)
{
    return v < 3 ? a != 5 : a != 4;
}

__device__ __forceinline__ void
div_jacobian(DivState const &s, double E[6][6])
{
    double const n = s.n, t = s.t, F = s.f2inv;
    double const it = s.inv_t, it2 = s.inv_t2, it2f = s.inv_t2f;
    // the numerators of the six divergence entries
    double const Kxx = s.wxx * n - s.wx * s.nx, Kyx = s.wxy * n - s.wy * s.nx;
    double const Kzx = s.ax * n - s.a * s.nx, Kxy = s.wxy * n - s.wx * s.ny;
    double const Kyy = s.wyy * n - s.wy * s.ny, Kzy = s.ay * n - s.a * s.ny;
    double const Fa = F * s.a;

    // ---- along w: ap = 1 ----
    {
        double const t2p = Fa;
        double const np = t2p * s.inv_n;
        double const bp = F * s.ax, cp = F * s.ay;
        double const nxp = (bp * n - s.b * np) * it;
        double const nyp = (cp * n - s.c * np) * it;
        double const tt = 2.0 * t2p;
        E[0][0] = ((s.wxx * np - s.wx * nxp) * t - Kxx * tt) * it2;
        E[1][0] = -(((s.wxy * np - s.wy * nxp) * t - Kyx * tt) * it2);
        E[2][0] = ((s.ax * np - s.nx - s.a * nxp) * t - Kzx * tt) * it2f;
        E[3][0] = ((s.wxy * np - s.wx * nyp) * t - Kxy * tt) * it2;
        E[4][0] = -(((s.wyy * np - s.wy * nyp) * t - Kyy * tt) * it2);
        E[5][0] = ((s.ay * np - s.ny - s.a * nyp) * t - Kzy * tt) * it2f;
    }
    // ---- along w_x: ap = x, axp = 2 ----
    {
        double const t2p = s.wx + Fa * s.x;
        double const np = t2p * s.inv_n;
        double const bp = s.wxx + F * (s.x * s.ax + s.a * 2.0);
        double const cp = s.wxy + F * (s.x * s.ay);
        double const nxp = (bp * n - s.b * np) * it;
        double const nyp = (cp * n - s.c * np) * it;
        double const tt = 2.0 * t2p;
        E[0][1] = ((s.wxx * np - s.nx - s.wx * nxp) * t - Kxx * tt) * it2;
        E[1][1] = -(((s.wxy * np - s.wy * nxp) * t - Kyx * tt) * it2);
        E[2][1] = ((2.0 * n + s.ax * np - s.x * s.nx - s.a * nxp) * t - Kzx * tt) * it2f;
        E[3][1] = ((s.wxy * np - s.ny - s.wx * nyp) * t - Kxy * tt) * it2;
        E[4][1] = -(((s.wyy * np - s.wy * nyp) * t - Kyy * tt) * it2);
        E[5][1] = ((s.ay * np - s.x * s.ny - s.a * nyp) * t - Kzy * tt) * it2f;
    }
    // ---- along w_y: ap = y, ayp = 2 ----
    {
        double const t2p = s.wy + Fa * s.y;
        double const np = t2p * s.inv_n;
        double const bp = s.wxy + F * (s.y * s.ax);
        double const cp = s.wyy + F * (s.y * s.ay + s.a * 2.0);
        double const nxp = (bp * n - s.b * np) * it;
        double const nyp = (cp * n - s.c * np) * it;
        double const tt = 2.0 * t2p;
        E[0][2] = ((s.wxx * np - s.wx * nxp) * t - Kxx * tt) * it2;
        E[1][2] = -(((s.wxy * np - s.nx - s.wy * nxp) * t - Kyx * tt) * it2);
        E[2][2] = ((s.ax * np - s.y * s.nx - s.a * nxp) * t - Kzx * tt) * it2f;
        E[3][2] = ((s.wxy * np - s.wx * nyp) * t - Kxy * tt) * it2;
        E[4][2] = -(((s.wyy * np - s.ny - s.wy * nyp) * t - Kyy * tt) * it2);
        E[5][2] = ((2.0 * n + s.ay * np - s.y * s.ny - s.a * nyp) * t - Kzy * tt) * it2f;
    }
    // ---- along w_xy: axp = y, ayp = x; the length of the normal does not move ----
    {
        double const bp = s.wy + Fa * s.y;
        double const cp = s.wx + Fa * s.x;
        double const nxp = (bp * n) * it;
        double const nyp = (cp * n) * it;
        E[0][3] = ((-s.wx * nxp) * t) * it2;
        E[1][3] = -(((n - s.wy * nxp) * t) * it2);
        E[2][3] = ((s.y * n - s.a * nxp) * t) * it2f;
        E[3][3] = ((n - s.wx * nyp) * t) * it2;
        E[4][3] = -(((-s.wy * nyp) * t) * it2);
        E[5][3] = ((s.x * n - s.a * nyp) * t) * it2f;
    }
    // ---- along w_xx: axp = x ----
    {
        double const bp = s.wx + Fa * s.x;
        double const nxp = (bp * n) * it;
        E[0][4] = ((n - s.wx * nxp) * t) * it2;
        E[1][4] = -(((-s.wy * nxp) * t) * it2);
        E[2][4] = ((s.x * n - s.a * nxp) * t) * it2f;
        E[3][4] = 0.0;
        E[4][4] = 0.0;
        E[5][4] = 0.0;
    }
    // ---- along w_yy: ayp = y ----
    {
        double const cp = s.wy + Fa * s.y;
        double const nyp = (cp * n) * it;
        E[0][5] = 0.0;
        E[1][5] = 0.0;
        E[2][5] = 0.0;
        E[3][5] = ((-s.wx * nyp) * t) * it2;
        E[4][5] = -(((n - s.wy * nyp) * t) * it2);
        E[5][5] = ((s.y * n - s.a * nyp) * t) * it2f;
    }
}

// surface_derivative.cc:42-63 along p (only w, w_x, w_y matter)
__device__ __forceinline__ void
normal_along(DivState const &s, double wp, double dxp, double dyp,
    double out[3])
{
    double const ap = wp + s.x * dxp + s.y * dyp;
    double const t2p = s.wx * dxp + s.wy * dyp + s.f2inv * s.a * ap;
    double const np = t2p * s.inv_n;
    out[0] = (dxp * s.n - s.wx * np) * s.inv_t;
    out[1] = (-dyp * s.n + s.wy * np) * s.inv_t;
    out[2] = (ap * s.n - s.a * np) * s.inv_t * s.inv_f;
}

// spherical_harmonics.h:53-73,133-151 and :79-127,157-201 contracted with the
// lighting parameters: shading = l . sh(n), G = sum_{l>=1} l_l dsh_l/dn.
__device__ __forceinline__ void
shading_and_gradient(const double *lp, const double n[3], double *shading,
    double G[3])
{
    double const nx = n[0], ny = n[1], nz = n[2];
    double const x2 = nx * nx, y2 = ny * ny, z2 = nz * nz;
    double sh[16];
    sh[0] = 1.0; sh[1] = ny; sh[2] = nz; sh[3] = nx;
    sh[4] = nx * ny; sh[5] = ny * nz; sh[6] = -x2 - y2 + 2.0 * z2;
    sh[7] = nx * nz; sh[8] = x2 - y2;
    sh[9] = (3.0 * x2 - y2) * ny; sh[10] = nx * ny * nz;
    sh[11] = (4.0 * z2 - x2 - y2) * ny;
    sh[12] = (2.0 * z2 - 3.0 * x2 - 3.0 * y2) * nz;
    sh[13] = (4.0 * z2 - x2 - y2) * nx;
    sh[14] = (x2 - y2) * nz; sh[15] = (x2 - 3.0 * y2) * nx;
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l)
        s += lp[l] * sh[l];
    *shading = s;

    double d[16][3];
    d[0][0] = 0; d[0][1] = 0; d[0][2] = 0;
    d[1][0] = 0; d[1][1] = 1; d[1][2] = 0;
    d[2][0] = 0; d[2][1] = 0; d[2][2] = 1;
    d[3][0] = 1; d[3][1] = 0; d[3][2] = 0;
    d[4][0] = ny; d[4][1] = nx; d[4][2] = 0;
    d[5][0] = 0; d[5][1] = nz; d[5][2] = ny;
    d[6][0] = -2.0 * nx; d[6][1] = -2.0 * ny; d[6][2] = 4.0 * nz;
    d[7][0] = nz; d[7][1] = 0; d[7][2] = nx;
    d[8][0] = 2.0 * nx; d[8][1] = -2.0 * ny; d[8][2] = 0;
    d[9][0] = 6.0 * nx * ny; d[9][1] = 3.0 * (x2 - y2); d[9][2] = 0;
    d[10][0] = ny * nz; d[10][1] = nx * nz; d[10][2] = nx * ny;
    d[11][0] = -2.0 * nx * ny; d[11][1] = 4.0 * z2 - x2 - 3.0 * y2;
    d[11][2] = 8.0 * ny * nz;
    d[12][0] = -6.0 * nx * nz; d[12][1] = -6.0 * ny * nz;
    d[12][2] = 6.0 * z2 - 3.0 * (x2 + y2);
    d[13][0] = 4.0 * z2 - 3.0 * x2 - y2; d[13][1] = -2.0 * nx * ny;
    d[13][2] = 8.0 * nx * nz;
    d[14][0] = 2.0 * nx * nz; d[14][1] = -2.0 * ny * nz; d[14][2] = x2 - y2;
    d[15][0] = 3.0 * (x2 - y2); d[15][1] = -6.0 * nx * ny; d[15][2] = 0;
    G[0] = G[1] = G[2] = 0.0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {  // sh0 is constant
        G[0] += lp[l] * d[l][0];
        G[1] += lp[l] * d[l][1];
        G[2] += lp[l] * d[l][2];
    }
}

// Column `col` (= 4*node + param, node = 2*b + a) of the basis table is the
// product of the x-function ex and the y-function ey of the Hermite table.
__device__ __forceinline__ void
col_functions(int col, int *ex, int *ey)
{
    int const n = col >> 2, i = col & 3;
    int const a = n & 1, b = n >> 1;
    *ex = (i & 1) ? 2 + a : a;
    *ey = (i & 2) ? 2 + b : b;
}

// ---------------------------------------------------------------------------
// Phase 1: one lane = one sampled pixel -> (M6[21], v6[6])
// ---------------------------------------------------------------------------
// `nb` points at this lane's column of the per-neighbour scratch in LDS:
// value k of neighbour slot c lives at nb[(c * 5 + k) * 64].
__device__ __forceinline__ void
pixel_system(PatchKernelArgs const &A, const double *tabs /*LDS [spr][12]*/,
    double *nb, double const theta[16], int px, int py, int sx, int sy,
    uint32_t vis, double M6[21], double v6[6])
{
#pragma unroll
    for (int i = 0; i < 21; ++i)
        M6[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        v6[i] = 0.0;

    // ---- bicubic surface values at the pixel (surface_patch.cc:95-109) ----
    double w, wx, wy, wxy, wxx, wyy;
    {
        const double *X = tabs + sx * 12;
        const double *Y = tabs + sy * 12;
        double G[4][3];
#pragma unroll
        for (int ey = 0; ey < 4; ++ey) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                G[ey][k] = 0.0;
#pragma unroll
            for (int ex = 0; ex < 4; ++ex) {
                int const a = ex & 1, ix = ex >> 1, b = ey & 1, iy = ey >> 1;
                double const c = theta[4 * (2 * b + a) + ix + 2 * iy];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    G[ey][k] = __builtin_fma(c, X[ex * 3 + k], G[ey][k]);
            }
        }
        w = wx = wy = wxy = wxx = wyy = 0.0;
#pragma unroll
        for (int ey = 0; ey < 4; ++ey) {
            double const y0 = Y[ey * 3 + 0], y1 = Y[ey * 3 + 1],
                y2 = Y[ey * 3 + 2];
            w = __builtin_fma(G[ey][0], y0, w);
            wx = __builtin_fma(G[ey][1], y0, wx);
            wy = __builtin_fma(G[ey][0], y1, wy);
            wxy = __builtin_fma(G[ey][1], y1, wxy);
            wxx = __builtin_fma(G[ey][2], y0, wxx);
            wyy = __builtin_fma(G[ey][0], y2, wyy);
        }
    }

    float2 const gm = A.main_grad[(size_t)py * A.W + px];
    double const gm0 = gm.x, gm1 = gm.y;

    // ---- neighbours (gauss_newton_step.cc:175-208) ----
    // Neighbour loop, software pipelined: the taps of neighbour j are issued,
    // then the IRLS terms of the PREVIOUS neighbour (reference term and its
    // pairs with all earlier neighbours, read from LDS) run while the loads
    // are in flight, then neighbour j's row coefficients are formed.
    double m_ww = 0.0, m_wx = 0.0, m_xx = 0.0, m_wy = 0.0, m_yy = 0.0;
    double v_w = 0.0, v_x = 0.0, v_y = 0.0;
    int num_subs = 0;            // neighbours folded into the sums so far
    bool have_prev = false;
    double s0p = 0.0, s1p = 0.0, P0p = 0.0, P1p = 0.0, Qp = 0.0;

    // reference term of the pending neighbour and its pairs with slots
    // 0 .. num_subs-1, then the pending values go to slot num_subs
    auto fold_pending = [&](bool keep) {
        {
            double const diff0 = s0p - gm0, diff1 = s1p - gm1;
            double const w0 = weight_rcp(fabs(diff0) + R_FACTOR);
            double const w1 = weight_rcp(fabs(diff1) + R_FACTOR);
            double const a0 = w0 * P0p, b0 = w0 * Qp;
            double const a1 = w1 * P1p, b1 = w1 * Qp;
            m_ww += a0 * P0p + a1 * P1p;
            m_wx += a0 * Qp;
            m_xx += b0 * Qp;
            m_wy += a1 * Qp;
            m_yy += b1 * Qp;
            v_w += a0 * diff0 + a1 * diff1;
            v_x += b0 * diff0;
            v_y += b1 * diff1;
        }
#pragma unroll 1
        for (int j2 = 0; j2 < num_subs; ++j2) {
            const double *sk = nb + (j2 * 5) * 64;
            double const sd0 = sk[0] - s0p, sd1 = sk[64] - s1p;
            double const w0 = weight_rcp(fabs(sd0) + R_FACTOR);
            double const w1 = weight_rcp(fabs(sd1) + R_FACTOR);
            double const dP0 = sk[128] - P0p, dP1 = sk[192] - P1p;
            double const dQ = sk[256] - Qp;
            double const a0 = w0 * dP0, b0 = w0 * dQ;
            double const a1 = w1 * dP1, b1 = w1 * dQ;
            m_ww += a0 * dP0 + a1 * dP1;
            m_wx += a0 * dQ;
            m_xx += b0 * dQ;
            m_wy += a1 * dQ;
            m_yy += b1 * dQ;
            v_w += a0 * sd0 + a1 * sd1;
            v_x += b0 * sd0;
            v_y += b1 * sd1;
        }
        if (keep) {
            double *slot = nb + (num_subs * 5) * 64;
            slot[0 * 64] = s0p;
            slot[1 * 64] = s1p;
            slot[2 * 64] = P0p;
            slot[3 * 64] = P1p;
            slot[4 * 64] = Qp;
        }
        num_subs += 1;
    };

#pragma unroll 1
    for (int j = 0; j < A.n_subs; ++j) {
        if (!((vis >> j) & 1u))
            continue;
        // (copied up front: after the LDS stores below the compiler could
        // no longer prove the cameras unmodified and would re-read them with
        // vector loads)
        double M[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            M[i] = A.cams->M[j][i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            t[i] = A.cams->t[j][i];
        SubPlanes const sp = A.subs[j];
        double p, q, r, a, b, d, proj0, proj1, inv_d;
        {
            // correspondence.cc:36-51 in the reference's operation order (no
            // contraction, correctly rounded quotients): the projection feeds
            // the float tap coordinates.
#pragma clang fp contract(off)
            double const u = (double)px + 0.5, v = (double)py + 0.5;
            p = M[0] * u + M[1] * v + M[2];
            q = M[3] * u + M[4] * v + M[5];
            r = M[6] * u + M[7] * v + M[8];
            a = w * p + t[0];
            b = w * q + t[1];
            d = w * r + t[2];
            inv_d = fast_rcp(d);
            proj0 = div_corrected(a, d, inv_d);
            proj1 = div_corrected(b, d, inv_d);
            proj0 -= 0.5;
            proj1 -= 0.5;
        }
        Taps const tp = make_taps((float)proj0, (float)proj1, sp.width,
            sp.height);
        // (pointers read from memory are generic to the compiler; the
        // planes live in global memory: global_load instead of flat_load)
        typedef float fvec2 __attribute__((ext_vector_type(2)));
        typedef float fvec4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) fvec2 *gf2_ptr;
        typedef const __attribute__((address_space(1))) fvec4 *gf4_ptr;
        gf2_ptr const grad = (gf2_ptr)sp.grad;
        gf4_ptr const hess = (gf4_ptr)sp.hess;
        fvec2 const g00 = grad[tp.o00], g10 = grad[tp.o10],
            g01 = grad[tp.o01], g11 = grad[tp.o11];
        fvec4 const h00 = hess[tp.o00], h10 = hess[tp.o10],
            h01 = hess[tp.o01], h11 = hess[tp.o11];

        // ---- the previous neighbour's IRLS terms hide the tap latency ----
        if (have_prev)
            fold_pending(true);

        // ... and so does everything of this neighbour that needs no texel:
        // correspondence.cc:88-100 with reciprocals
        double const inv_d2 = inv_d * inv_d;
        double const rx = wx * r + w * M[6], ry = wy * r + w * M[7];
        double const jac0 = (wx * p + w * M[0]) * inv_d - a * rx * inv_d2;
        double const jac2 = (wy * p + w * M[1]) * inv_d - a * ry * inv_d2;
        double const jac1 = (wx * q + w * M[3]) * inv_d - b * rx * inv_d2;
        double const jac3 = (wy * q + w * M[4]) * inv_d - b * ry * inv_d2;
        double const du_w = (p * d - r * a) * inv_d2;
        double const dv_w = (q * d - r * b) * inv_d2;
        // correspondence.cc:102-168
        double const d_prime_d4 = 2.0 * d * r * inv_d2 * inv_d2;
        double const du_cp = p * t[2] - r * t[0];
        double const dv_cp = q * t[2] - r * t[1];
        double du_A[2], dv_A[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double const pp = M[k], qp = M[3 + k], rp = M[6 + k];
            double const wk = k == 0 ? wx : wy;
            double const du_at = w * (pp * r - p * rp);
            double const dv_at = w * (qp * r - q * rp);
            double const du_bp = pp * t[2] - rp * t[0];
            double const dv_bp = qp * t[2] - rp * t[1];
            double const du_abc = w * (du_at + du_bp) + wk * du_cp;
            double const dv_abc = w * (dv_at + dv_bp) + wk * dv_cp;
            du_A[k] = (2.0 * du_at + du_bp) * inv_d2 - du_abc * d_prime_d4;
            dv_A[k] = (2.0 * dv_at + dv_bp) * inv_d2 - dv_abc * d_prime_d4;
        }
        double const cu = du_cp * inv_d2;
        double const cv = dv_cp * inv_d2;

        // The texels are first touched HERE: the empty asm takes them as
        // operands, so the compiler cannot hoist their conversions (and the
        // wait that goes with them) above the pair loop.
        fvec2 c00 = g00, c10 = g10, c01 = g01, c11 = g11;
        fvec4 k00 = h00, k10 = h10, k01 = h01, k11 = h11;
        // (not volatile -- that would count as a memory access and demote
        // the uniform camera loads to vector loads; the dependence on m_ww,
        // which the pair loop produces, is what keeps it below that loop)
        asm("" : "+v"(c00), "+v"(c10), "+v"(c01), "+v"(c11),
            "+v"(k00), "+v"(k10), "+v"(k01), "+v"(k11) : "v"(m_ww));
        double const g0 = tap_mix(c00.x, c10.x, c01.x, c11.x, tp);
        double const g1 = tap_mix(c00.y, c10.y, c01.y, c11.y, tp);
        double const hxx = tap_mix(k00.x, k10.x, k01.x, k11.x, tp);
        double const hxy = tap_mix(k00.y, k10.y, k01.y, k11.y, tp);
        double const hyy = tap_mix(k00.z, k10.z, k01.z, k11.z, tp);

        double const JH0 = jac0 * hxx + jac1 * hxy;
        double const JH1 = jac0 * hxy + jac1 * hyy;
        double const JH2 = jac2 * hxx + jac3 * hxy;
        double const JH3 = jac2 * hxy + jac3 * hyy;
        s0p = jac0 * g0 + jac1 * g1;
        s1p = jac2 * g0 + jac3 * g1;
        P0p = du_A[0] * g0 + dv_A[0] * g1 + JH0 * du_w + JH1 * dv_w;
        P1p = du_A[1] * g0 + dv_A[1] * g1 + JH2 * du_w + JH3 * dv_w;
        Qp = cu * g0 + cv * g1;
        have_prev = true;
    }
    if (have_prev)
        fold_pending(false);   // the last neighbour needs no LDS slot

    M6[sym6(0, 0)] = m_ww;
    M6[sym6(0, 1)] = m_wx;
    M6[sym6(1, 1)] = m_xx;
    M6[sym6(0, 2)] = m_wy;
    M6[sym6(2, 2)] = m_yy;
    v6[0] = v_w;
    v6[1] = v_x;
    v6[2] = v_y;

    if (!(A.reg > 0.0))
        return;

    // ---- regulariser + shading (gauss_newton_step.cc:210-240, 385-517) ----
    double const num_diffs = (double)((num_subs * (num_subs + 1)) / 2);
    double const brw = A.reg * 0.005 * fast_rcp(fmax(0.03, fabs(gm0) + fabs(gm1)))
        * num_diffs;

    DivState s;
    s.x = (double)px + 0.5 - (double)A.W / 2.0;
    s.y = (double)py + 0.5 - (double)A.H / 2.0;
    s.f = A.flen;
    s.f2inv = A.f2inv;
    s.w = w; s.wx = wx; s.wy = wy; s.wxy = wxy; s.wxx = wxx; s.wyy = wyy;
    s.a = w + s.x * wx + s.y * wy;
    s.ax = 2.0 * wx + s.x * wxx + s.y * wxy;
    s.ay = 2.0 * wy + s.y * wyy + s.x * wxy;
    double const a_f2 = s.a * s.f2inv;
    s.t = wx * wx + wy * wy + s.a * a_f2;
    s.n = sqrt(s.t);
    s.b = wx * wxx + wy * wxy + a_f2 * s.ax;
    s.c = wx * wxy + wy * wyy + a_f2 * s.ay;
    s.inv_n = fast_rcp(s.n);
    s.inv_t = s.inv_n * s.inv_n;
    s.inv_t2 = s.inv_t * s.inv_t;
    s.inv_t2f = s.inv_t2 * A.inv_f;
    s.inv_f = A.inv_f;
    s.nx = s.b * s.inv_n;
    s.ny = s.c * s.inv_n;

    // normal divergence (surface_derivative.cc:69-107)
    double div[6];
    div[0] = (wxx * s.n - wx * s.nx) * s.inv_t;               // xx
    div[1] = -((wxy * s.n - wy * s.nx) * s.inv_t);            // -yx
    div[2] = (s.ax * s.n - s.a * s.nx) * s.inv_t * s.inv_f;   // zx
    div[3] = (wxy * s.n - wx * s.ny) * s.inv_t;               // xy
    div[4] = -((wyy * s.n - wy * s.ny) * s.inv_t);            // -yy
    div[5] = (s.ay * s.n - s.a * s.ny) * s.inv_t * s.inv_f;   // zy

    // E[v][a] = d div[v] / d (surface quantity a)
    double E[6][6];
    div_jacobian(s, E);

    bool const lit = A.use_lighting != 0;
    if (!lit || A.light_reg > 0.0) {
        double geom_weight = 1.0;
        if (lit)
            geom_weight *= A.light_reg / 100;
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            double const wgt = geom_weight * brw
                * fast_rcp(R_FACTOR + fabs(div[v]));
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                if (!div_depends_on(v, a))
                    continue;
                double const wa = wgt * E[v][a];
                v6[a] += wa * div[v];
#pragma unroll
                for (int b = a; b < 6; ++b)
                    if (div_depends_on(v, b))
                        M6[sym6(a, b)] += wa * E[v][b];
            }
        }
    }
    if (!lit)
        return;

    // shading term (gauss_newton_step.cc:420-517)
    double normal[3];
    {
        double nz = (s.x * wx + s.y * wy + w) * A.inv_flen;
        double const len = sqrt(wx * wx + wy * wy + nz * nz);
        normal[0] = wx / len;
        normal[1] = -wy / len;
        normal[2] = nz / len;
    }
    double shading, G[3];
    shading_and_gradient(A.lighting, normal, &shading, G);
    float2 const lgf = A.main_shading_grad[(size_t)py * A.W + px];
    double lig0 = lgf.x, lig1 = lgf.y;
    double const liv = A.main_shading[(size_t)py * A.W + px];
    double const shading_weight = 0.001 * num_diffs
        / (R_FACTOR + (fabs(lig0) + fabs(lig1)));
    if (sqrt(lig0 * lig0 + lig1 * lig1) < 1e-10)
        return;
    if (shading * shading < 1e-10 || liv * liv < 1e-10)
        return;

    double sg[2];
    sg[0] = G[0] * div[0] + G[1] * div[1] + G[2] * div[2];
    sg[1] = G[0] * div[3] + G[1] * div[4] + G[2] * div[5];
    double const inv_sh = 1.0 / shading;
    double const inv_liv = 1.0 / liv;
    double err[2];
    err[0] = sg[0] * inv_sh - lig0 * inv_liv;
    err[1] = sg[1] * inv_sh - lig1 * inv_liv;

    // N[k][a]: d normal[k] / d (w, w_x, w_y)
    double Nn[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double out[3];
        normal_along(s, a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0,
            a == 2 ? 1.0 : 0.0, out);
        Nn[0][a] = out[0]; Nn[1][a] = out[1]; Nn[2][a] = out[2];
    }
    double const inv_sh2 = inv_sh * inv_sh;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        double rho[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double const sgd = G[0] * E[3 * c + 0][a] + G[1] * E[3 * c + 1][a]
                + G[2] * E[3 * c + 2][a];
            double const sd = a < 3
                ? G[0] * Nn[0][a] + G[1] * Nn[1][a] + G[2] * Nn[2][a] : 0.0;
            rho[a] = (sgd * shading - sg[c] * sd) * inv_sh2;
        }
        double const wgt = shading_weight / (R_FACTOR + fabs(err[c]));
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double const wa = wgt * rho[a];
            v6[a] += wa * err[c];
#pragma unroll
            for (int b = a; b < 6; ++b)
                M6[sym6(a, b)] += wa * rho[b];
        }
    }
}

// ---------------------------------------------------------------------------
// The patch kernel: one wavefront per block -- or, WAVES > 1, one wavefront per
// CHUNK of a patch.
//   PPW patches per wave (4 when a patch has <= 16 sampled pixels, else 1),
//   SLOTS = 64 / PPW pixel slots per patch and chunk.
//   WAVES (round 5, PPW == 1 only): a patch of more than 64 samples (scale 6:
//   16 x 16) was four chunks walked by ONE wave one after the other, and a
//   coarse step has a few hundred patches: 375 waves on 1,024 SIMDs, each
//   running four dependent chains of ~20 us (projection -> taps -> IRLS, eight
//   neighbours in a row).  With WAVES = 4 every chunk has its own wave (its own
//   LDS scratch) for phase 1; phase 2 -- the accumulation on the matrix cores,
//   a few microseconds per chunk -- is taken in turns in chunk order with the
//   accumulators handed on through LDS, so the patch system is bit-identical
//   with the one-wave form; the last wave stores it.
// ---------------------------------------------------------------------------
//   ONE_CHUNK (PPW == 1, WAVES == 1): the patch has at most 64 samples (scales 4
//   and 5): no loop over chunks, so the accumulators are only live in phase 2
//   (the generic form carries them around the loop: 256 VGPRs and scratch).
template <int PPW, int WAVES, bool ONE_CHUNK>
__global__ void __launch_bounds__(64 * WAVES, 2)
gn_patch_kernel(PatchKernelArgs A)
{
    static_assert(WAVES == 1 || PPW == 1, "chunks are split over waves for whole-wave patches only");
    static_assert(!ONE_CHUNK || (PPW == 1 && WAVES == 1), "ONE_CHUNK is a variant of <1, 1>");
    constexpr int SLOTS = 64 / PPW;
    extern __shared__ double lds[];
    int const wave = WAVES == 1 ? 0 : (int)(threadIdx.x >> 6);
    // [27][64] pixel systems, aliased with the [5 * n_subs][64] neighbour
    // scratch of phase 1 (dead once M6 is formed); one region per wave
    int const scratch_rows = max(27, 5 * (A.n_subs - 1));
    double *Msh = lds + (size_t)wave * scratch_rows * 64;
    double *tabs = lds + (size_t)WAVES * scratch_rows * 64;  // [spr][12] sampled coordinates
    double *hand = tabs + A.spr * 12;     // WAVES > 1: [4][64] accumulators from wave to wave

    int const lane = WAVES == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    // a pipelined Newton loop enqueues steps before it knows whether the loop
    // goes on (update.hip): once it has ended they do nothing
    if (A.check_stop && (A.status[I_STOP] | A.status[I_STEP_ABORT]) != 0)
        return;
    // Work = the compacted list of live patches (a patch with an active node,
    // gauss_newton_step.cc:73-79): a sparse step launches / occupies waves in
    // proportion to its active set.  The list's blocks are dealt to the XCDs
    // in contiguous bands (neighbouring patches share texels: one image
    // region per L2); blocks beyond the list leave at once.
    int const live_count = A.status[I_LIVE_PATCHES];
    unsigned const live_blocks = (unsigned)((live_count + PPW - 1) / PPW);
    unsigned const band = (live_blocks + 7u) >> 3;
    if (band > (gridDim.x >> 3)) {
        // the pipelined loop sized this launch before the list existed and
        // the list came out longer: the whole step is abandoned (every
        // workgroup takes this branch) and the host enqueues it again
        if (blockIdx.x == 0 && threadIdx.x == 0)
            A.status[I_STEP_ABORT] = ABORT_GRID;
        return;
    }
    unsigned const wv = (blockIdx.x & 7u) * band + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= band || wv >= live_blocks)
        return;
    int const slot_base = (int)wv * PPW;

    for (int i = (int)threadIdx.x; i < A.spr * 12; i += 64 * WAVES) {
        int const row = i / 12, e = i - row * 12;
        tabs[i] = A.hermite_tab[(size_t)(row * A.sampling) * 12 + e];
    }

    // the patch this lane works for in phase 1
    int const q1 = lane / SLOTS;
    int const sidx = lane - q1 * SLOTS;
    bool const live = slot_base + q1 < live_count;
    int const patch1 = live ? A.live_list[slot_base + q1] : 0;
    double theta[16];
    uint32_t vis = 0;
    int pox = 0, poy = 0;
    if (live) {
        int const ix = patch1 % A.npx, iy = patch1 / A.npx;
        int const n00 = iy * A.stride + ix;
        int const ids[4] = { n00, n00 + 1, n00 + A.stride, n00 + A.stride + 1 };
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const double *src = A.nodes + 4 * (size_t)ids[n];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                theta[4 * n + k] = src[k];
        }
        vis = A.patch_vis[patch1];
        pox = A.start_x + ix * A.ps;
        poy = A.start_y + iy * A.ps;
    }

    // phase 2 roles
    int const kg = lane >> 4, col = lane & 15;
    // Only the 10 node blocks (bi <= bj) of the symmetric 16 x 16 system are
    // needed: three v_mfma_f64_4x4x4 (four independent 4 x 4 blocks each)
    // instead of one 16x16x4 (all 16 blocks).  Lane (kg, col) feeds pixel kg,
    // column col; in block slot b = col >> 2 the B operand T = M6 D6 is this
    // lane's own column in all three instructions, the A operand is D6 of the
    // column rotated left by 0, 4, 8 within the 16:
    //   variant 0: blocks (b, b);  1: ((b - 1) & 3, b), slot 0 holding (3, 0) =
    //   (0, 3) transposed;  2: ((b - 2) & 3, b), slots 2 and 3 = (0, 2), (1, 3).
    int ex[3], ey[3];
#pragma unroll
    for (int v = 0; v < 3; ++v)
        col_functions((col - 4 * v) & 15, &ex[v], &ey[v]);

    double acc[PPW][3];
    double gacc[PPW];
    // PPW == 4 means P <= 16 = SLOTS: a single chunk, so the accumulators are
    // only live in phase 2
    int const chunks = PPW == 4 || ONE_CHUNK ? 1 : (A.P + SLOTS - 1) / SLOTS;
    // (WAVES > 1: wave w runs chunk w -- the host launches chunks == WAVES --
    // and every wave passes the barriers below exactly once)
    int const c_first = WAVES == 1 ? 0 : wave;
    int const c_last = WAVES == 1 ? chunks : wave + 1;
    for (int c = c_first; c < c_last; ++c) {
        __syncthreads();
        // ---- phase 1 ----
        {
            double M6[21], v6[6];
            int const si = c * SLOTS + sidx;
            if (live && si < A.P) {
                int const sy = si / A.spr, sx = si - sy * A.spr;
                pixel_system(A, tabs, Msh + lane, theta, pox + sx * A.sampling,
                    poy + sy * A.sampling, sx, sy, vis, M6, v6);
            } else {
#pragma unroll
                for (int i = 0; i < 21; ++i)
                    M6[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    v6[i] = 0.0;
            }
            // every lane is done reading its neighbour scratch before the
            // aliased rows are overwritten
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 21; ++i)
                Msh[i * 64 + lane] = M6[i];
#pragma unroll
            for (int i = 0; i < 6; ++i)
                Msh[(21 + i) * 64 + lane] = v6[i];
        }
        __syncthreads();
        // WAVES > 1: the waves take turns at phase 2 in chunk order and hand the
        // accumulators on through LDS, so that the matrix-core accumulation runs
        // through the chunks exactly as in one wave -- the patch system has the
        // same bits (a reordered sum moved three of 121,524 patches of a
        // 1920 x 1080 optimize() across a validity decision five scales later).
        // Phase 1, the long dependent chain, is what runs side by side.
#pragma unroll 1
        for (int turn = 0; turn < WAVES; ++turn) {
        if (WAVES > 1 && wave != turn) {
            __syncthreads();
            continue;
        }
        if (WAVES == 1 ? c == c_first : turn == 0) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                acc[q][0] = acc[q][1] = acc[q][2] = 0.0;
                gacc[q] = 0.0;
            }
        } else if (WAVES > 1) {
            acc[0][0] = hand[0 * 64 + lane];
            acc[0][1] = hand[1 * 64 + lane];
            acc[0][2] = hand[2 * 64 + lane];
            gacc[0] = hand[3 * 64 + lane];
        }
        // ---- phase 2: H += sum_pix D6^T (M6 D6) on the matrix cores ----
        // With PPW == 4 and 4 x 4 samples per patch, pixel slot
        // pl = 16 qq + 4 tt + kg is sample (sx, sy) = (kg, tt) of patch qq:
        // the x-functions of a lane are loop invariant, the y-functions
        // depend on tt only -- so tt is the outer loop and the three column
        // variants of D6 are formed once for the four patches.
        bool const grid4 = PPW == 4 && A.spr == 4;
        double xv[3][3];
        if (grid4) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double *X = tabs + kg * 12 + ex[v] * 3;
                xv[v][0] = X[0]; xv[v][1] = X[1]; xv[v][2] = X[2];
            }
        }
#pragma unroll 1
        for (int tt = 0; tt < 4; ++tt) {
            double D[3][6];
            if (grid4) {
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const double *Y = tabs + tt * 12 + ey[v] * 3;
                    double const y0 = Y[0], y1 = Y[1], y2 = Y[2];
                    D[v][0] = xv[v][0] * y0; D[v][1] = xv[v][1] * y0;
                    D[v][2] = xv[v][0] * y1; D[v][3] = xv[v][1] * y1;
                    D[v][4] = xv[v][2] * y0; D[v][5] = xv[v][0] * y2;
                }
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                int const q = PPW == 1 ? 0 : qq;
                int const pl = 4 * (4 * qq + tt) + kg;  // pixel slot in the wave
                if (!grid4) {
                    int si = c * SLOTS + (pl & (SLOTS - 1));
                    si = min(si, A.P - 1);
                    int const sy = si / A.spr, sx = si - sy * A.spr;
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        const double *X = tabs + sx * 12 + ex[v] * 3;
                        const double *Y = tabs + sy * 12 + ey[v] * 3;
                        double const x0 = X[0], x1 = X[1], x2 = X[2];
                        double const y0 = Y[0], y1 = Y[1], y2 = Y[2];
                        D[v][0] = x0 * y0; D[v][1] = x1 * y0; D[v][2] = x0 * y1;
                        D[v][3] = x1 * y1; D[v][4] = x2 * y0; D[v][5] = x0 * y2;
                    }
                }
                double Mx[21];
#pragma unroll
                for (int i = 0; i < 21; ++i)
                    Mx[i] = Msh[i * 64 + pl];
                double a0 = acc[q][0], a1 = acc[q][1], a2 = acc[q][2];
                double gsum = gacc[q];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    double T = 0.0;
#pragma unroll
                    for (int b = 0; b < 6; ++b)
                        T = __builtin_fma(Mx[sym6(a, b)], D[0][b], T);
                    a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(D[0][a], T, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(D[1][a], T, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(D[2][a], T, a2, 0, 0, 0);
                    gsum = __builtin_fma(Msh[(21 + a) * 64 + pl], D[0][a], gsum);
                }
                acc[q][0] = a0; acc[q][1] = a1; acc[q][2] = a2;
                gacc[q] = gsum;
            }
        }
        if (WAVES > 1) {
            if (turn + 1 < WAVES) {
                hand[0 * 64 + lane] = acc[0][0];
                hand[1 * 64 + lane] = acc[0][1];
                hand[2 * 64 + lane] = acc[0][2];
                hand[3 * 64 + lane] = gacc[0];
            }
            __syncthreads();
        }
        }   // turns
    }

    // (the wave of the last chunk holds the patch system)
    if (WAVES > 1 && wave != WAVES - 1)
        return;

    // ---- phase 3: store the per-patch systems ----
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        double gv = gacc[q];
        gv += __shfl_xor(gv, 16);
        gv += __shfl_xor(gv, 32);
        if (slot_base + q >= live_count)
            continue;
        int const patch = A.live_list[slot_base + q];
        // Packed store (common.h, PATCH_H_STRIDE): the upper triangles of the
        // four diagonal node blocks, then the six blocks (bi < bj) in full,
        // row-major (the assembly mirrors the rest).  v_mfma_f64_4x4x4 leaves element (i, j) of block slot b in
        // lane 16 i + 4 b + j, i.e. in lane (kg, col) = (i, 4 b + j).
        // (where element e of the packed record goes: PatchLayout, common.h)
        auto at = [&](int e) { return patch_h_at(A.layout, (size_t)patch, e); };
        int const bs = col >> 2, jc = col & 3;
        if (kg <= jc)   // (the diagonal blocks: upper triangle only, packed)
            A.Hp[at(patch_diag_offset(bs) + upper_block(kg, jc))] = acc[q][0];
        if (bs >= 1)
            A.Hp[at(patch_upper_offset(bs - 1, bs) + kg * 4 + jc)] = acc[q][1];
        else   // block (3, 0): the transpose of the stored block (0, 3)
            A.Hp[at(patch_upper_offset(0, 3) + jc * 4 + kg)] = acc[q][1];
        if (bs >= 2)
            A.Hp[at(patch_upper_offset(bs - 2, bs) + kg * 4 + jc)] = acc[q][2];
        if (lane < 16)
            A.gp[patch_g_at(A.layout, (size_t)patch, lane)] = gv;
    }
}

// ---------------------------------------------------------------------------
// Assembly: gather form of gauss_newton_step.cc:88-142.  One thread per
// (node, block row); the <= 4 incident patches are added in ascending patch
// id, which is the order the reference's patch loop feeds its std::map.
// Only upper-triangle entries of the per-patch system are used and mirrored
// (gauss_newton_step.cc:103, 113-119).
// ---------------------------------------------------------------------------
struct AssembleArgs {
    const double *Hp;
    const double *gp;
    PatchLayout layout;
    const uint8_t *patch_valid;
    const uint8_t *active;
    double *H9;     // [5][N][16]: slots 4..8 of the block stencil
    double *Pinv;   // [N][16]
    double *g;      // [N][4]
    int npx, npy, stride, num_nodes;
    int *status;
    uint8_t *active_next;   // cleared here for the node update of this step
    double *scalars;
};

__global__ void __launch_bounds__(256)
gn_assemble_kernel(AssembleArgs A)
{
    int const gid = blockIdx.x * blockDim.x + threadIdx.x;
    int const n = gid >> 2, r = gid & 3;
    bool const in_range = n < A.num_nodes;
    int const ix = in_range ? n % A.stride : 0;
    int const iy = in_range ? n / A.stride : 0;

    double out[9][4];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            out[s][c] = 0.0;
    double gout = 0.0;

    if (in_range && A.active[n]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int const pxq = ix - (1 - (q & 1));
            int const pyq = iy - (1 - (q >> 1));
            int const ln = 3 - q;  // local index of node n in that patch
            if (pxq < 0 || pxq >= A.npx || pyq < 0 || pyq >= A.npy)
                continue;
            int const p = pyq * A.npx + pxq;
            if (!A.patch_valid[p])
                continue;
            auto Hl = [&](int e) { return A.Hp[patch_h_at(A.layout, (size_t)p, e)]; };
            int const n00 = pyq * A.stride + pxq;
#pragma unroll
            for (int lm = 0; lm < 4; ++lm) {
                int const m = n00 + (lm & 1) + (lm >> 1) * A.stride;
                if (!A.active[m])
                    continue;
                int const dx = (lm & 1) - (1 - (q & 1));
                int const dy = (lm >> 1) - (1 - (q >> 1));
                int const slot = (dy + 1) * 3 + (dx + 1);
                // only the stored stencil slots (neighbour >= node, i.e.
                // lm >= ln) are assembled; the diagonal block is mirrored
                // from its upper triangle
                if (lm < ln)
                    continue;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double const v = lm > ln
                        ? Hl(patch_upper_offset(ln, lm) + r * 4 + c)
                        : Hl(patch_diag_offset(ln) + (r <= c ? upper_block(r, c)
                                                              : upper_block(c, r)));
                    out[slot][c] += v;
                }
            }
            gout += A.gp[patch_g_at(A.layout, (size_t)p, 4 * ln + r)];
        }
    }

    // what prepare_update_kernel does for a stand-alone update: clear the
    // re-activation flags and the counters of this step's node update
    if (in_range && r == 0)
        A.active_next[n] = 0;
    if (gid == 0) {
        // patches the construction touched (gauss_newton_step.cc:73-79) = the
        // length of the live list the patch kernel has just worked through
        A.status[I_ACTIVE_PATCHES] = A.status[I_LIVE_PATCHES];
        A.status[I_NUM_ACTIVE] = 0;
        A.scalars[S_SUMDIFF] = 0.0;
        A.scalars[S_COUNT_DIFF] = 0.0;
    }

    if (in_range) {
        size_t const N = (size_t)A.num_nodes;
        // H is symmetric (block (n, m) is the transpose of block (m, n), bit
        // for bit): only the diagonal and the four "upper" neighbour slots
        // are stored, H5[s - 4][n][16] for s = 4..8
#pragma unroll
        for (int s = 4; s < 9; ++s) {
            double *dst = A.H9 + ((size_t)(s - 4) * N + n) * 16 + r * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                dst[c] = out[s][c];
        }
        A.g[(size_t)n * 4 + r] = gout;
    }

    // block-Jacobi preconditioner: the 4 row-lanes of a node hand their row
    // of the diagonal block to lane r == 0 (block_sparse_matrix.h:300-316).
    double blk[16];
    int const base_lane = (threadIdx.x & 63) & ~3;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            blk[rr * 4 + c] = __shfl(out[4][c], base_lane + rr);
    if (in_range && r == 0) {
        double inv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            inv[i] = blk[i];
        ldl_inverse4(inv);
        bool nancheck = false;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            nancheck |= isnan(inv[i]);
        double *dst = A.Pinv + (size_t)n * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            dst[i] = nancheck ? blk[i] : inv[i];
    }
}

// Compacted list of the patches the construction has to evaluate: valid
// patches with at least one active node (gauss_newton_step.cc:73-79).  The
// order inside a block of 256 patches is kept, blocks land in the order of
// their atomic (neighbouring patches stay together for texel locality; which
// wave evaluates which patch has no effect on the result).
__global__ void __launch_bounds__(256)
live_patch_list_kernel(const uint8_t *__restrict__ patch_valid,
    const uint8_t *__restrict__ active, int npx, int stride, int num_patches,
    int *__restrict__ list, int *__restrict__ status)
{
    int const p = blockIdx.x * 256 + threadIdx.x;
    bool live = false;
    if (p < num_patches && patch_valid[p]) {
        int const ix = p % npx, iy = p / npx;
        int const n00 = iy * stride + ix;
        live = (active[n00] | active[n00 + 1] | active[n00 + stride]
            | active[n00 + stride + 1]) != 0;
    }
    __shared__ int wave_cnt[4];
    __shared__ int base;
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long const ballot = __ballot(live);
    int const before = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0)
        wave_cnt[wave] = __popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0) {
        int const total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        base = total > 0 ? atomicAdd(&status[I_LIVE_PATCHES], total) : 0;
    }
    __syncthreads();
    if (live) {
        int off = base + before;
        for (int wv = 0; wv < wave; ++wv)
            off += wave_cnt[wv];
        list[off] = p;
    }
}

int
live_patch_list_launch(smvs_ctx *ctx)
{
    SMVS_HIP_CHECK(hipMemsetAsync(ctx->status + I_LIVE_PATCHES, 0, sizeof(int),
        ctx->stream));
    ScopedKernelTimer timer(ctx, SMVS_K_MISC);
    hipLaunchKernelGGL(live_patch_list_kernel,
        dim3((unsigned)((ctx->num_patches + 255) / 256)), dim3(256), 0,
        ctx->stream, ctx->patch_valid, ctx->active, ctx->npx, ctx->node_stride,
        ctx->num_patches, ctx->live_list, ctx->status);
    SMVS_HIP_CHECK(hipGetLastError());
    return SMVS_OK;
}

static int
sampling_for_scale(int scale)
{
    // gauss_newton_step.cc:157-161
    int sampling = 4;
    if (scale < 5)
        sampling = 2;
    if (scale < 3)
        sampling = 1;
    return sampling;
}

int
gn_construct_launch(smvs_ctx *ctx, double reg, double light_reg,
    bool use_lighting, int known_live, bool skip_assembly, bool check_stop)
{
    // Shared by smvs_gn_construct and smvs_gn_run_loop: the patch kernel
    // dereferences every neighbour's planes and the main gradient.
    for (int j = 0; j < ctx->n_subs; ++j)
        if (!((ctx->planes_ok >> j) & 1u)) {
            set_error("gn_construct: sub view %d has no gradient / Hessian "
                "planes (smvs_ctx_upload_sub or smvs_ctx_set_scale first)", j);
            return SMVS_ERR_STATE;
        }
    if (use_lighting && !ctx->has_shading) {
        set_error("gn_construct: lighting given but no shading planes");
        return SMVS_ERR_STATE;
    }
    SMVS_REQUIRE(reg >= 0.0 && light_reg >= 0.0, "negative regularization");
    PatchKernelArgs A;
    A.nodes = ctx->nodes;
    A.patch_valid = ctx->patch_valid;
    A.patch_vis = ctx->patch_vis;
    A.active = ctx->active;
    A.live_list = ctx->live_list;
    A.hermite_tab = ctx->hermite_tab;
    A.main_grad = ctx->main_grad;
    A.main_shading = ctx->main_shading;
    A.main_shading_grad = ctx->main_shading_grad;
    A.subs = ctx->subs_dev;
    A.cams = ctx->cams;
    A.lighting = ctx->lighting;
    A.Hp = ctx->Hp;
    A.gp = ctx->gp;
    A.layout = patch_layout(ctx);
    A.W = ctx->width;
    A.H = ctx->height;
    A.npx = ctx->npx;
    A.npy = ctx->npy;
    A.stride = ctx->node_stride;
    A.ps = ctx->patchsize;
    A.start_x = ctx->start_x;
    A.start_y = ctx->start_y;
    A.sampling = sampling_for_scale(ctx->scale);
    if (A.sampling > A.ps)
        A.sampling = A.ps;
    A.spr = A.ps / A.sampling;
    A.P = A.spr * A.spr;
    A.n_subs = ctx->n_subs;
    A.num_patches = ctx->num_patches;
    A.flen = (double)ctx->flen;
    A.inv_flen = (double)ctx->inv_flen;
    A.inv_f = 1.0 / A.flen;
    A.f2inv = 1.0 / (A.flen * A.flen);
    A.reg = reg;
    A.light_reg = light_reg;
    A.use_lighting = use_lighting ? 1 : 0;
    A.status = ctx->status;
    A.check_stop = check_stop ? 1 : 0;

    int const scratch_rows = 5 * (ctx->n_subs - 1) > 27 ? 5 * (ctx->n_subs - 1) : 27;
    size_t const lds = (size_t)(scratch_rows * 64 + A.spr * 12) * sizeof(double);
    bool const four = A.P <= 16;
    int const ppw = four ? 4 : 1;
    // grid = the live list when the host knows its length (read back after the
    // previous step of the same Newton loop), otherwise every patch (the
    // surplus blocks leave at once)
    int live = ctx->num_patches;
    if (known_live >= 0) {
        live = known_live;
    } else {
        int const rc = live_patch_list_launch(ctx);
        if (rc != SMVS_OK)
            return rc;
    }
    // (8 XCD bands of ceil(blocks / 8) blocks each)
    unsigned const live_blocks = (unsigned)((live + ppw - 1) / ppw);
    unsigned const blocks = ((live_blocks + 7u) >> 3) << 3;
    if (blocks > 0) {
        ScopedKernelTimer timer(ctx, SMVS_K_PATCH);
        // (SMVS_PATCH_SPLIT=0: a patch's chunks one after the other in one wave,
        // rounds 1-4)
        static bool const split_off = [] {
            const char *e = std::getenv("SMVS_PATCH_SPLIT");
            return e != nullptr && e[0] == '0';
        }();
        int const chunks = four ? 1 : (A.P + 63) / 64;
        size_t const lds4 = (size_t)(4 * scratch_rows * 64 + A.spr * 12 + 4 * 64) * sizeof(double);
        if (four)
            hipLaunchKernelGGL((gn_patch_kernel<4, 1, false>), dim3(blocks), dim3(64), lds,
                ctx->stream, A);
        else if (chunks == 4 && !split_off
            && allow_dynamic_lds(ctx->device,
                   reinterpret_cast<const void *>(gn_patch_kernel<1, 4, false>), lds4) == SMVS_OK)
            // a wave per chunk (scale 6: 16 x 16 samples per patch)
            hipLaunchKernelGGL((gn_patch_kernel<1, 4, false>), dim3(blocks), dim3(256), lds4,
                ctx->stream, A);
        else if (chunks == 1 && !split_off)
            hipLaunchKernelGGL((gn_patch_kernel<1, 1, true>), dim3(blocks), dim3(64), lds,
                ctx->stream, A);
        else
            hipLaunchKernelGGL((gn_patch_kernel<1, 1, false>), dim3(blocks), dim3(64), lds,
                ctx->stream, A);
    }
    SMVS_HIP_CHECK(hipGetLastError());

    if (skip_assembly) {
        // the resident solver assembles H, g, P from the per-patch systems
        // itself; nothing is materialised for smvs_gn_download
        ctx->has_system = false;
        ctx->cg_use_active = true;
        ctx->update_prepared = true;
        return SMVS_OK;
    }
    return gn_assemble_launch(ctx);
}

int
gn_assemble_launch(smvs_ctx *ctx)
{
    AssembleArgs B;
    B.Hp = ctx->Hp;
    B.gp = ctx->gp;
    B.layout = patch_layout(ctx);
    B.patch_valid = ctx->patch_valid;
    B.active = ctx->active;
    B.status = ctx->status;
    B.active_next = ctx->active_next;
    B.scalars = ctx->scalars;
    B.H9 = ctx->H9;
    B.Pinv = ctx->Pinv;
    B.g = ctx->g;
    B.npx = ctx->npx;
    B.npy = ctx->npy;
    B.stride = ctx->node_stride;
    B.num_nodes = ctx->num_nodes;
    {
        ScopedKernelTimer timer(ctx, SMVS_K_ASSEMBLE);
        hipLaunchKernelGGL(gn_assemble_kernel,
            dim3((unsigned)(((size_t)ctx->num_nodes * 4 + 255) / 256)),
            dim3(256), 0, ctx->stream, B);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    ctx->has_system = true;
    ctx->cg_use_active = true;
    ctx->update_prepared = true;
    return SMVS_OK;
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_gn_construct(smvs_ctx *ctx, double regularization,
    double light_surf_regularization, const double *lighting16,
    int *num_active_patches)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_surface || !ctx->has_cameras) {
        set_error("smvs_gn_construct: cameras and surface must be set first");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    if (lighting16 != nullptr)
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->lighting, lighting16,
            16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc = gn_construct_launch(ctx, regularization,
        light_surf_regularization, lighting16 != nullptr);
    if (rc != SMVS_OK)
        return rc;
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->status_host, ctx->status,
        sizeof(int) * I_NUM, hipMemcpyDeviceToHost, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (num_active_patches != nullptr)
        *num_active_patches = ctx->status_host[I_ACTIVE_PATCHES];
    return SMVS_OK;
}

extern "C" int
smvs_gn_download(smvs_ctx *ctx, double *H9, double *g, double *P)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_system) {
        set_error("smvs_gn_download: no system constructed");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const N = (size_t)ctx->num_nodes;
    if (H9 != nullptr) {
        std::vector<double> tmp(N * 5 * 16);
        SMVS_HIP_CHECK(hipMemcpyAsync(tmp.data(), ctx->H9,
            tmp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        memset(H9, 0, N * 9 * 16 * sizeof(double));
        long const stride = ctx->node_stride;
        for (size_t s = 4; s < 9; ++s)
            for (size_t n = 0; n < N; ++n) {
                const double *blk = tmp.data() + ((s - 4) * N + n) * 16;
                memcpy(H9 + (n * 9 + s) * 16, blk, 16 * sizeof(double));
                if (s == 4)
                    continue;
                // mirrored block of the neighbour: slot 8 - s at node m
                long const dx = (long)(s % 3) - 1, dy = (long)(s / 3) - 1;
                long const mx = (long)(n % stride) + dx;
                long const m = (long)n + dy * stride + dx;
                if (mx < 0 || mx >= stride || m < 0 || m >= (long)N)
                    continue;
                double *dst = H9 + ((size_t)m * 9 + (8 - s)) * 16;
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c)
                        dst[c * 4 + r] = blk[r * 4 + c];
            }
    }
    if (g != nullptr)
        SMVS_HIP_CHECK(hipMemcpyAsync(g, ctx->g, N * 4 * sizeof(double),
            hipMemcpyDeviceToHost, ctx->stream));
    if (P != nullptr)
        SMVS_HIP_CHECK(hipMemcpyAsync(P, ctx->Pinv, N * 16 * sizeof(double),
            hipMemcpyDeviceToHost, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_gn_upload(smvs_ctx *ctx, const double *H9, const double *g,
    const double *P)
{
    SMVS_REQUIRE(ctx && H9 && g && P, "null argument");
    if (!ctx->has_surface) {
        set_error("smvs_gn_upload: no surface");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const N = (size_t)ctx->num_nodes;
    // the system must be symmetric: the upper slots are taken
    std::vector<double> tmp(N * 5 * 16);
    for (size_t s = 4; s < 9; ++s)
        for (size_t n = 0; n < N; ++n)
            memcpy(tmp.data() + ((s - 4) * N + n) * 16, H9 + (n * 9 + s) * 16,
                16 * sizeof(double));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->H9, tmp.data(),
        tmp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->g, g, N * 4 * sizeof(double),
        hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->Pinv, P, N * 16 * sizeof(double),
        hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->update_prepared = false;
    ctx->has_system = true;
    ctx->cg_use_active = false;
    return SMVS_OK;
}

extern "C" int
smvs_gn_download_patch_systems(smvs_ctx *ctx, double *Hp, double *gp)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    if (!ctx->has_system) {
        set_error("smvs_gn_download_patch_systems: no system constructed");
        return SMVS_ERR_STATE;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const P = (size_t)ctx->num_patches;
    PatchLayout const L = patch_layout(ctx);
    // (the buffers as they lie on the device: the record of the last patch
    // ends where its last element does, in either layout)
    size_t const h_doubles = P == 0 ? 0 : patch_h_at(L, P - 1, PATCH_H_STRIDE - 1) + 1;
    size_t const g_doubles = P == 0 ? 0 : patch_g_at(L, P - 1, 15) + 1;
    std::vector<double> packed, gpacked;
    if (Hp != nullptr) {
        packed.resize(h_doubles);
        SMVS_HIP_CHECK(hipMemcpyAsync(packed.data(), ctx->Hp,
            packed.size() * sizeof(double), hipMemcpyDeviceToHost,
            ctx->stream));
    }
    if (gp != nullptr) {
        gpacked.resize(g_doubles);
        SMVS_HIP_CHECK(hipMemcpyAsync(gpacked.data(), ctx->gp, g_doubles * sizeof(double),
            hipMemcpyDeviceToHost, ctx->stream));
    }
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (gp != nullptr)
        for (size_t p = 0; p < P; ++p)
            for (int e = 0; e < 16; ++e)
                gp[p * 16 + e] = gpacked[patch_g_at(L, p, e)];
    if (Hp != nullptr) {
        // unpack the upper block triangle; blocks below it are the mirror
        for (size_t p = 0; p < P; ++p) {
            auto src = [&](int e) { return packed[patch_h_at(L, p, e)]; };
            double *dst = Hp + p * 256;
            for (int bi = 0; bi < 4; ++bi)
                for (int bj = bi; bj < 4; ++bj) {
                    for (int r = 0; r < 4; ++r)
                        for (int c = 0; c < 4; ++c) {
                            double const v = bj > bi
                                ? src(patch_upper_offset(bi, bj) + r * 4 + c)
                                : src(patch_diag_offset(bi) + (r <= c ? upper_block(r, c)
                                                                      : upper_block(c, r)));
                            dst[(4 * bi + r) * 16 + 4 * bj + c] = v;
                            if (bj > bi)
                                dst[(4 * bj + c) * 16 + 4 * bi + r] = v;
                        }
                }
        }
    }
    return SMVS_OK;
}
