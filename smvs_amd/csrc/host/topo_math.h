// Arithmetic shared by the C++ host mirror (g++) and the device topology
// kernels (hipcc): bicubic Hermite patch evaluation, the reprojection of a
// pixel into a neighbour view and bilinear taps.  One source for both sides
// so that the per-patch decisions of create_subview_surfaces / cut_boundaries
// (lib/depth_optimizer.cc:360-604, 747-912) come out identical on the host
// and on the device: plain IEEE operations in source order, no contraction.
#ifndef SMVS_TOPO_MATH_H
#define SMVS_TOPO_MATH_H

#if defined(__HIPCC__)
#define SMVS_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#define SMVS_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define SMVS_HD inline
#define SMVS_NO_CONTRACT
#endif

namespace smvs_topo {

// 1-D cubic Hermite basis and derivatives: value@0, value@1, slope@0, slope@1
SMVS_HD void
hermite(double t, int k, double* b)
{
    SMVS_NO_CONTRACT
    double const t2 = t * t, t3 = t2 * t;
    switch (k) {
    case 0:
        b[0] = 1.0 - 3.0 * t2 + 2.0 * t3; b[1] = 3.0 * t2 - 2.0 * t3;
        b[2] = t - 2.0 * t2 + t3;         b[3] = t3 - t2;
        break;
    case 1:
        b[0] = -6.0 * t + 6.0 * t2;       b[1] = 6.0 * t - 6.0 * t2;
        b[2] = 1.0 - 4.0 * t + 3.0 * t2;  b[3] = 3.0 * t2 - 2.0 * t;
        break;
    default:
        b[0] = -6.0 + 12.0 * t;           b[1] = 6.0 - 12.0 * t;
        b[2] = -4.0 + 6.0 * t;            b[3] = 6.0 * t - 2.0;
        break;
    }
}

// nodes16 = {n00, n10, n01, n11} x {f, dx, dy, dxy}; derivative orders
// (kx, ky) at (x, y) in [0,1]^2 (patch units)
SMVS_HD double
patch_eval(double const* n, double x, double y, int kx, int ky)
{
    SMVS_NO_CONTRACT
    double bx[4], by[4];
    hermite(x, kx, bx);
    hermite(y, ky, by);
    double r = 0.0;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            double const* nd = n + 4 * (2 * b + a);
            r += nd[0] * bx[a] * by[b] + nd[1] * bx[2 + a] * by[b]
                + nd[2] * bx[a] * by[2 + b] + nd[3] * bx[2 + a] * by[2 + b];
        }
    return r;
}

// Correspondence (lib/correspondence.cc:20-51, 88-100) for one pixel
struct Warp
{
    double p, q, r, a, b, d;
    SMVS_HD Warp(double const* M, double const* t, double u, double v, double w)
    {
        SMVS_NO_CONTRACT
        p = M[0] * u + M[1] * v + M[2];
        q = M[3] * u + M[4] * v + M[5];
        r = M[6] * u + M[7] * v + M[8];
        a = w * p + t[0];
        b = w * q + t[1];
        d = w * r + t[2];
    }
    SMVS_HD double x(void) const { return a / d; }
    SMVS_HD double y(void) const { return b / d; }
    SMVS_HD void jacobian(double const* M, double w, double wx, double wy,
        double* jac) const
    {
        SMVS_NO_CONTRACT
        double const d2 = d * d;
        jac[0] = (wx * p + w * M[0]) / d - a * (wx * r + w * M[6]) / d2;
        jac[2] = (wy * p + w * M[1]) / d - a * (wy * r + w * M[7]) / d2;
        jac[1] = (wx * q + w * M[3]) / d - b * (wx * r + w * M[6]) / d2;
        jac[3] = (wy * q + w * M[4]) / d - b * (wy * r + w * M[7]) / d2;
    }
};

// mve::Image<float>::linear_at on interleaved channels [MVE-unverified]
template <typename Ptr>
SMVS_HD float
linear_at(Ptr data, int w, int h, int c, float x, float y, int ch)
{
    SMVS_NO_CONTRACT
    x = x < 0.0f ? 0.0f : (x > (float)(w - 1) ? (float)(w - 1) : x);
    y = y < 0.0f ? 0.0f : (y > (float)(h - 1) ? (float)(h - 1) : y);
    int const fx = (int)x, fy = (int)y;
    int const fx1 = fx + 1 < w - 1 ? fx + 1 : w - 1;
    int const fy1 = fy + 1 < h - 1 ? fy + 1 : h - 1;
    float const w1 = x - (float)fx, w0 = 1.0f - w1;
    float const w3 = y - (float)fy, w2 = 1.0f - w3;
    float const v00 = data[((long)fy * w + fx) * c + ch];
    float const v10 = data[((long)fy * w + fx1) * c + ch];
    float const v01 = data[((long)fy1 * w + fx) * c + ch];
    float const v11 = data[((long)fy1 * w + fx1) * c + ch];
    return v00 * (w0 * w2) + v10 * (w1 * w2) + v01 * (w0 * w3)
        + v11 * (w1 * w3);
}

// Sample list of ncc_for_patch (lib/depth_optimizer.cc:803-857) relative to
// the patch origin.  The reference grows the list while walking it; which
// samples are appended depends only on the patch size and on five border
// predicates, so the walk is done once per (patch size, predicates) and
// replayed.  src >= 0: the sample takes the depth of grid sample src
// (i = src % ps, j = src / ps); src = -1 - c: the depth of corner node c.
struct NccSample { short dx, dy, src; };

enum { NCC_CORNERS = 1, NCC_TOP = 2, NCC_BOTTOM = 4, NCC_LEFT = 8, NCC_RIGHT = 16 };

SMVS_HD int
ncc_flags(int px, int py, int ps, int width, int height)
{
    int const min_x = px, min_y = py, max_x = px + ps, max_y = py + ps;
    int f = 0;
    if (min_x > 1 && max_x < width - 2 && min_y > 1 && max_y < height - 2)
        f |= NCC_CORNERS;
    if (min_y > 2)
        f |= NCC_TOP;
    if (max_y < height - 3)
        f |= NCC_BOTTOM;
    if (min_x > 2)
        f |= NCC_LEFT;
    if (max_x < width - 3)
        f |= NCC_RIGHT;
    return f;
}

} // namespace smvs_topo

#include <vector>

namespace smvs_topo {

// The walk of lib/depth_optimizer.cc:803-857 in patch-relative coordinates.
inline std::vector<NccSample>
build_ncc_template(int ps, int flags)
{
    std::vector<NccSample> s;
    s.reserve((size_t)ps * ps + 8 * (size_t)ps + 32);
    for (int j = 0; j < ps; ++j)
        for (int i = 0; i < ps; ++i)
            s.push_back({ (short)i, (short)j, (short)(j * ps + i) });
    if (flags & NCC_CORNERS) {
        s.push_back({ (short)-1, (short)-1, (short)-1 });
        s.push_back({ (short)(ps + 1), (short)-1, (short)-2 });
        s.push_back({ (short)-1, (short)(ps + 1), (short)-3 });
        s.push_back({ (short)(ps + 1), (short)(ps + 1), (short)-4 });
    }
    // the list grows while it is walked
    for (std::size_t i = 0; i < s.size(); ++i) {
        NccSample const c = s[i];
        if ((flags & NCC_TOP) && c.dy == 0) {
            s.push_back({ c.dx, (short)(c.dy - 2), c.src });
            s.push_back({ c.dx, (short)(c.dy - 1), c.src });
        }
        if ((flags & NCC_BOTTOM) && c.dy == ps) {
            s.push_back({ c.dx, (short)(c.dy + 2), c.src });
            s.push_back({ c.dx, (short)(c.dy + 1), c.src });
        }
        if ((flags & NCC_LEFT) && c.dx == 0) {
            s.push_back({ (short)(c.dx - 2), c.dy, c.src });
            s.push_back({ (short)(c.dx - 1), c.dy, c.src });
        }
        if ((flags & NCC_RIGHT) && c.dx == ps) {
            s.push_back({ (short)(c.dx + 2), c.dy, c.src });
            s.push_back({ (short)(c.dx + 1), c.dy, c.src });
        }
    }
    return s;
}

} // namespace smvs_topo

#endif
