// Scale space on the device: byte -> float, separable Gaussian blur,
// luminance and the 3x3 quadratic-fit gradient / Hessian.
//
// Replaces the host work of StereoView::StereoView (reference:
// lib/stereo_view.cc:16-22) and StereoView::set_scale /
// compute_gradients_and_hessian (:24-46, :97-188) for every view of a context.
// All float arithmetic keeps the reference's operation order with FMA
// contraction off, so the planes are bit-identical to the host computation
// (the Gaussian weights are evaluated on the host with the same expf).
// MVE's blur_gaussian / desaturate semantics are recalled [MVE-unverified].
#include "common.h"

#include <cstdlib>

#include <cmath>
#include <vector>

namespace smvs_hip {

__global__ void __launch_bounds__(256)
byte_to_float_kernel(const uint8_t *__restrict__ in, float *__restrict__ out,
    size_t n)
{
#pragma clang fp contract(off)
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = (float)in[i] / 255.0f;
}

// Sixteen bytes per thread (one 16-byte load, four 16-byte stores) out of the
// context's own staging buffers (device allocations: 256-byte aligned) instead
// of a byte per thread: 24 k workgroups of four-byte stores were 16 us
// for a 1920 x 1080 x 3 image whose 31 MB cost 8 (round 6).
__global__ void __launch_bounds__(256)
byte_to_float16_kernel(const uint8_t *__restrict__ in, float *__restrict__ out,
    size_t n)
{
#pragma clang fp contract(off)
    size_t const t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t const chunks = n / 16;
    if (t < chunks) {
        uint4 const v = reinterpret_cast<const uint4 *>(in)[t];
        uint32_t const w[4] = { v.x, v.y, v.z, v.w };
        float4 *dst = reinterpret_cast<float4 *>(out + t * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = make_float4((float)(w[q] & 0xFFu) / 255.0f,
                (float)((w[q] >> 8) & 0xFFu) / 255.0f, (float)((w[q] >> 16) & 0xFFu) / 255.0f,
                (float)(w[q] >> 24) / 255.0f);
    }
    // (the last n % 16 bytes: the first threads of the grid)
    if (t < n - chunks * 16)
        out[chunks * 16 + t] = (float)in[chunks * 16 + t] / 255.0f;
}

// its grid for n bytes
static inline unsigned
byte_to_float_blocks(size_t n)
{
    size_t const threads = std::max<size_t>(n / 16, n % 16);
    return (unsigned)std::max<size_t>(1, (threads + 255) / 256);
}

// The same conversion out of the caller's page-locked memory (the bytes cross
// the bus as the kernel reads them) with a SMALL grid that strides over the
// image, sixteen bytes per thread and step: enough requests in flight for the
// bus, and the rest of the chip stays free for the context's own kernels
// (SMVS_UPLOAD_STREAM=kernel, smvs_ctx_upload_image_async).
constexpr int UPLOAD_BLOCKS = 64;

__global__ void __launch_bounds__(256)
host_bytes_to_float_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, size_t n)
{
#pragma clang fp contract(off)
    size_t const threads = (size_t)gridDim.x * blockDim.x;
    size_t const t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t const chunks = n / 16;
    for (size_t c = t; c < chunks; c += threads) {
        uint4 const v = reinterpret_cast<const uint4 *>(in)[c];
        uint32_t const w[4] = { v.x, v.y, v.z, v.w };
        float4 *dst = reinterpret_cast<float4 *>(out + c * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = make_float4((float)(w[q] & 0xFFu) / 255.0f,
                (float)((w[q] >> 8) & 0xFFu) / 255.0f, (float)((w[q] >> 16) & 0xFFu) / 255.0f,
                (float)(w[q] >> 24) / 255.0f);
    }
    for (size_t i = chunks * 16 + t; i < n; i += threads)
        out[i] = (float)in[i] / 255.0f;
}

// The Gaussian taps of one scale, handed to the kernels by value: the tap
// index is uniform across a wave, so the weights come out of the kernel
// arguments with scalar loads.  wsum is Accum<float>'s weight total: the
// reference adds every tap's weight in tap order for every element, clamped
// or not, so it is one number per scale, summed here in that order.
struct BlurTaps {
    float k[64];
    float wsum;
    int ks;
};

// Separable pass along x: v += value * weight per tap in ascending tap order,
// result v / wsum, borders clamped -- the same operations in the same order
// per element as the reference's loop (bit-exact against the oracle).  One
// thread per element of a row (grid y = image row): the taps of a wave are
// contiguous in memory; workgroups whose taps stay inside the row (all but
// the first and the last of a row) skip the clamps.
// KS > 0: the half width is a compile-time constant (the scales 0 .. 6 of
// stereo_view.cc:29-31 give 1, 2, 2, 4, 7, 12, 23): the tap loops unroll, the
// weights sit in scalar registers and the loads of a thread are all in
// flight at once.  KS == 0: any half width, taps.ks at run time.
constexpr int BLUR_X_LDS_FROM = 12;   // half width from which blur_x_kernel stages its row segment in LDS

// ROWS (round 6, the per-thread-loads form only): image rows per workgroup.  With
// one row a 1920 x 1080 x 3 image was 24,840 workgroups of one output per
// thread; the rows of a thread are independent, so their loads are all in flight
// together and a launch is a quarter of the workgroups (SMVS_BLUR_X_ROWS=1: one
// row, A/B).
constexpr int BLUR_X_ROWS = 4;

template <int C, int KS, int ROWS>
__global__ void __launch_bounds__(256)
blur_x_kernel(const float *__restrict__ in, float *__restrict__ out, int w, int h,
    BlurTaps taps)
{
#pragma clang fp contract(off)
    static_assert(ROWS == 1 || KS < BLUR_X_LDS_FROM, "the LDS form takes one row");
    int const e0 = (int)(blockIdx.x * blockDim.x);
    int const e = e0 + (int)threadIdx.x;       // element of the row
    int const y = (int)blockIdx.y * ROWS;
    int const ks = KS > 0 ? KS : taps.ks;
    const float *row = in + (size_t)y * w * C;
    if constexpr (KS >= BLUR_X_LDS_FROM) {
        // Round 6: the workgroup's pixels and their KS neighbours on either side
        // go through LDS once -- every element is a tap of 2 KS + 1 outputs, and
        // those 47 loads per thread at KS = 23 through the vector L1 were the
        // kernel's time (47 -> 29 us per image at scale 6, 29 -> 22-27 at scale 5; at
        // half widths up to 7 the staging and its barrier cost more than the loads
        // they replace -- 22 -> 25, 20 -> 24, 18 -> 23 us measured -- so those keep
        // the per-thread loads below).  A staged value is the CLAMPED
        // pixel's, so the taps need no clamp and see exactly the operands of the
        // per-thread loop below, in the same ascending order.
        constexpr int SPAN = (255 / C + 2 + 2 * KS) * C;    // pixels the 256 elements touch
        __shared__ float seg[SPAN];
        if (e0 >= w * C)
            return;
        int const x_first = e0 / C;
        for (int j = (int)threadIdx.x; j < SPAN; j += 256) {
            int const xp = x_first - KS + j / C;
            int const xx = min(max(xp, 0), w - 1);
            seg[j] = row[xx * C + (j - (j / C) * C)];
        }
        __syncthreads();
        if (e >= w * C)
            return;
        int const x = e / C, cc = e - x * C;
        const float *p = seg + (x - x_first + KS) * C + cc;
        float av = 0.0f;
#pragma unroll
        for (int k = -KS; k <= KS; ++k)
            av += p[k * C] * taps.k[k < 0 ? -k : k];
        out[(size_t)y * w * C + e] = av / taps.wsum;
        return;
    }
    bool const interior = e0 / C - ks >= 0 && (e0 + 255) / C + ks <= w - 1;
    if (e >= w * C)
        return;
    if (interior && KS > 0) {
        // every load of the thread's rows is issued before its values are used
        float v[ROWS][2 * KS + 1];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            // (a row below the image repeats the last one; it is not stored)
            const float *p = in + (size_t)min(y + r, h - 1) * w * C + e;
#pragma unroll
            for (int k = -KS; k <= KS; ++k)
                v[r][k + KS] = p[k * C];
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float av = 0.0f;
#pragma unroll
            for (int k = -KS; k <= KS; ++k)
                av += v[r][k + KS] * taps.k[k < 0 ? -k : k];
            if (y + r < h)
                out[(size_t)(y + r) * w * C + e] = av / taps.wsum;
        }
        return;
    }
#pragma unroll 1
    for (int r = 0; r < ROWS && y + r < h; ++r) {
        const float *rw = row + (size_t)r * w * C;
        float av = 0.0f;
        if (interior) {
            const float *p = rw + e;
            for (int k = -ks; k <= ks; ++k)
                av += p[k * C] * taps.k[k < 0 ? -k : k];
        } else {
            int const x = e / C, cc = e - x * C;
            for (int k = -ks; k <= ks; ++k) {
                int const xx = min(max(x + k, 0), w - 1);
                av += rw[xx * C + cc] * taps.k[k < 0 ? -k : k];
            }
        }
        out[(size_t)(y + r) * w * C + e] = av / taps.wsum;
    }
}

// Separable pass along y, BLUR_ROWS output rows per thread: a loaded value
// feeds the (up to) BLUR_ROWS outputs it is a tap of, each of which still
// sees its taps in ascending order.
#define BLUR_ROWS 4
template <int KS>
__global__ void __launch_bounds__(256)
blur_y_kernel(const float *__restrict__ in, float *__restrict__ out, int row_len,
    int h, BlurTaps taps)
{
#pragma clang fp contract(off)
    int const e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int const y0 = (int)blockIdx.y * BLUR_ROWS;
    int const ks = KS > 0 ? KS : taps.ks;
    if (e >= row_len)
        return;
    float av[BLUR_ROWS];
#pragma unroll
    for (int j = 0; j < BLUR_ROWS; ++j)
        av[j] = 0.0f;
    if (KS > 0) {
        constexpr int N = 2 * KS + BLUR_ROWS;
        // (in groups: every load of a group is issued before its values are used)
        constexpr int GROUP = 16;
#pragma unroll
        for (int g = 0; g < N; g += GROUP) {
            float v[GROUP];
#pragma unroll
            for (int i = 0; i < GROUP; ++i)
                if (g + i < N) {
                    int const yy = min(max(y0 + g + i - KS, 0), h - 1);
                    v[i] = in[(size_t)yy * row_len + e];
                }
#pragma unroll
            for (int i = 0; i < GROUP; ++i)
                if (g + i < N) {
                    int const t = g + i - KS;
#pragma unroll
                    for (int j = 0; j < BLUR_ROWS; ++j) {
                        int const k = t - j;     // tap of output row y0 + j
                        if (k >= -KS && k <= KS)
                            av[j] += v[i] * taps.k[k < 0 ? -k : k];
                    }
                }
        }
    } else {
        for (int t = -ks; t <= ks + BLUR_ROWS - 1; ++t) {
            int const yy = min(max(y0 + t, 0), h - 1);
            float const v = in[(size_t)yy * row_len + e];
#pragma unroll
            for (int j = 0; j < BLUR_ROWS; ++j) {
                int const k = t - j;
                if (k >= -ks && k <= ks)
                    av[j] += v * taps.k[k < 0 ? -k : k];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < BLUR_ROWS; ++j)
        if (y0 + j < h)
            out[(size_t)(y0 + j) * row_len + e] = av[j] / taps.wsum;
}

template <int KS>
static void
launch_blur_x(hipStream_t stream, const float *in, float *tmp, int w, int h, int c,
    BlurTaps const &taps)
{
    unsigned const bx = (unsigned)((w * c + 255) / 256);
    constexpr bool staged = KS >= BLUR_X_LDS_FROM;
    const char *env = std::getenv("SMVS_BLUR_X_ROWS");
    if (staged || (env != nullptr && env[0] == '1')) {
        if (c == 1)
            hipLaunchKernelGGL((blur_x_kernel<1, KS, 1>), dim3(bx, (unsigned)h), dim3(256), 0,
                stream, in, tmp, w, h, taps);
        else
            hipLaunchKernelGGL((blur_x_kernel<3, KS, 1>), dim3(bx, (unsigned)h), dim3(256), 0,
                stream, in, tmp, w, h, taps);
        return;
    }
    constexpr int ROWS = staged ? 1 : BLUR_X_ROWS;
    dim3 const grid(bx, (unsigned)((h + ROWS - 1) / ROWS));
    if (c == 1)
        hipLaunchKernelGGL((blur_x_kernel<1, KS, ROWS>), grid, dim3(256), 0, stream, in, tmp,
            w, h, taps);
    else
        hipLaunchKernelGGL((blur_x_kernel<3, KS, ROWS>), grid, dim3(256), 0, stream, in, tmp,
            w, h, taps);
}

template <int KS>
static void
launch_blur_ks(hipStream_t stream, const float *in, float *tmp, float *out, int w, int h,
    int c, BlurTaps const &taps)
{
    unsigned const bx = (unsigned)((w * c + 255) / 256);
    launch_blur_x<KS>(stream, in, tmp, w, h, c, taps);
    // The y pass re-reads every input row for the 2 KS / BLUR_ROWS + 1 row blocks
    // it is a tap of; the reuse has to happen in an L2, and each XCD has its own.
    // Workgroups go to the XCDs round robin by linear id (x fastest): with the
    // 23 column blocks of a 1920 x 3 row, vertically adjacent row blocks landed
    // on different XCDs and the kernel fetched 8 x its input (225 MB at KS = 23,
    // profiles/r6_hbm_traffic.txt).  A row of the grid padded to a multiple of 8
    // keeps a column of blocks on ONE XCD (the padding blocks leave at once).
    // SMVS_BLUR_XCD=0: unpadded (A/B).
    static bool const pad = [] {
        const char *e = std::getenv("SMVS_BLUR_XCD");
        return !(e != nullptr && e[0] == '0');
    }();
    unsigned const by = pad ? (bx + 7u) & ~7u : bx;
    hipLaunchKernelGGL(blur_y_kernel<KS>,
        dim3(by, (unsigned)((h + BLUR_ROWS - 1) / BLUR_ROWS)), dim3(256), 0, stream, tmp,
        out, w * c, h, taps);
}

static void
launch_blur(hipStream_t stream, const float *in, float *tmp, float *out, int w, int h,
    int c, BlurTaps const &taps)
{
    switch (taps.ks) {
    case 1: launch_blur_ks<1>(stream, in, tmp, out, w, h, c, taps); break;
    case 2: launch_blur_ks<2>(stream, in, tmp, out, w, h, c, taps); break;
    case 4: launch_blur_ks<4>(stream, in, tmp, out, w, h, c, taps); break;
    case 7: launch_blur_ks<7>(stream, in, tmp, out, w, h, c, taps); break;
    case 12: launch_blur_ks<12>(stream, in, tmp, out, w, h, c, taps); break;
    case 23: launch_blur_ks<23>(stream, in, tmp, out, w, h, c, taps); break;
    default: launch_blur_ks<0>(stream, in, tmp, out, w, h, c, taps); break;
    }
}

// luminance (0.21, 0.72, 0.07) + quadratic fit on the 3x3 window, double
// accumulation in the reference's order (stereo_view.cc:167-187)
struct FitMatrix { double m[6][9]; };

// GRAD_ROWS output rows per workgroup (round 6; one row before): the luminance
// of a pixel is formed once for the up to three output rows it is a tap of
// instead of once per row -- the rows' windows overlap in LDS, not in the L2 --
// and a launch is 1,080 workgroups of real work instead of 8,640 short ones.
constexpr int GRAD_ROWS = 8;

// r = M v of stereo_view.cc:167-187 without the terms whose coefficient is zero
// (round 6: 11 of the 45 products of the rows that are used -- the constant row
// never is).  Such a term is +0 (the window holds blurred luminances: finite,
// never negative), and a sum that starts at +0 is never -0, so leaving the term
// out changes no bit: rows x (a = 0 columns) and y (b = 0 columns) keep six
// terms, row xy its four corners, in ascending column order like the full rows.
// The main view has no Hessian plane (want_hess false, wave-uniform): its three
// rows are not formed at all.
__device__ __forceinline__ void
quadratic_fit(FitMatrix const &fit, const double (&v)[9], bool want_hess, float2 *g, float4 *hs)
{
#pragma clang fp contract(off)
    double r3 = 0.0, r4 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i / 3 != 1)          // a != 0
            r3 += fit.m[3][i] * v[i];
        if (i % 3 != 1)          // b != 0
            r4 += fit.m[4][i] * v[i];
    }
    *g = make_float2((float)r3, (float)r4);
    if (want_hess) {
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            r0 += fit.m[0][i] * v[i];
            r1 += fit.m[1][i] * v[i];
            if (i / 3 != 1 && i % 3 != 1)      // a b != 0
                r2 += fit.m[2][i] * v[i];
        }
        *hs = make_float4((float)(2.0 * r0), (float)r2, (float)(2.0 * r1), 0.f);
    }
}

__global__ void __launch_bounds__(256)
gradients_kernel(const float *__restrict__ img, int w, int h, int c,
    FitMatrix fit, float2 *__restrict__ grad, float4 *__restrict__ hess)
{
#pragma clang fp contract(off)
    // The luminance of a pixel is needed by its nine neighbours: a workgroup
    // (256 pixels of GRAD_ROWS rows) forms it once per pixel for the rows of its
    // windows in LDS -- the same float expression, so the same values.
    __shared__ float lum[GRAD_ROWS + 2][256 + 2];
    int const x0 = blockIdx.x * blockDim.x;
    int const t = threadIdx.x;
    int const y0 = blockIdx.y * GRAD_ROWS;
    // (padding workgroups of the row, see the launch: nothing to do)
    if (x0 >= w)
        return;
    for (int col = t; col < 256 + 2; col += 256) {
        int const gx = min(max(x0 + col - 1, 0), w - 1);
#pragma unroll
        for (int row = 0; row < GRAD_ROWS + 2; ++row) {
            int const gy = min(max(y0 + row - 1, 0), h - 1);
            const float *px = img + ((size_t)gy * w + gx) * c;
            lum[row][col] = c >= 3 ? px[0] * 0.21f + px[1] * 0.72f + px[2] * 0.07f : px[0];
        }
    }
    __syncthreads();
    int const x = x0 + t;
    if (x >= w)
        return;
#pragma unroll 1
    for (int r = 0; r < GRAD_ROWS; ++r) {
        int const y = y0 + r;
        if (y >= h)
            break;
        size_t const p = (size_t)y * w + x;
        float2 g = make_float2(0.f, 0.f);
        float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
            double v[9];
            int k = 0;
            for (int a = -1; a < 2; ++a)
                for (int b = -1; b < 2; ++b)
                    v[k++] = lum[r + b + 1][t + 1 + a];
            quadratic_fit(fit, v, hess != nullptr, &g, &hs);
        }
        grad[p] = g;
        if (hess != nullptr)
            hess[p] = hs;
    }
}

// The y pass of the blur, the luminance and the quadratic fit in ONE kernel
// (round 6): blur_y_kernel wrote the blurred image (25 MB at 1920 x 1080 x 3)
// only for gradients_kernel to read it back -- nothing else uses it.  A
// workgroup owns FUSE_COLS x FUSE_ROWS (254 x 9) output pixels; its 256 threads first
// form the blurred values of the (FUSE_ROWS + 2) x 256 pixels its windows span
// -- one thread per ELEMENT column (pixel x channel: the loads of a wave are
// contiguous), FUSE_ROWS + 2 output rows per thread, every loaded row feeding
// all the outputs it is a tap of in ascending tap order, exactly blur_y_kernel's
// operations with BLUR_ROWS = FUSE_ROWS + 2 -- into LDS, then the luminance per
// pixel (gradients_kernel's float expression), then the fit.  Same operands,
// same order, same bits: tests/test_gpu_parity.py compares both forms with the
// oracle and with each other at 1920 x 1080.  SMVS_SCALE_FUSED=0: the two
// kernels (A/B).
// nine rows per tile: 960 tiles at 1920 x 1080, which the chip holds at once (four
// workgroups of 33 KB per CU); with eight rows the 1,080 tiles ran as a full round
// and a tail (2.39 against 2.34 ms per view, SMVS_FUSE_ROWS=8)
constexpr int FUSE_ROWS_DEFAULT = 9;
constexpr int FUSE_COLS = 254;          // + the two halo columns = 256 pixels = the workgroup

template <int C, int KS, int FUSE_ROWS>
__global__ void __launch_bounds__(256)
blur_y_gradients_kernel(const float *__restrict__ in, int w, int h, BlurTaps taps,
    FitMatrix fit, float2 *__restrict__ grad, float4 *__restrict__ hess)
{
#pragma clang fp contract(off)
    constexpr int R = FUSE_ROWS + 2;
    // (the luminances take the place of the blurred values they are formed from:
    // read, barrier, written -- 3 KB per tile row instead of 4)
    __shared__ float blurred[R][256 * C];
    float (*const lum)[256] = reinterpret_cast<float (*)[256]>(&blurred[0][0]);
    int const x0 = (int)blockIdx.x * FUSE_COLS;     // first output pixel; tile column j is pixel x0 - 1 + j
    int const y0 = (int)blockIdx.y * FUSE_ROWS;
    int const t = (int)threadIdx.x;
    // (padding workgroups of a grid row, see the launch)
    if (x0 >= w)
        return;
    int const row_len = w * C;
    // the tap rows of the tile stay inside the image: no clamps along y (wave-uniform)
    bool const inner = y0 - 1 - KS >= 0 && y0 + FUSE_ROWS + KS <= h - 1;
#pragma unroll 1
    for (int part = 0; part < C; ++part) {
        int const ec = part * 256 + t;              // element column of the tile
        int const px = ec / C, ch = ec - px * C;
        int const gx = min(max(x0 - 1 + px, 0), w - 1);
        const float *col = in + (size_t)gx * C + ch;
        float av[R];
#pragma unroll
        for (int j = 0; j < R; ++j)
            av[j] = 0.0f;
        // tile row j (image row y0 - 1 + j) takes input row y0 - 1 - KS + i as
        // its tap k = i - KS - j; at the image border the input row is clamped
        // like blur_y_kernel's.  (A tile row outside the image then holds
        // something else than the reference's clamped row -- only the zero
        // gradients of the image's rim look at it.)
        constexpr int N = 2 * KS + R;
        constexpr int GROUP = 16;
        int const first = y0 - 1 - KS;
#pragma unroll
        for (int g = 0; g < N; g += GROUP) {
            float v[GROUP];
#pragma unroll
            for (int i = 0; i < GROUP; ++i)
                if (g + i < N) {
                    int const yy = inner ? first + g + i
                        : min(max(first + g + i, 0), h - 1);
                    v[i] = col[(size_t)yy * row_len];
                }
#pragma unroll
            for (int i = 0; i < GROUP; ++i)
                if (g + i < N) {
                    int const tt = g + i - KS;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        int const k = tt - j;     // tap of tile row j
                        if (k >= -KS && k <= KS)
                            av[j] += v[i] * taps.k[k < 0 ? -k : k];
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
            blurred[j][ec] = av[j] / taps.wsum;
    }
    __syncthreads();
    float l[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float *px = &blurred[j][t * C];
        l[j] = C >= 3 ? px[0] * 0.21f + px[1] * 0.72f + px[2] * 0.07f : px[0];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j)
        lum[j][t] = l[j];
    __syncthreads();
    int const x = x0 + t;
    if (t >= FUSE_COLS || x >= w)
        return;
#pragma unroll 1
    for (int r = 0; r < FUSE_ROWS; ++r) {
        int const y = y0 + r;
        if (y >= h)
            break;
        size_t const p = (size_t)y * w + x;
        float2 g = make_float2(0.f, 0.f);
        float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
            double v[9];
            int k = 0;
            for (int a = -1; a < 2; ++a)
                for (int b = -1; b < 2; ++b)
                    v[k++] = lum[r + b + 1][t + 1 + a];
            quadratic_fit(fit, v, hess != nullptr, &g, &hs);
        }
        grad[p] = g;
        if (hess != nullptr)
            hess[p] = hs;
    }
}

__global__ void __launch_bounds__(256)
compact_hessian_kernel(const float4 *__restrict__ src, float *__restrict__ dst,
    size_t count)
{
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    float4 const v = src[i];
    dst[3 * i] = v.x;
    dst[3 * i + 1] = v.y;
    dst[3 * i + 2] = v.z;
}

static FitMatrix
quadratic_fit_matrix(void)
{
    // least-squares fit of c_xx a^2 + c_yy b^2 + c_xy ab + c_x a + c_y b + c_0
    // to the window (a outer, b inner): rows xx, yy, xy, x, y, 1
    FitMatrix f;
    int col = 0;
    for (int a = -1; a <= 1; ++a)
        for (int b = -1; b <= 1; ++b, ++col) {
            f.m[0][col] = a == 0 ? -1.0 / 3.0 : 1.0 / 6.0;
            f.m[1][col] = b == 0 ? -1.0 / 3.0 : 1.0 / 6.0;
            f.m[2][col] = (double)(a * b) / 4.0;
            f.m[3][col] = (double)a / 6.0;
            f.m[4][col] = (double)b / 6.0;
            f.m[5][col] = (a == 0 && b == 0) ? 5.0 / 9.0
                : ((a == 0 || b == 0) ? 2.0 / 9.0 : -1.0 / 9.0);
        }
    return f;
}

// blur_x + the fused y pass / luminance / fit (the half widths of the scales
// 0 .. 5 only: any other takes the separate kernels).  false: not launched.
template <int KS, int ROWS>
static bool
launch_blur_gradients_rows(hipStream_t stream, const float *in, float *tmp, int w, int h,
    int c, BlurTaps const &taps, FitMatrix const &fit, float2 *grad, float4 *hess)
{
    launch_blur_x<KS>(stream, in, tmp, w, h, c, taps);
    // (a grid row is a multiple of 8 workgroups: a column of tiles, whose tap rows
    // overlap, stays on ONE XCD and its L2 -- launch_blur_ks)
    dim3 const grid((((unsigned)w + FUSE_COLS - 1) / FUSE_COLS + 7u) & ~7u,
        ((unsigned)h + ROWS - 1) / ROWS);
    if (c == 1)
        hipLaunchKernelGGL((blur_y_gradients_kernel<1, KS, ROWS>), grid, dim3(256), 0, stream,
            tmp, w, h, taps, fit, grad, hess);
    else
        hipLaunchKernelGGL((blur_y_gradients_kernel<3, KS, ROWS>), grid, dim3(256), 0, stream,
            tmp, w, h, taps, fit, grad, hess);
    return true;
}

template <int KS>
static bool
launch_blur_gradients_ks(hipStream_t stream, const float *in, float *tmp, int w, int h,
    int c, BlurTaps const &taps, FitMatrix const &fit, float2 *grad, float4 *hess)
{
    const char *e = std::getenv("SMVS_FUSE_ROWS");
    if (e != nullptr && std::atoi(e) == 8)
        return launch_blur_gradients_rows<KS, 8>(stream, in, tmp, w, h, c, taps, fit, grad, hess);
    return launch_blur_gradients_rows<KS, FUSE_ROWS_DEFAULT>(stream, in, tmp, w, h, c, taps, fit,
        grad, hess);
}

static bool
launch_blur_gradients(hipStream_t stream, const float *in, float *tmp, int w, int h, int c,
    BlurTaps const &taps, FitMatrix const &fit, float2 *grad, float4 *hess)
{
    // (read per call, a few times per view: a test compares both forms in one process)
    const char *e = std::getenv("SMVS_SCALE_FUSED");
    bool const fused = !(e != nullptr && e[0] == '0');
    if (!fused || (c != 1 && c != 3))
        return false;
    switch (taps.ks) {
    case 1: return launch_blur_gradients_ks<1>(stream, in, tmp, w, h, c, taps, fit, grad, hess);
    case 2: return launch_blur_gradients_ks<2>(stream, in, tmp, w, h, c, taps, fit, grad, hess);
    case 4: return launch_blur_gradients_ks<4>(stream, in, tmp, w, h, c, taps, fit, grad, hess);
    case 7: return launch_blur_gradients_ks<7>(stream, in, tmp, w, h, c, taps, fit, grad, hess);
    case 12: return launch_blur_gradients_ks<12>(stream, in, tmp, w, h, c, taps, fit, grad, hess);
    // (half width 23, scale 6: the two rows a tile forms beyond its eight cost more
    // arithmetic than the round trip of the blurred image -- 0.674 against 0.663 ms
    // for nine views, profiles/r6_scale_space_cost.txt -- so it keeps the two kernels)
    default: return false;
    }
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_ctx_upload_image(smvs_ctx *ctx, int view, int width, int height,
    int channels, const uint8_t *bytes)
{
    SMVS_REQUIRE(ctx && bytes, "null argument");
    SMVS_REQUIRE(view >= -1 && view < ctx->n_subs, "view index out of range");
    SMVS_REQUIRE(width > 2 && height > 2 && (channels == 1 || channels == 3),
        "bad image");
    if (view == -1)
        SMVS_REQUIRE(width == ctx->width && height == ctx->height,
            "main image size differs from the context");
    SMVS_HIP_CHECK(set_device(ctx->device));
    smvs_ctx::ViewImage &vi = ctx->images[view + 1];
    size_t const n = (size_t)width * height * channels;
    int rc;
    if (vi.w != width || vi.h != height || vi.c != channels || !vi.data) {
        if ((rc = device_alloc(&vi.data, n)) != SMVS_OK)
            return rc;
        vi.w = width;
        vi.h = height;
        vi.c = channels;
    }
    ctx->image_ok &= ~(1u << (view + 1));
    ctx->image_pending &= ~(1u << (view + 1));   // (this upload replaces one on its way)
    // (device staging owned by the context: no allocation per image; the
    // bytes cross PCIe from pinned memory)
    if (ctx->byte_stage_cap < n) {
        if ((rc = device_alloc(&ctx->byte_stage, n)) != SMVS_OK) {
            ctx->byte_stage_cap = 0;
            return rc;
        }
        ctx->byte_stage_cap = n;
    }
    uint8_t *staging = ctx->byte_stage;
    if ((rc = ctx_upload(ctx, staging, bytes, n)) != SMVS_OK)
        return rc;
    hipLaunchKernelGGL(byte_to_float16_kernel, dim3(byte_to_float_blocks(n)),
        dim3(256), 0, ctx->stream, staging, vi.data, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("smvs_ctx_upload_image: %s", hipGetErrorString(e));
        return SMVS_ERR_HIP;
    }
    ctx->image_ok |= 1u << (view + 1);
    return SMVS_OK;
}

// StereoView::create's byte_to_float_image without waiting for it: when `bytes`
// is page-locked memory (smvs_pinned_alloc) the DMA is enqueued on the
// context's copy stream into the view's own staging buffer and the call
// returns; the conversion follows on the context's stream where the image is
// first needed (ctx_materialise_images), behind an event -- the nine uploads
// of a view overlap each other's conversion and the first scale's blur and
// gradient kernels instead of alternating with them (round 5: 9 x (138 us of
// DMA + 17 us of kernel), the GPU's compute idle for 1.3 ms per view).
// `bytes` must stay valid and unchanged until the context has been
// synchronised.  Pageable memory: the same as smvs_ctx_upload_image.
extern "C" int
smvs_ctx_upload_image_async(smvs_ctx *ctx, int view, int width, int height,
    int channels, const uint8_t *bytes)
{
    SMVS_REQUIRE(ctx && bytes, "null argument");
    SMVS_REQUIRE(view >= -1 && view < ctx->n_subs, "view index out of range");
    SMVS_REQUIRE(width > 2 && height > 2 && (channels == 1 || channels == 3),
        "bad image");
    size_t const n = (size_t)width * height * channels;
    if (n < ((size_t)1 << 20) || !host_pointer_is_pinned(bytes))
        return smvs_ctx_upload_image(ctx, view, width, height, channels, bytes);
    if (view == -1)
        SMVS_REQUIRE(width == ctx->width && height == ctx->height,
            "main image size differs from the context");
    SMVS_HIP_CHECK(set_device(ctx->device));
    int const v = view + 1;
    smvs_ctx::ViewImage &vi = ctx->images[v];
    int rc;
    if (vi.w != width || vi.h != height || vi.c != channels || !vi.data) {
        if ((rc = device_alloc(&vi.data, n)) != SMVS_OK)
            return rc;
        vi.w = width;
        vi.h = height;
        vi.c = channels;
    }
    bool const had_image = ((ctx->image_ok >> v) & 1u) != 0u;
    ctx->image_ok &= ~(1u << v);
    ctx->image_pending &= ~(1u << v);
    ctx->image_direct &= ~(1u << v);
    // SMVS_UPLOAD_STREAM=same: the copies on the context's own stream (A/B)
    static bool const same_stream = [] {
        const char *e = std::getenv("SMVS_UPLOAD_STREAM");
        return e != nullptr && e[0] == 's';
    }();
    auto const ensure_stage = [&]() -> int {
        if (ctx->upload_stage_cap[v] < n) {
            ctx->upload_stage_cap[v] = 0;
            int const arc = device_alloc(&ctx->upload_stage[v], n);
            if (arc != SMVS_OK)
                return arc;
            ctx->upload_stage_cap[v] = n;
        }
        return SMVS_OK;
    };
    if (same_stream) {
        if ((rc = ensure_stage()) != SMVS_OK)
            return rc;
        SMVS_HIP_CHECK(hipMemcpyAsync(ctx->upload_stage[v], bytes, n, hipMemcpyHostToDevice,
            ctx->stream));
        hipLaunchKernelGGL(byte_to_float16_kernel, dim3(byte_to_float_blocks(n)),
            dim3(256), 0, ctx->stream, ctx->upload_stage[v], vi.data, n);
        SMVS_HIP_CHECK(hipGetLastError());
        ctx->image_ok |= 1u << v;
        return SMVS_OK;
    }
    // SMVS_UPLOAD_STREAM=kernel (round 6, measured and left off): the images
    // converted straight out of the caller's page-locked memory by small-grid
    // kernels (the bytes cross the bus as the kernel reads them) -- the main image
    // on the context's stream, the others on the copy stream with their events --
    // instead of DMAs into staging buffers + a conversion each.  A warm optimize()
    // with SGM 16.2-16.4 against 15.9-16.1 ms, 69-70 against 70-72 views/s with
    // eight views in flight: the DMA engines cost the compute queues nothing.
    static bool const by_dma = [] {
        const char *e = std::getenv("SMVS_UPLOAD_STREAM");
        return !(e != nullptr && e[0] == 'k');
    }();
    auto const convert_from_host = [&](hipStream_t stream) {
        if ((reinterpret_cast<uintptr_t>(bytes) & 15u) == 0u)
            hipLaunchKernelGGL(host_bytes_to_float_kernel, dim3(UPLOAD_BLOCKS), dim3(256), 0,
                stream, bytes, vi.data, n);
        else
            hipLaunchKernelGGL(byte_to_float_kernel, dim3((unsigned)((n + 255) / 256)),
                dim3(256), 0, stream, bytes, vi.data, n);
    };
    if (v == 0 && !by_dma) {
        convert_from_host(ctx->stream);
        SMVS_HIP_CHECK(hipGetLastError());
        ctx->image_ok |= 1u << v;
        return SMVS_OK;
    }
    if (ctx->copy_stream == nullptr)
        SMVS_HIP_CHECK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (ctx->image_ready[v] == nullptr)
        SMVS_HIP_CHECK(hipEventCreateWithFlags(&ctx->image_ready[v], hipEventDisableTiming));
    if (!by_dma) {
        // (an image this view had before may still be read by kernels of the
        // context's stream: the new one is then written behind them)
        if (had_image) {
            SMVS_HIP_CHECK(hipEventRecord(ctx->image_ready[v], ctx->stream));
            SMVS_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, ctx->image_ready[v], 0));
        }
        convert_from_host(ctx->copy_stream);
        SMVS_HIP_CHECK(hipGetLastError());
        SMVS_HIP_CHECK(hipEventRecord(ctx->image_ready[v], ctx->copy_stream));
        ctx->image_pending |= 1u << v;
        ctx->image_direct |= 1u << v;
        ctx->image_ok |= 1u << v;   // (every consumer waits for what it reads)
        return SMVS_OK;
    }
    if ((rc = ensure_stage()) != SMVS_OK)
        return rc;
    // (a conversion still reading this staging buffer: only a second upload
    // of the same view before the first was read -- the copy then waits for the
    // context's stream; the common case has nothing to wait for and must not be
    // chained behind whatever the context's stream is doing)
    if (ctx->upload_stage_busy & (1u << v)) {
        SMVS_HIP_CHECK(hipEventRecord(ctx->image_ready[v], ctx->stream));
        SMVS_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, ctx->image_ready[v], 0));
    }
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->upload_stage[v], bytes, n, hipMemcpyHostToDevice,
        ctx->copy_stream));
    SMVS_HIP_CHECK(hipEventRecord(ctx->image_ready[v], ctx->copy_stream));
    ctx->image_pending |= 1u << v;
    ctx->image_ok |= 1u << v;   // (every consumer materialises what it reads)
    return SMVS_OK;
}

int
smvs_hip::ctx_materialise_images(smvs_ctx *ctx, uint32_t views)
{
    uint32_t const todo = ctx->image_pending & views;
    for (int v = 0; v <= SMVS_MAX_SUBS; ++v) {
        if (!((todo >> v) & 1u))
            continue;
        smvs_ctx::ViewImage const &vi = ctx->images[v];
        size_t const n = (size_t)vi.w * vi.h * vi.c;
        SMVS_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->image_ready[v], 0));
        ctx->image_pending &= ~(1u << v);
        if ((ctx->image_direct >> v) & 1u) {
            ctx->image_direct &= ~(1u << v);     // (converted by the upload itself)
            continue;
        }
        hipLaunchKernelGGL(byte_to_float16_kernel, dim3(byte_to_float_blocks(n)),
            dim3(256), 0, ctx->stream, ctx->upload_stage[v], vi.data, n);
        SMVS_HIP_CHECK(hipGetLastError());
        ctx->upload_stage_busy |= 1u << v;   // until the context's stream has been waited for
    }
    return SMVS_OK;
}

extern "C" int
smvs_ctx_set_scale(smvs_ctx *ctx, int scale)
{
    SMVS_REQUIRE(ctx != nullptr, "null context");
    SMVS_REQUIRE(scale >= 0 && scale <= 10, "scale out of range");
    for (int v = 0; v <= ctx->n_subs; ++v)
        if (!((ctx->image_ok >> v) & 1u)) {
            set_error("smvs_ctx_set_scale: view %d has no image", v - 1);
            return SMVS_ERR_STATE;
        }
    SMVS_HIP_CHECK(set_device(ctx->device));
    // stereo_view.cc:29-31; the weights use the host's expf like the reference
    double const sigma_d = 0.12 * std::pow(2.0, scale) + 0.2;
    float const sigma = (float)sigma_d;
    int const ks = (int)std::ceil(sigma * 2.884f);
    SMVS_REQUIRE(ks + 1 <= 64, "blur kernel too wide");
    BlurTaps taps = {};
    taps.ks = ks;
    for (int i = 0; i <= ks; ++i)
        taps.k[i] = std::exp(-((float)i * (float)i) / (2.0f * sigma * sigma));
    {
        // (volatile: the sum must be rounded to float after every addition,
        // like Accum<float>)
        volatile float wsum = 0.0f;
        for (int i = -ks; i <= ks; ++i)
            wsum = wsum + taps.k[i < 0 ? -i : i];
        taps.wsum = wsum;
    }
    int rc;
    hipError_t e = hipSuccess;
    FitMatrix const fit = quadratic_fit_matrix();
    bool const blur = !(std::fabs(sigma) < 0.1f);

    for (int v = 0; v <= ctx->n_subs && e == hipSuccess; ++v) {
        // (an image still on its way: its conversion goes here, so that this
        // view's blur and gradients run while the next views' DMA is in flight)
        if ((rc = ctx_materialise_images(ctx, 1u << v)) != SMVS_OK)
            return rc;
        smvs_ctx::ViewImage const &vi = ctx->images[v];
        size_t const n = (size_t)vi.w * vi.h * vi.c;
        if (n > ctx->blur_cap) {
            if ((rc = device_alloc(&ctx->blur_tmp[0], n)) != SMVS_OK
                || (rc = device_alloc(&ctx->blur_tmp[1], n)) != SMVS_OK)
                return rc;
            ctx->blur_cap = n;
        }
        float2 *grad;
        float4 *hess;
        if (v == 0) {
            grad = ctx->main_grad;
            hess = nullptr;
        } else {
            SubPlanes &sp = ctx->subs[v - 1];
            size_t const npix = (size_t)vi.w * vi.h;
            if (sp.width != vi.w || sp.height != vi.h || sp.grad == nullptr
                || sp.hess == nullptr) {
                sp.width = sp.height = 0;   // absent until both planes exist
                if ((rc = device_alloc(&sp.grad, npix)) != SMVS_OK
                    || (rc = device_alloc(&sp.hess, npix)) != SMVS_OK) {
                    (void)device_alloc(&sp.grad, 0);
                    (void)device_alloc(&sp.hess, 0);
                    return rc;
                }
                sp.width = vi.w;
                sp.height = vi.h;
            }
            grad = sp.grad;
            hess = sp.hess;
            ctx->planes_ok |= 1u << (v - 1);
        }
        const float *src = vi.data;
        bool planes_done = false;
        if (blur) {
            ScopedKernelTimer timer(ctx, SMVS_K_MISC);
            // the y pass, the luminance and the fit in one kernel, or ...
            planes_done = launch_blur_gradients(ctx->stream, vi.data, ctx->blur_tmp[0], vi.w,
                vi.h, vi.c, taps, fit, grad, hess);
            if (!planes_done) {
                // ... the blurred image written out and read back
                launch_blur(ctx->stream, vi.data, ctx->blur_tmp[0], ctx->blur_tmp[1], vi.w,
                    vi.h, vi.c, taps);
                src = ctx->blur_tmp[1];
            }
        }
        if (!planes_done) {
            ScopedKernelTimer timer(ctx, SMVS_K_MISC);
            // (a row of the grid is a multiple of 8 workgroups, so that the three
            // rows a window spans meet in ONE XCD's L2 -- see launch_blur_ks; at
            // w = 1920 the 8 column blocks happen to be that already)
            hipLaunchKernelGGL(gradients_kernel,
                dim3((((unsigned)vi.w + 255u) / 256u + 7u) & ~7u,
                    ((unsigned)vi.h + GRAD_ROWS - 1) / GRAD_ROWS),
                dim3(256), 0, ctx->stream, src, vi.w, vi.h, vi.c, fit, grad, hess);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(ctx->subs_dev, ctx->subs,
            sizeof(SubPlanes) * SMVS_MAX_SUBS, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("smvs_ctx_set_scale: %s", hipGetErrorString(e));
        return SMVS_ERR_HIP;
    }
    return SMVS_OK;
}

extern "C" int
smvs_ctx_download_planes(smvs_ctx *ctx, int view, float *grad2, float *hess3)
{
    SMVS_REQUIRE(ctx && grad2, "null argument");
    SMVS_REQUIRE(view >= -1 && view < ctx->n_subs, "view index out of range");
    SMVS_HIP_CHECK(set_device(ctx->device));
    if (view == -1) {
        SMVS_REQUIRE(hess3 == nullptr, "the main view keeps no Hessian plane");
        SMVS_HIP_CHECK(hipMemcpyAsync(grad2, ctx->main_grad,
            (size_t)ctx->width * ctx->height * sizeof(float2),
            hipMemcpyDeviceToHost, ctx->stream));
        SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return SMVS_OK;
    }
    SubPlanes const &sp = ctx->subs[view];
    if (!((ctx->planes_ok >> view) & 1u)) {
        set_error("smvs_ctx_download_planes: view %d has no planes", view);
        return SMVS_ERR_STATE;
    }
    size_t const npix = (size_t)sp.width * sp.height;
    SMVS_HIP_CHECK(hipMemcpyAsync(grad2, sp.grad, npix * sizeof(float2),
        hipMemcpyDeviceToHost, ctx->stream));
    if (hess3 != nullptr) {
        if (ctx->stage_cap < npix * 3) {
            int rc = device_alloc(&ctx->stage, npix * 3);
            if (rc != SMVS_OK)
                return rc;
            ctx->stage_cap = npix * 3;
        }
        hipLaunchKernelGGL(compact_hessian_kernel,
            dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream,
            sp.hess, ctx->stage, npix);
        SMVS_HIP_CHECK(hipGetLastError());
        SMVS_HIP_CHECK(hipMemcpyAsync(hess3, ctx->stage,
            npix * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMVS_OK;
}

extern "C" int
smvs_ctx_upload_shading(smvs_ctx *ctx, const float *shading1,
    const float *shading_grad2)
{
    SMVS_REQUIRE(ctx && shading1 && shading_grad2, "null argument");
    SMVS_HIP_CHECK(set_device(ctx->device));
    size_t const npix = (size_t)ctx->width * ctx->height;
    int rc;
    if (ctx->main_shading == nullptr) {
        if ((rc = device_alloc(&ctx->main_shading, npix)) != SMVS_OK)
            return rc;
        if ((rc = device_alloc(&ctx->main_shading_grad, npix)) != SMVS_OK)
            return rc;
    }
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->main_shading, shading1,
        npix * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipMemcpyAsync(ctx->main_shading_grad, shading_grad2,
        npix * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->has_shading = true;
    return SMVS_OK;
}
