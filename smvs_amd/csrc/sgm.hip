// Census / plane-sweep cost volume, 8-path SGM aggregation, WTA and the joint
// bilateral upsample on gfx950.
//
// Replaces SGMStereo::run_sgm (reference: lib/sgm_stereo.cc:98-124):
// census_filter (:126-148), warped_neighbors_for_depth (:150-190),
// create_cost_volume (:192-244), aggregate_sgm_costs (:429-667, the SSE
// branch with constant penalty2, :361-406), depth_from_sgm_volume (:274-306);
// and DepthOptimizer::depthmap_bilateral_filter (lib/depth_optimizer.cc:957-1004).
//
// Integer path: results are bit-exact with the reference semantics.  The
// float warp is evaluated in the reference's operation order with FMA
// contraction off.  Volumes are [y][x][d], d fastest (sgm_stereo.cc:436-437);
// the cost volume is kept as u8 (values <= 255), S as u16.
//
// Aggregation: every path direction is an independent 1-D recurrence along a
// row, a column or a diagonal line of the image, so one wavefront walks one
// line (lane l owns planes 2l, 2l+1; neighbours and the minimum by DPP), with
// the loads of the next pixels issued ahead of the dependent chain.  With an
// even plane count all eight directions run in ONE launch and add into S with
// atomics on packed u16 pairs (integer adds commute: bit-exact); odd plane
// counts take one launch per direction.  Each path reads C once and
// read-modify-writes S once.
#include "common.h"
#include <type_traits>

#include <utility>

#include <mutex>

#include <cmath>
#include <vector>

namespace smvs_hip {

// ------------------------------------------------------------------ census
// sgm_stereo.cc:126-148: 9x7 window, x in [4, w-5), y in [3, h-4); bit = centre
// < neighbour, MSB first in (i outer, j inner) order; centre 0 -> census 0.
__global__ void __launch_bounds__(256)
census_main_kernel(const uint8_t *__restrict__ img, int w, int h,
    unsigned long long *__restrict__ out)
{
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= w)
        return;
    unsigned long long census = 0;
    if (x >= 4 && x < w - 5 && y >= 3 && y < h - 4) {
        uint8_t const thr = img[(size_t)y * w + x];
        if (thr != 0) {
            for (int i = x - 4; i < x + 5; ++i)
                for (int j = y - 3; j < y + 4; ++j) {
                    census <<= 1;
                    if (thr < img[(size_t)j * w + i])
                        census |= 1ull;
                }
        }
    }
    out[(size_t)y * w + x] = census;
}

struct WarpArgs {
    const uint8_t *neighbor;
    int nw, nh;
    float M[9], t[3];
    const float *depths;
    int D, w, h;
    uint8_t *warped;   // [h][w][D]
};

// sgm_stereo.cc:150-190.  One wavefront = 64 pixels of a row x 64 planes: a
// lane keeps its pixel's M p (plane independent) and walks the planes, so the
// 64 samples of an instruction are neighbours in the neighbour image; the
// bytes go through an LDS tile [pixel][plane] and leave as 64 contiguous
// bytes per pixel.  (One thread per (pixel, plane) spent twice the
// instructions: M p, the plane's depth and the address arithmetic per sample.)
constexpr int WARP_TILE = 64;            // pixels and planes per wave
constexpr int WARP_PITCH = WARP_TILE + 4;   // bytes per pixel row of the tile (17 dwords: no bank conflicts)

__global__ void __launch_bounds__(WARP_TILE)
warp_kernel(WarpArgs A)
{
#pragma clang fp contract(off)
    __shared__ uint8_t tile[WARP_TILE * WARP_PITCH];
    int const lane = (int)threadIdx.x;
    int const x0 = (int)blockIdx.x * WARP_TILE;
    int const x = x0 + lane;
    int const y = (int)blockIdx.y;
    int const dbase = (int)blockIdx.z * WARP_TILE;
    int const nd = min(WARP_TILE, A.D - dbase);
    if (x < A.w) {
        float const px = 0.5f + (float)x, py = 0.5f + (float)y;
        float tp[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float s = 0.0f;
            s += A.M[3 * r + 0] * px;
            s += A.M[3 * r + 1] * py;
            s += A.M[3 * r + 2] * 1.f;
            tp[r] = s;
        }
        float const xmax = (float)(A.nw - 1), ymax = (float)(A.nh - 1);
#pragma unroll 4
        for (int dd = 0; dd < nd; ++dd) {
            float const depth = A.depths[dbase + dd];
            float p0 = tp[0] * depth + A.t[0];
            float p1 = tp[1] * depth + A.t[1];
            float const p2 = tp[2] * depth + A.t[2];
            uint8_t out = 0;
            if (!(p2 < 0)) {
                p0 /= p2;
                p1 /= p2;
                p0 -= 0.5f;
                p1 -= 0.5f;
                if (!(p0 < 0 || p1 < 0 || p0 > xmax || p1 > ymax)) {
                    // mve::Image<uint8_t>::linear_at [MVE-unverified]
                    float const fx = fmaxf(0.0f, fminf(xmax, p0));
                    float const fy = fmaxf(0.0f, fminf(ymax, p1));
                    int const ix = (int)fx, iy = (int)fy;
                    int const ix1 = min(ix + 1, A.nw - 1), iy1 = min(iy + 1, A.nh - 1);
                    float const w1 = fx - (float)ix, w0 = 1.0f - w1;
                    float const w3 = fy - (float)iy, w2 = 1.0f - w3;
                    const uint8_t *r0 = A.neighbor + (size_t)iy * A.nw;
                    const uint8_t *r1 = A.neighbor + (size_t)iy1 * A.nw;
                    float const v1 = (float)r0[ix];
                    float const v2 = (float)r0[ix1];
                    float const v3 = (float)r1[ix];
                    float const v4 = (float)r1[ix1];
                    out = (uint8_t)(v1 * (w0 * w2) + v2 * (w1 * w2) + v3 * (w0 * w3)
                        + v4 * (w1 * w3) + 0.5f);
                }
            }
            tile[lane * WARP_PITCH + dd] = out;
        }
    }
    __syncthreads();
    // 16 lanes x 4 bytes per pixel, 4 pixels per sweep
    bool const quads = (A.D & 3) == 0;
    for (int idx = lane; idx < WARP_TILE * (WARP_TILE / 4); idx += WARP_TILE) {
        int const pix = idx >> 4, q = idx & 15;
        int const gx = x0 + pix, dd = 4 * q;
        if (gx >= A.w || dd >= nd)
            continue;
        size_t const o = ((size_t)y * A.w + gx) * A.D + dbase + dd;
        const uint8_t *src = tile + pix * WARP_PITCH + dd;
        if (quads) {
            *reinterpret_cast<uint32_t *>(A.warped + o) = *reinterpret_cast<const uint32_t *>(src);
        } else {
            for (int k = 0; k < 4 && dd + k < nd; ++k)
                A.warped[o + k] = src[k];
        }
    }
}

// ---------------------------------------------------------------------------
// The same cost volume with TWO planes per lane in the two 16-bit halves of a
// register and the census window kept in registers (round 5).
//
// cost_tiled_kernel below spends 3 vector instructions and one LDS byte read
// per census bit and plane (63 x 3 x 66 M: the ~410 us it takes).  Two
// observations:
//   * the Hamming distance needs no census word: with the main view's bit m_k
//     (the same for every plane of a pixel, so it lives in scalar registers)
//         popcount(census_warped ^ census_main) = sum_k [thr < v_k] ^ m_k
//                                               = popcount(m) + sum_k s_k [thr < v_k],
//     s_k = +1 where m_k = 0 and -1 where m_k = 1.  Per bit and PAIR of planes:
//     a saturating packed subtraction (v_k - thr, zero unless thr < v_k), a
//     packed minimum with 1, a packed multiply-add with the scalar s_k --
//     three instructions for two planes instead of three for one;
//   * the 9 x 7 windows of neighbouring pixels share eight of their nine
//     columns: a wave that walks a row keeps the window in 63 registers and
//     reads the ONE new column per pixel, 7 LDS reads instead of 63.
// Same tile, same bits (tests/test_gpu_parity.py, test_sgm_bit_exact: cost
// volume array_equal with the oracle); plane counts that are not a multiple
// of four keep the kernel below.
constexpr int CP_W = 16, CP_H = 8, CP_D = 128;

// (written as instructions: from `min(sub_sat(v, thr), 1)` on a 2 x u16 vector
// type the compiler builds compares and selects per half, five instructions
// where these are two)
__device__ __forceinline__ uint32_t
pk_sub_sat_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t
pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * s + c per half, s in a scalar register
__device__ __forceinline__ uint32_t
pk_mad_u16(uint32_t a, uint32_t s, uint32_t c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(s), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t
pk_mad_u16_vvv(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// f(integral_constant<0>), f(integral_constant<1>), ... in order
template <typename F, int... I>
__device__ __forceinline__ void
for_each_index(F &f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>()), ...);
}

__global__ void __launch_bounds__(256)
cost_packed_kernel(const uint8_t *__restrict__ warped,
    const unsigned long long *__restrict__ main_census, int w, int h, int D,
    uint8_t *__restrict__ cost, int xcd_bands)
{
    constexpr int TW = CP_W + 8;
    __shared__ uint8_t tile[(CP_H + 6) * TW * CP_D];
    int const tiles_x = (w + CP_W - 1) / CP_W;
    // Tiles are dealt to the XCDs in contiguous bands (round 6): a tile stages a
    // (16 + 8) x (8 + 6) window, 2.6 x its own pixels, and with neighbouring tiles
    // on different XCDs (workgroups go round robin) that halo came from HBM every
    // time: 178 MB read per launch for a 66 MB volume (profiles/r6_sgm_counters.txt).
    // The launch is padded to eight bands of equal length.
    unsigned const band = gridDim.x >> 3;
    unsigned const tile_id = xcd_bands ? (blockIdx.x & 7u) * band + (blockIdx.x >> 3) : blockIdx.x;
    if (tile_id >= (unsigned)(tiles_x * ((h + CP_H - 1) / CP_H)))
        return;
    int const x0 = (int)(tile_id % (unsigned)tiles_x) * CP_W;
    int const y0 = (int)(tile_id / (unsigned)tiles_x) * CP_H;
    int const dbase = blockIdx.y * CP_D;
    int const tid = threadIdx.x;

    // stage: 4 planes (one u32) per thread and position (D is a multiple of 4)
    for (int idx = tid; idx < (CP_H + 6) * TW * (CP_D / 4); idx += 256) {
        int const pos = idx / (CP_D / 4), q = idx - pos * (CP_D / 4);
        int const ty = pos / TW, tx = pos - ty * TW;
        int const gx = x0 - 4 + tx, gy = y0 - 3 + ty;
        int const d4 = dbase + 4 * q;
        uint32_t v = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h && d4 < D)
            v = *reinterpret_cast<const uint32_t *>(warped + ((size_t)gy * w + gx) * D + d4);
        *reinterpret_cast<uint32_t *>(tile + (size_t)pos * CP_D + 4 * q) = v;
    }
    __syncthreads();

    int const lane = tid & 63, wave = tid >> 6;
    int const d0 = dbase + 2 * lane;            // this lane's planes: d0, d0 + 1
    uint32_t const one = 0x00010001u, c255 = 0x00FF00FFu;
    for (int py = wave; py < CP_H; py += 4) {
        int const y = y0 + py;
        if (y >= h)
            break;
        // column c of the tile, rows py .. py + 6: this lane's two planes, one
        // per 16-bit half
        auto load_column = [&](int c, uint32_t (&col)[7]) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                uint32_t const raw = *reinterpret_cast<const uint16_t *>(
                    tile + ((py + j) * TW + c) * CP_D + 2 * lane);
                col[j] = (raw & 0xFFu) | ((raw & 0xFF00u) << 8);
            }
        };
        uint32_t win[9][7];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            load_column(i, win[i]);
        // one pixel of the row; PX is a template argument so that every index
        // into the window is a constant and the window stays in registers (a
        // `#pragma unroll` of a loop this long is only partly honoured, and the
        // window then lives in scratch)
        auto pixel = [&](auto px_tag) {
            constexpr int px = decltype(px_tag)::value;
            int const x = x0 + px;
            if (x < w) {
                uint32_t const thr = win[(px + 4) % 9][3];
                unsigned long long const mc = main_census[(size_t)y * w + x];
                // the address is wave-uniform: the bits go to scalar registers
                uint32_t const mhi = __builtin_amdgcn_readfirstlane((uint32_t)(mc >> 32));
                uint32_t const mlo = __builtin_amdgcn_readfirstlane((uint32_t)mc);
                uint32_t const pc = (uint32_t)(__popc(mhi) + __popc(mlo));
                uint32_t c = pc | (pc << 16);
                if (x >= 4 && x < w - 5 && y >= 3 && y < h - 4) {
                    uint32_t cnt = 0u;
                    // sgm_stereo.cc:139-145: i outer, j inner, MSB first: bit k
                    // of the census is bit 62 - k of the 64-bit word
#pragma unroll
                    for (int k = 0; k < 63; ++k) {
                        int const i = k / 7, j = k - 7 * i;
                        if (i == 4 && j == 3)
                            continue;   // the centre: thr < thr never holds
                        uint32_t const v = win[(px + i) % 9][j];
                        uint32_t const lt = pk_min_u16(pk_sub_sat_u16(v, thr), one);
                        int const bit = 62 - k;
                        bool const m = ((bit >= 32 ? mhi >> (bit - 32) : mlo >> bit) & 1u) != 0u;
                        cnt = pk_mad_u16(lt, m ? 0xFFFFFFFFu : 0x00010001u, cnt);
                    }
                    // (mod 2^16 per half; the true value is 0 .. 63)
                    c = pk_mad_u16_vvv(cnt, one, c);
                }
                // a plane that was not warped here (thr = 0) costs 255:
                // c = (c - 255) * [thr > 0] + 255 per half
                uint32_t const warped_here = pk_min_u16(thr, one);
                uint32_t const cm = pk_mad_u16_vvv(c255, 0xFFFFFFFFu, c);     // c - 255
                c = pk_mad_u16_vvv(cm, warped_here, c255);
                if (d0 < D)
                    *reinterpret_cast<uint16_t *>(cost + ((size_t)y * w + x) * D + d0)
                        = (uint16_t)((c & 0xFFu) | ((c >> 8) & 0xFF00u));
            }
            // the column that leaves makes room for the one that enters
            if constexpr (px + 1 < CP_W)
                load_column(px + 9, win[px % 9]);
        };
        for_each_index(pixel, std::make_integer_sequence<int, CP_W>());
    }
}

// Cost volume (sgm_stereo.cc:192-244): census of the warped planes + Hamming
// distance to the main census.  A block stages a (16+8) x (8+6) pixel window of
// 64 planes in LDS (each warped byte is needed by 63 census windows), one
// wavefront walks pixels with its lanes along the plane axis, so LDS reads,
// global loads and the u8 cost stores are all 64 contiguous bytes.
constexpr int CT_W = 16, CT_H = 8, CT_D = 64;

__global__ void __launch_bounds__(256)
cost_tiled_kernel(const uint8_t *__restrict__ warped,
    const unsigned long long *__restrict__ main_census, int w, int h, int D,
    uint8_t *__restrict__ cost)
{
    __shared__ uint8_t tile[(CT_H + 6) * (CT_W + 8) * CT_D];
    int const tiles_x = (w + CT_W - 1) / CT_W;
    int const x0 = (blockIdx.x % tiles_x) * CT_W;
    int const y0 = (blockIdx.x / tiles_x) * CT_H;
    int const dbase = blockIdx.y * CT_D;
    int const tid = threadIdx.x;

    // stage: 4 planes (one u32) per thread and position
    bool const aligned = (D & 3) == 0;
    for (int idx = tid; idx < (CT_H + 6) * (CT_W + 8) * (CT_D / 4); idx += 256) {
        int const pos = idx / (CT_D / 4), q = idx - pos * (CT_D / 4);
        int const ty = pos / (CT_W + 8), tx = pos - ty * (CT_W + 8);
        int const gx = x0 - 4 + tx, gy = y0 - 3 + ty;
        int const d4 = dbase + 4 * q;
        uint32_t v = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h && d4 < D) {
            size_t const o = ((size_t)gy * w + gx) * D + d4;
            if (aligned)
                v = *reinterpret_cast<const uint32_t *>(warped + o);
            else
                for (int k = 0; k < 4 && d4 + k < D; ++k)
                    v |= (uint32_t)warped[o + k] << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(tile + (size_t)pos * CT_D + 4 * q) = v;
    }
    __syncthreads();

    int const lane = tid & 63, wave = tid >> 6;
    int const d = dbase + lane;
    for (int py = wave; py < CT_H; py += 4) {
        int const y = y0 + py;
        if (y >= h)
            break;
        for (int px = 0; px < CT_W; ++px) {
            int const x = x0 + px;
            if (x >= w)
                break;
            const uint8_t *centre = tile + ((py + 3) * (CT_W + 8) + px + 4) * CT_D + lane;
            uint32_t const thr = *centre;
            uint32_t c = 255;
            if (thr != 0) {
                uint32_t hi = 0, lo = 0;
                if (x >= 4 && x < w - 5 && y >= 3 && y < h - 4) {
                    // sgm_stereo.cc:139-145: i outer, j inner, MSB first
#pragma unroll
                    for (int k = 0; k < 63; ++k) {
                        int const i = k / 7, j = k - 7 * i;
                        uint32_t const v = tile[((py + j) * (CT_W + 8) + px + i) * CT_D + lane];
                        uint32_t const lt = thr < v ? 1u : 0u;
                        if (k < 31)
                            hi = (hi << 1) | lt;
                        else
                            lo = (lo << 1) | lt;
                    }
                }
                unsigned long long const mc = main_census[(size_t)y * w + x];
                c = __popc(hi ^ (uint32_t)(mc >> 32)) + __popc(lo ^ (uint32_t)mc);
            }
            if (d < D)
                cost[((size_t)y * w + x) * D + d] = (uint8_t)c;
        }
    }
}

// ------------------------------------------------------------- aggregation
struct PathArgs {
    const uint8_t *cost;
    uint16_t *sgm;
    int w, h, D;
    int dx, dy;        // direction of travel
    uint32_t p1, p2;
    int first;         // 1: S is written, not accumulated
    int last;          // 1: last path, fuse the winner-takes-all
    uint8_t *delta;    // all-paths kernel, DELTA form: eight [h][w][D] u8 volumes
    size_t vol;        // bytes of one of them
};

// Wave-wide unsigned minimum with DPP row operations (VALU latency instead of
// the LDS crossbar of ds_bpermute): prefix-min inside each row of 16 lanes,
// row_bcast:15 / row_bcast:31 to combine the rows, total in lane 63.
__device__ __forceinline__ uint32_t
wave_min_u32(uint32_t v)
{
    uint32_t const ident = 0xFFFFFFFFu;
#define SMVS_DPP(x, ctrl, rmask)                                             \
    (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)(x), ctrl, rmask, \
        0xf, false)
    v = min(v, SMVS_DPP(v, 0x111, 0xf));   // row_shr:1
    v = min(v, SMVS_DPP(v, 0x112, 0xf));   // row_shr:2
    v = min(v, SMVS_DPP(v, 0x114, 0xf));   // row_shr:4
    v = min(v, SMVS_DPP(v, 0x118, 0xf));   // row_shr:8
    v = min(v, SMVS_DPP(v, 0x142, 0xa));   // row_bcast:15 -> rows 1, 3
    v = min(v, SMVS_DPP(v, 0x143, 0xc));   // row_bcast:31 -> rows 2, 3
#undef SMVS_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// value of the previous / next lane (wave_shr:1 / wave_shl:1); lanes without
// a source get `fill`
__device__ __forceinline__ uint32_t
lane_prev(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf,
        0xf, false);
}

__device__ __forceinline__ uint32_t
lane_next(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf,
        0xf, false);
}

// One wavefront per line.  Lines: for a horizontal path the rows, for a
// vertical path the columns, for a diagonal path all diagonals that start on
// the entry row or the entry column.
//
// Seeding follows the reference exactly (sgm_stereo.cc:457-464, 511-534,
// 589-612): the first pixel of a line copies C and adds it to S; for a
// diagonal path the corner pixel that lies on both the entry row and the
// entry column is added twice.
__global__ void __launch_bounds__(64)
sgm_path_kernel(PathArgs A)
{
    int const lane = threadIdx.x;
    int const line = blockIdx.x;
    int const w = A.w, h = A.h, D = A.D;
    int x, y, len;
    int extra_seed = 0;
    if (A.dy == 0) {            // horizontal: one line per row
        if (line >= h)
            return;
        y = line;
        x = A.dx > 0 ? 0 : w - 1;
        len = w;
    } else if (A.dx == 0) {     // vertical: one line per column
        if (line >= w)
            return;
        x = line;
        y = A.dy > 0 ? 0 : h - 1;
        len = h;
    } else {                    // diagonal
        if (line >= w + h - 1)
            return;
        int const y_entry = A.dy > 0 ? 0 : h - 1;
        int const x_entry = A.dx > 0 ? 0 : w - 1;
        if (line < w) {         // starts on the entry row
            x = line;
            y = y_entry;
            if (x == x_entry)
                extra_seed = 1; // corner: row seed + column seed
        } else {                // starts on the entry column, off the corner
            int const k = line - w + 1;
            x = x_entry;
            y = A.dy > 0 ? k : h - 1 - k;
        }
        int const nx = A.dx > 0 ? w - x : x + 1;
        int const ny = A.dy > 0 ? h - y : y + 1;
        len = min(nx, ny);
    }

    int const d0 = 2 * lane, d1 = 2 * lane + 1;
    bool const ok0 = d0 < D, ok1 = d1 < D;
    uint32_t const BIG = 0xFFFFu;
    uint32_t prev0 = BIG, prev1 = BIG;

    for (int s = 0; s < len; ++s, x += A.dx, y += A.dy) {
        size_t const base = ((size_t)y * w + x) * D;
        uint32_t c0 = ok0 ? A.cost[base + d0] : 0u;
        uint32_t c1 = ok1 ? A.cost[base + d1] : 0u;
        uint32_t l0, l1;
        if (s == 0) {
            l0 = c0;
            l1 = c1;
        } else {
            uint32_t const mn = wave_min_u32(min(prev0, prev1));
            uint32_t const left = (uint32_t)__shfl_up((int)prev1, 1);
            uint32_t const right = (uint32_t)__shfl_down((int)prev0, 1);
            bool const has_left = lane > 0;
            bool const has_right = lane < 63 && d1 + 1 < D;
            uint32_t const far = (mn + A.p2) & 0xFFFFu;
            // u16 wrapping arithmetic of the SSE code (_mm_add_epi16)
            uint32_t u0 = prev0;
            u0 = min(u0, has_left ? ((left + A.p1) & 0xFFFFu) : BIG);
            u0 = min(u0, ok1 ? ((prev1 + A.p1) & 0xFFFFu) : BIG);
            u0 = min(u0, far);
            uint32_t u1 = prev1;
            u1 = min(u1, (prev0 + A.p1) & 0xFFFFu);
            u1 = min(u1, has_right ? ((right + A.p1) & 0xFFFFu) : BIG);
            u1 = min(u1, far);
            l0 = (c0 + u0 - mn) & 0xFFFFu;
            l1 = (c1 + u1 - mn) & 0xFFFFu;
        }
        uint32_t add0 = l0, add1 = l1;
        if (s == 0 && extra_seed) {
            add0 = (2 * c0) & 0xFFFFu;
            add1 = (2 * c1) & 0xFFFFu;
        }
        if (ok0) {
            uint32_t const old = A.first ? 0u : A.sgm[base + d0];
            A.sgm[base + d0] = (uint16_t)(old + add0);
        }
        if (ok1) {
            uint32_t const old = A.first ? 0u : A.sgm[base + d1];
            A.sgm[base + d1] = (uint16_t)(old + add1);
        }
        prev0 = ok0 ? l0 : BIG;
        prev1 = ok1 ? l1 : BIG;
    }
}

// Line geometry shared by both path kernels.  Seeding follows the reference
// exactly (sgm_stereo.cc:457-464, 511-534, 589-612): the first pixel of a
// line copies C and adds it to S; for a diagonal path the corner pixel that
// lies on both the entry row and the entry column is added twice (every
// entry-column pixel is the start of its own diagonal, so the reference's
// column seeding needs nothing else).
__device__ __forceinline__ bool
path_line(PathArgs const &A, int line, int *x, int *y, int *len, int *extra)
{
    int const w = A.w, h = A.h;
    *extra = 0;
    if (A.dy == 0) {
        if (line >= h)
            return false;
        *y = line;
        *x = A.dx > 0 ? 0 : w - 1;
        *len = w;
    } else if (A.dx == 0) {
        if (line >= w)
            return false;
        *x = line;
        *y = A.dy > 0 ? 0 : h - 1;
        *len = h;
    } else {
        if (line >= w + h - 1)
            return false;
        int const y_entry = A.dy > 0 ? 0 : h - 1;
        int const x_entry = A.dx > 0 ? 0 : w - 1;
        if (line < w) {
            *x = line;
            *y = y_entry;
            if (*x == x_entry)
                *extra = 1;
        } else {
            int const k = line - w + 1;
            *x = x_entry;
            *y = A.dy > 0 ? k : h - 1 - k;
        }
        int const nx = A.dx > 0 ? w - *x : *x + 1;
        int const ny = A.dy > 0 ? h - *y : *y + 1;
        *len = min(nx, ny);
    }
    return true;
}

// ---- two lines per wavefront, four planes per lane, packed 16-bit arithmetic
// (round 6; DELTA form, plane counts that are multiples of four up to 128) ----
// sgm_all_paths_kernel spends ~37 vector instructions per step of a line for
// 128 planes -- two per lane, every minimum and sum a 32-bit operation, and a
// quarter of them the minimum over the wave -- and the launch is bound by
// exactly those (profiles/r6_sgm_counters.txt: 54 % issuing, 0.33 of HBM).
// Here a lane holds FOUR planes as two u16 pairs (v_pk_add_u16 / v_pk_min_u16
// work on both halves), so 32 lanes cover a line and a wave walks TWO adjacent
// lines of one direction: the step's instructions are shared by both, the
// minimum over a line is five DPP steps over a half wave instead of six over a
// whole one.  Same integers as sgm_all_paths_kernel<K, true> (every value stays
// below 2^16: L <= 255 + P2, BIG + P1 + P2 < 65536), hence the same bytes.
typedef unsigned short u16x2_r __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t
pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2_r)(__builtin_bit_cast(u16x2_r, a)
        + __builtin_bit_cast(u16x2_r, b)));
}

__device__ __forceinline__ uint32_t
pk_sub(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2_r)(__builtin_bit_cast(u16x2_r, a)
        - __builtin_bit_cast(u16x2_r, b)));
}

__device__ __forceinline__ uint32_t
pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(
        __builtin_bit_cast(u16x2_r, a), __builtin_bit_cast(u16x2_r, b)));
}

// max over a DPP pattern with bound_ctrl: a lane without a source reads 0, the
// identity of an unsigned maximum -- one v_max_u32_dpp, nothing to move into
// the destination first (the minimum below is taken as the maximum of the
// complements)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t
max_dpp0(uint32_t v)
{
    return max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true));
}

// FULL: 128 planes, every lane of a half wave has four (no idle lanes to reset)
template <int K, bool FULL>
__global__ void __launch_bounds__(64)
sgm_paths2_kernel(PathArgs A)
{
    int const w = A.w, h = A.h, D = A.D;
    int const ndiag = w + h - 1;
    // block -> (direction, pair of lines); the long horizontal lines first
    int const counts[8] = { h, h, w, ndiag, ndiag, w, ndiag, ndiag };
    int b = blockIdx.x;
    int dir = 0;
    for (; dir < 8; ++dir) {
        int const pairs_of_dir = (counts[dir] + 1) >> 1;
        if (b < pairs_of_dir)
            break;
        b -= pairs_of_dir;
    }
    if (dir > 7)
        return;
    int const dirs[8][2] = { { 1, 0 }, { -1, 0 }, { 0, 1 }, { 1, 1 }, { -1, 1 },
        { 0, -1 }, { 1, -1 }, { -1, -1 } };
    A.dx = dirs[dir][0];
    A.dy = dirs[dir][1];

    int const lane = threadIdx.x;
    int const half = lane >> 5, hl = lane & 31;
    int x0 = 0, y0 = 0, len = 0, extra_seed = 0;
    bool const has_line = path_line(A, 2 * b + half, &x0, &y0, &len, &extra_seed);
    if (!has_line)
        len = 0;
    bool const upper = half != 0;
    bool const ok = has_line && (FULL || 4 * hl < D);
    int const li = ok ? hl : 0;
    // the line as running pointers, four planes (bytes) per lane
    size_t const o0 = has_line ? (((size_t)y0 * w + x0) * D >> 2) + li : 0;
    ptrdiff_t const step = ((ptrdiff_t)A.dy * w + A.dx) * D / 4;
    const uint32_t *__restrict__ cin = reinterpret_cast<const uint32_t *>(A.cost) + o0;
    uint32_t *__restrict__ e32
        = reinterpret_cast<uint32_t *>(A.delta + (size_t)dir * A.vol) + o0;
    // "no such plane": above every path cost, and + P1 (<= 255 in this form)
    // still fits 16 bits
    uint32_t const BIG2 = 0x7FFF7FFFu;
    uint32_t const p1p1 = (uint32_t)A.p1 | ((uint32_t)A.p1 << 16);
    uint32_t const p2p2 = (uint32_t)A.p2 | ((uint32_t)A.p2 << 16);
    // v_perm_b32 selectors of the two neighbour vectors that reach into the
    // adjacent lanes -- {plane 3 of the lane before, own plane 0} and {own plane
    // 3, plane 0 of the lane after} -- per lane: at the ends of a line (where
    // the lane before / after belongs to the OTHER line of the wave) the bytes
    // 0x00, 0xff instead, i.e. 0xff00: no such plane
    uint32_t const sel_below = hl == 0 ? 0x05040d0cu : 0x05040302u;   // perm(pa, pb_prev)
    uint32_t const sel_above = hl == 31 ? 0x0d0c0302u : 0x05040302u;  // perm(pa_next, pb)

    // ---- the first cell of a line: L = C (sgm_stereo.cc:457-464; a corner that
    // is seeded from its row and from its column adds C twice) ----
    uint32_t pa = BIG2, pb = BIG2;     // planes {4 hl, 4 hl + 1}, {4 hl + 2, 4 hl + 3}
    if (has_line) {
        uint32_t const c = *cin;
        uint32_t const ca = __builtin_amdgcn_perm(0u, c, 0x0c010c00u);
        uint32_t const cb = __builtin_amdgcn_perm(0u, c, 0x0c030c02u);
        if (ok) {
            pa = ca;
            pb = cb;
            *e32 = extra_seed ? c : 0u;
        }
    }
    cin += step;
    e32 += step;
    // the remaining steps of the two lines as scalars: every "is this step
    // inside my line" below is then a lane mask built by scalar instructions
    int const rest0 = max(__builtin_amdgcn_readlane(len, 0) - 1, 0);
    int const rest1 = max(__builtin_amdgcn_readlane(len, 32) - 1, 0);
    int const rest_max = max(rest0, rest1), rest_min = min(rest0, rest1);
    auto const inside = [&](int r) -> bool {
        return (!upper & (r < rest0)) | (upper & (r < rest1));
    };

    uint32_t c_cur[K], c_next[K], outv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        c_cur[k] = 0;
        if (inside(k))
            c_cur[k] = cin[(ptrdiff_t)k * step];
    }
    cin += (ptrdiff_t)K * step;
    // one chunk of K steps; FAST: this chunk and the next lie inside both lines
    // (no predicates on the loads and stores)
    auto const chunk = [&](auto fast_tag, int base) {
        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            c_next[k] = 0;
            if (FAST || inside(base + K + k))
                c_next[k] = cin[(ptrdiff_t)k * step];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            outv[k] = 0;
            if (FAST || base + k < rest_max) {
                // the four cost bytes as two u16 pairs
                uint32_t const ca = __builtin_amdgcn_perm(0u, c_cur[k], 0x0c010c00u);
                uint32_t const cb = __builtin_amdgcn_perm(0u, c_cur[k], 0x0c030c02u);
                // the minimum over the line: per lane, then over its half wave
                uint32_t m = pk_min(pa, pb);
                m = ~min(m & 0xFFFFu, m >> 16);
                m = max_dpp0<0x111, 0xf>(m);   // row_shr:1
                m = max_dpp0<0x112, 0xf>(m);   // row_shr:2
                m = max_dpp0<0x114, 0xf>(m);   // row_shr:4
                m = max_dpp0<0x118, 0xf>(m);   // row_shr:8
                m = max_dpp0<0x142, 0xa>(m);   // row_bcast:15 -> rows 1, 3
                uint32_t const m_lower = ~(uint32_t)__builtin_amdgcn_readlane((int)m, 31);
                uint32_t const m_upper = ~(uint32_t)__builtin_amdgcn_readlane((int)m, 63);
                uint32_t const mm_lower = m_lower | (m_lower << 16);
                uint32_t const mm_upper = m_upper | (m_upper << 16);
                uint32_t const mnmn = upper ? mm_upper : mm_lower;
                uint32_t const far = pk_add(mnmn, p2p2);
                // neighbouring planes: {3 of the lane before, 0}, {1, 2}, {3, 0 of the lane after}
                // (wave_shr:1 / wave_shl:1; the lanes without a source are ends of
                // a line, whose selectors do not look at what arrives)
                uint32_t const pb_prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pb, 0x138,
                    0xf, 0xf, true);
                uint32_t const pa_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pa, 0x130,
                    0xf, 0xf, true);
                uint32_t const below_a = __builtin_amdgcn_perm(pa, pb_prev, sel_below);
                uint32_t const mid = __builtin_amdgcn_alignbit(pb, pa, 16);
                uint32_t const above_b = __builtin_amdgcn_perm(pa_next, pb, sel_above);
                uint32_t const mid1 = pk_add(mid, p1p1);
                // :310-346: L = C + min(L'(d), L'(d -+ 1) + P1, min L' + P2) - min L'
                uint32_t const ua = pk_min(pk_min(pa, pk_add(below_a, p1p1)), pk_min(mid1, far));
                uint32_t const ub = pk_min(pk_min(pb, mid1), pk_min(pk_add(above_b, p1p1), far));
                uint32_t const ea = pk_sub(ua, mnmn), eb = pk_sub(ub, mnmn);
                pa = pk_add(ca, ea);
                pb = pk_add(cb, eb);
                outv[k] = __builtin_amdgcn_perm(eb, ea, 0x06040200u);
                if (!FULL && !ok)
                    pa = pb = BIG2;
            }
        }
        if (FULL && FAST) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                e32[(ptrdiff_t)k * step] = outv[k];
        } else if (ok) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (FAST || inside(base + k))
                    e32[(ptrdiff_t)k * step] = outv[k];
        }
    };
    for (int base = 0; base < rest_max; base += K) {
        if (base + 2 * K <= rest_min)
            chunk(std::true_type(), base);
        else
            chunk(std::false_type(), base);
        cin += (ptrdiff_t)K * step;
        e32 += (ptrdiff_t)K * step;
#pragma unroll
        for (int k = 0; k < K; ++k)
            c_cur[k] = c_next[k];
    }
}

// All eight path directions in ONE launch.  The recurrences of different
// directions are independent; only their sums meet in S.  Since the sums are
// wrapping integer adds, S is accumulated with device-scope atomic adds on
// the packed u32 (two u16 planes that cannot carry into each other: eight
// paths of at most 255 + P2 plus the seeds stay below 65536), so the result
// is bit-exact for any interleaving.  ~9000 wavefronts instead of <= 1500 per
// launch, and the wall time is the longest line instead of the sum over
// directions.  S must be zeroed first.
//
// DELTA form (penalty2 <= 255): what a path adds to S at a pixel is
// L = C + (u - min_prev) with 0 <= u - min_prev <= P2 (sgm_stereo.cc:310-346:
// u is the minimum of terms that are all >= min_prev, one of them min_prev +
// P2), and C -- or 2 C at a doubly seeded corner -- at the start of a line.
// So every direction stores L - C as ONE BYTE per cell into its own volume
// with plain coalesced stores (each cell lies on exactly one line per
// direction: no atomics, no zero fill) and sgm_sum_wta_kernel forms
// S = 8 C + the eight bytes on the fly: 16 + 9 bytes per cost cell instead of
// 8 x (1 + 4) + 2 with the read-modify-writes of the u16 volume.
template <int K, bool DELTA>
__global__ void __launch_bounds__(64)
sgm_all_paths_kernel(PathArgs A)
{
    // block -> (direction, line); the long horizontal lines come first
    int const w = A.w, h = A.h, D = A.D;
    int const ndiag = w + h - 1;
    int b = blockIdx.x;
    int dir;
    if (b < 2 * h) {
        dir = b / h;              // 0: ->, 1: <-
        b -= dir * h;
    } else {
        b -= 2 * h;
        // remaining six: (0,1) (1,1) (-1,1) (0,-1) (1,-1) (-1,-1)
        int const counts[6] = { w, ndiag, ndiag, w, ndiag, ndiag };
        dir = 2;
        for (int k = 0; k < 6; ++k) {
            if (b < counts[k])
                break;
            b -= counts[k];
            dir += 1;
        }
        if (dir > 7)
            return;
    }
    int const dirs[8][2] = { { 1, 0 }, { -1, 0 }, { 0, 1 }, { 1, 1 }, { -1, 1 },
        { 0, -1 }, { 1, -1 }, { -1, -1 } };
    A.dx = dirs[dir][0];
    A.dy = dirs[dir][1];

    int const lane = threadIdx.x;
    int x0, y0, len, extra_seed;
    if (!path_line(A, b, &x0, &y0, &len, &extra_seed))
        return;
    int const pairs = D >> 1;
    bool const ok = lane < pairs;
    int const li = ok ? lane : 0;
    // The line as two running pointers (cost in, path bytes / S out): the
    // cell of step s is `step` u16 pairs behind the cell of step s - 1.
    size_t const o0 = (((size_t)y0 * w + x0) * D >> 1) + li;
    ptrdiff_t const step = ((ptrdiff_t)A.dy * w + A.dx) * D / 2;
    const uint16_t *__restrict__ cin = reinterpret_cast<const uint16_t *>(A.cost) + o0;
    uint32_t *__restrict__ s32 = reinterpret_cast<uint32_t *>(A.sgm) + o0;
    uint16_t *__restrict__ e16
        = reinterpret_cast<uint16_t *>(A.delta + (size_t)dir * A.vol) + o0;
    // "no such plane" / "lane without planes": above every path cost
    // (L <= 255 + P2 < 2^15 by check_sgm_options), and BIG + P1 still fits 16
    // bits, so no sum below needs a mask
    uint32_t const BIG = 0x7FFFu;
    uint32_t prev0 = BIG, prev1 = BIG;

    uint32_t c_cur[K], c_next[K], addv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        c_cur[k] = 0;
        if (k < len)
            c_cur[k] = cin[(ptrdiff_t)k * step];
    }
    cin += (ptrdiff_t)K * step;
    for (int base = 0; base < len; base += K) {
        int const n = min(K, len - base);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            c_next[k] = 0;
            if (base + K + k < len)
                c_next[k] = cin[(ptrdiff_t)k * step];
        }
        cin += (ptrdiff_t)K * step;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            addv[k] = 0;
            if (k < n) {
                int const s = base + k;
                uint32_t const c0 = c_cur[k] & 0xFFu, c1 = c_cur[k] >> 8;
                uint32_t e0, e1;     // what the path adds beyond C
                if (s == 0) {
                    // sgm_stereo.cc:457-464: the line starts with L = C; a corner
                    // that is seeded from its row and from its column adds C twice
                    e0 = extra_seed ? c0 : 0u;
                    e1 = extra_seed ? c1 : 0u;
                    prev0 = c0;
                    prev1 = c1;
                } else {
                    // :310-346: L = C + min(L'(d), L'(d -+ 1) + P1, min L' + P2) - min L'
                    uint32_t const mn = wave_min_u32(min(prev0, prev1));
                    uint32_t const left = lane_prev(prev1, BIG);
                    uint32_t const right = lane_next(prev0, BIG);
                    uint32_t const far = mn + A.p2;
                    uint32_t const u0 = min(min(prev0, left + A.p1), min(prev1 + A.p1, far));
                    uint32_t const u1 = min(min(prev1, prev0 + A.p1), min(right + A.p1, far));
                    e0 = u0 - mn;
                    e1 = u1 - mn;
                    prev0 = c0 + e0;
                    prev1 = c1 + e1;
                }
                if (DELTA)
                    addv[k] = e0 | (e1 << 8);
                else
                    addv[k] = (c0 + e0) | ((c1 + e1) << 16);
                if (!ok)
                    prev0 = prev1 = BIG;
            }
        }
        if (ok) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k < n) {
                    if (DELTA)
                        e16[(ptrdiff_t)k * step] = (uint16_t)addv[k];
                    else
                        (void)__hip_atomic_fetch_add(&s32[(ptrdiff_t)k * step], addv[k],
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        e16 += (ptrdiff_t)K * step;
        s32 += (ptrdiff_t)K * step;
#pragma unroll
        for (int k = 0; k < K; ++k)
            c_cur[k] = c_next[k];
    }
}

// WTA with 16 lanes per pixel (sgm_stereo.cc:274-306): lane sub reads planes
// sub, sub + 16, ...; the first minimum wins through the (value, plane) key.
__global__ void __launch_bounds__(256)
wta_rows_kernel(const uint16_t *__restrict__ sgm,
    const uint8_t *__restrict__ main_img, const float *__restrict__ depths,
    size_t npix, int D, float *__restrict__ depth, int32_t *__restrict__ argmin)
{
    size_t const p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int const sub = threadIdx.x & 15;
    uint32_t key = 0xFFFFFFFFu;
    if (p < npix)
        for (int d = sub; d < D; d += 16)
            key = min(key, (uint32_t)sgm[p * D + d] * 256u + (uint32_t)d);
    uint32_t const ident = 0xFFFFFFFFu;
#define SMVS_DPP(x, ctrl)                                                     \
    (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)(x), ctrl, 0xf,   \
        0xf, false)
    key = min(key, SMVS_DPP(key, 0x111));
    key = min(key, SMVS_DPP(key, 0x112));
    key = min(key, SMVS_DPP(key, 0x114));
    key = min(key, SMVS_DPP(key, 0x118));
#undef SMVS_DPP
    if (sub == 15 && p < npix) {
        int const min_index = (int)(key & 0xFFu);
        if (argmin != nullptr)
            argmin[p] = min_index;
        if (depth != nullptr)
            depth[p] = (min_index < 2 || main_img[p] < 25) ? 0.0f
                : depths[min_index];
    }
}

// S = 8 C + the eight path bytes of the DELTA form, and the winner-takes-all of
// wta_rows_kernel on it, 32 lanes per pixel with four planes each (one u32
// per volume and lane).  S itself is only written when the caller wants the
// volume (smvs_sgm_run's `sgm` output).
__global__ void __launch_bounds__(256)
sgm_sum_wta_kernel(const uint8_t *__restrict__ cost, const uint8_t *__restrict__ delta,
    size_t vol, const uint8_t *__restrict__ main_img, const float *__restrict__ depths,
    size_t npix, int D, float *__restrict__ depth, int32_t *__restrict__ argmin,
    uint16_t *__restrict__ sgm_out)
{
    size_t const p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    int const sub = threadIdx.x & 31;
    int const d0 = 4 * sub;
    uint32_t key = 0xFFFFFFFFu;
    if (p < npix && d0 < D) {
        size_t const o = p * (size_t)D + d0;   // D % 4 == 0: aligned u32
        uint32_t const c = *reinterpret_cast<const uint32_t *>(cost + o);
        uint32_t e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            e[k] = *reinterpret_cast<const uint32_t *>(delta + (size_t)k * vol + o);
        uint32_t sv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t sum = 8u * ((c >> (8 * j)) & 0xFFu);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                sum += (e[k] >> (8 * j)) & 0xFFu;
            sv[j] = sum & 0xFFFFu;
            key = min(key, sv[j] * 256u + (uint32_t)(d0 + j));
        }
        if (sgm_out != nullptr) {
            uint2 const packed = make_uint2(sv[0] | (sv[1] << 16), sv[2] | (sv[3] << 16));
            *reinterpret_cast<uint2 *>(sgm_out + o) = packed;
        }
    }
    // minimum over the 32 lanes of the pixel: inside the rows of 16 by DPP
    // shifts, then across the two rows
    uint32_t const ident = 0xFFFFFFFFu;
#define SMVS_DPP(x, ctrl)                                                     \
    (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)(x), ctrl, 0xf,   \
        0xf, false)
    key = min(key, SMVS_DPP(key, 0x111));
    key = min(key, SMVS_DPP(key, 0x112));
    key = min(key, SMVS_DPP(key, 0x114));
    key = min(key, SMVS_DPP(key, 0x118));
#undef SMVS_DPP
    // lanes 15 and 31 of the pixel hold the row minima
    uint32_t const other = (uint32_t)__shfl_xor((int)key, 16);
    key = min(key, other);
    if (sub == 31 && p < npix) {
        int const min_index = (int)(key & 0xFFu);
        if (argmin != nullptr)
            argmin[p] = min_index;
        if (depth != nullptr)
            depth[p] = (min_index < 2 || main_img[p] < 25) ? 0.0f
                : depths[min_index];
    }
}

__global__ void __launch_bounds__(256)
widen_u8_kernel(const uint8_t *__restrict__ src, uint16_t *__restrict__ dst,
    size_t n)
{
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = src[i];
}

// depth_optimizer.cc:957-1004
struct BilateralArgs {
    const float *dm;
    const float *ci;
    float *out;
    int dm_w, dm_h, w, h, channels, kernel_size;
    float sigma;
};

__device__ __forceinline__ float
exp_rounded(float x)
{
    return (float)exp((double)x);
}

__global__ void __launch_bounds__(256)
bilateral_kernel(BilateralArgs A)
{
#pragma clang fp contract(off)
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= A.w)
        return;
    float const scale_x = (float)A.dm_w / (float)A.w;
    float const scale_y = (float)A.dm_h / (float)A.h;
    float acc_v = 0.0f, acc_w = 0.0f;
    for (int ky = -A.kernel_size; ky <= A.kernel_size; ++ky)
        for (int kx = -A.kernel_size; kx <= A.kernel_size; ++kx) {
            int const ci_x = min(max(x + kx, 0), A.w - 1);
            int const ci_y = min(max(y + ky, 0), A.h - 1);
            float fx = scale_x * (float)ci_x, fy = scale_y * (float)ci_y;
            fx = fminf(fmaxf(fx, 0.f), (float)A.dm_w - 1.f);
            fy = fminf(fmaxf(fy, 0.f), (float)A.dm_h - 1.f);
            int const dm_x = (int)fx, dm_y = (int)fy;
            float const dv = A.dm[(size_t)dm_y * A.dm_w + dm_x];
            if (dv == 0.0f)
                continue;
            // math::gaussian / gaussian_2d are std::exp on floats: the host's
            // expf is correctly rounded (glibc), the device's float expf is
            // not, so the exponential is taken in double and rounded once.
            // The initial surface then matches the CPU path bit for bit.
            float weight = 1.0f;
            weight *= exp_rounded(-((float)kx * (float)kx
                / (2.0f * A.sigma * A.sigma)
                + (float)ky * (float)ky / (2.0f * A.sigma * A.sigma)));
            for (int c = 0; c < A.channels; ++c) {
                float const diff =
                    A.ci[((size_t)ci_y * A.w + ci_x) * A.channels + c]
                    - A.ci[((size_t)y * A.w + x) * A.channels + c];
                weight *= exp_rounded(-(diff * diff) / (2.0f * 0.1f * 0.1f));
            }
            acc_v += dv * weight;
            acc_w += weight;
        }
    A.out[(size_t)y * A.w + x] = acc_w > 0 ? acc_v / acc_w : 0.0f;
}

// The same filter when the guidance image is a byte image divided by 255 (the
// main image a context holds): the colour weight of a tap is a function of the
// two bytes only.  The HOST evaluates the reference's own float expression
// with expf for all 256 x 256 pairs (math::gaussian is std::exp on floats, and
// glibc's expf is not correctly rounded in ~0.3 % of its arguments, so only the
// host's own values give the CPU path's weights bit for bit) and compresses
// them for LDS: the weight depends on the pair almost only through the
// difference d = tap - centre -- the float rounding of the two quotients
// leaves at most four distinct values per d -- so the table is 511 x 4 floats
// plus a 2-bit selector per pair: 24 KB.  No exponential on the device; the
// spatial weights of the (2 k + 1)^2 taps are kernel arguments.
constexpr int BIL_MAX_K = 7;
constexpr int BIL_VALS = 2048;            // 511 differences x 4 candidates (floats)
constexpr int BIL_SEL = 65536 / 16;       // 2-bit selectors, 16 per word
constexpr int BIL_TABLE_WORDS = BIL_VALS + BIL_SEL;
struct BilateralSpatial { float w[(2 * BIL_MAX_K + 1) * (2 * BIL_MAX_K + 1)]; };

// f = (float)b / 255.0f is inverted exactly by rounding f * 255
__global__ void __launch_bounds__(256)
float_to_byte_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, size_t n)
{
#pragma clang fp contract(off)
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = (uint8_t)(in[i] * 255.0f + 0.5f);
}

template <int C>
__global__ void __launch_bounds__(256)
bilateral_table_kernel(BilateralArgs A, const uint8_t *__restrict__ ci8,
    const uint32_t *__restrict__ table, BilateralSpatial S)
{
#pragma clang fp contract(off)
    __shared__ uint32_t lds[BIL_TABLE_WORDS];
    for (int i = threadIdx.x; i < BIL_TABLE_WORDS; i += 256)
        lds[i] = table[i];
    __syncthreads();
    const float *vals = reinterpret_cast<const float *>(lds);
    const uint32_t *sel = lds + BIL_VALS;
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= A.w)
        return;
    float const scale_x = (float)A.dm_w / (float)A.w;
    float const scale_y = (float)A.dm_h / (float)A.h;
    int const ks = A.kernel_size, kw = 2 * ks + 1;
    unsigned centre[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
        centre[c] = ci8[((size_t)y * A.w + x) * C + c];
    float acc_v = 0.0f, acc_w = 0.0f;
    for (int ky = -ks; ky <= ks; ++ky) {
        int const ci_y = min(max(y + ky, 0), A.h - 1);
        float fy = scale_y * (float)ci_y;
        fy = fminf(fmaxf(fy, 0.f), (float)A.dm_h - 1.f);
        const float *dm_row = A.dm + (size_t)(int)fy * A.dm_w;
        const uint8_t *ci_row = ci8 + (size_t)ci_y * A.w * C;
        for (int kx = -ks; kx <= ks; ++kx) {
            int const ci_x = min(max(x + kx, 0), A.w - 1);
            float fx = scale_x * (float)ci_x;
            fx = fminf(fmaxf(fx, 0.f), (float)A.dm_w - 1.f);
            float const dv = dm_row[(int)fx];
            if (dv == 0.0f)
                continue;
            float weight = 1.0f;
            weight *= S.w[(ky + ks) * kw + (kx + ks)];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                unsigned const b = ci_row[ci_x * C + c];
                unsigned const pair = (centre[c] << 8) | b;
                unsigned const which = (sel[pair >> 4] >> ((pair & 15u) * 2u)) & 3u;
                weight *= vals[((b + 255u - centre[c]) << 2) | which];
            }
            acc_v += dv * weight;
            acc_w += weight;
        }
    }
    A.out[(size_t)y * A.w + x] = acc_w > 0 ? acc_v / acc_w : 0.0f;
}

// The compressed colour-weight table, or nullptr when some difference has
// more than four distinct weights (another libm: the exponentials are then
// taken on the device as in bilateral_kernel).
static const uint32_t *
bilateral_colour_table(void)
{
#pragma clang fp contract(off)
    static std::vector<uint32_t> table;
    static bool usable = false;
    static std::once_flag once;
    std::call_once(once, []() {
        std::vector<uint32_t> t(BIL_TABLE_WORDS, 0u);
        int count[511] = { 0 };
        bool ok = true;
        for (int a = 0; a < 256 && ok; ++a)
            for (int b = 0; b < 256; ++b) {
                // gaussian(tap - centre, 0.1) as the reference evaluates it on floats
                float const diff = (float)b / 255.0f - (float)a / 255.0f;
                float const wgt = expf(-(diff * diff) / (2.0f * 0.1f * 0.1f));
                uint32_t bits;
                memcpy(&bits, &wgt, sizeof(bits));
                int const d = b + 255 - a;
                int k = 0;
                while (k < count[d] && t[(size_t)d * 4 + k] != bits)
                    k += 1;
                if (k == count[d]) {
                    if (k == 4) {
                        ok = false;
                        break;
                    }
                    t[(size_t)d * 4 + k] = bits;
                    count[d] += 1;
                }
                unsigned const pair = ((unsigned)a << 8) | (unsigned)b;
                t[BIL_VALS + (pair >> 4)] |= (uint32_t)k << ((pair & 15u) * 2u);
            }
        usable = ok;
        table.swap(t);
    });
    return usable ? table.data() : nullptr;
}

// Round 6: the same weights as ONE lookup per tap and channel.  The colour
// weight of a pair of bytes is symmetric to the bit -- (float)b / 255 - (float)a
// / 255 changes its sign exactly when the bytes change places, and the weight
// squares it -- so the table of all pairs is a triangle of 256 x 257 / 2 floats
// = 131,584 bytes: it fits the CU's 160 KB of LDS whole.  A persistent grid of
// one workgroup of 1,024 lanes per CU loads it once and walks over the image;
// per tap and channel: minimum, maximum, the triangle's index, one LDS read (the
// compressed table above: two dependent LDS reads and twelve vector
// instructions, and the kernel was bound by both).  Half width BIL_TRI_K (the
// reference's default, depth_optimizer.h:70-72) with the window's columns
// unrolled -- the clamped column of the guidance image and the column of the
// depth map a tap reads are formed once per pixel, not once per tap.  Same taps,
// same order, same products: the filtered map is array_equal with the oracle's
// (tests/test_gpu_front.py).  SMVS_BILATERAL=compressed: the kernel above.
constexpr int BIL_TRI_K = 5;
constexpr int BIL_TRI_FLOATS = 256 * 257 / 2;
constexpr int BIL_TRI_THREADS = 1024;

template <int C>
__global__ void __launch_bounds__(BIL_TRI_THREADS)
bilateral_triangle_kernel(BilateralArgs A, const uint8_t *__restrict__ ci8,
    const float *__restrict__ triangle, BilateralSpatial S)
{
#pragma clang fp contract(off)
    extern __shared__ float tri[];
    for (int i = threadIdx.x; i < BIL_TRI_FLOATS; i += BIL_TRI_THREADS)
        tri[i] = triangle[i];
    __syncthreads();
    constexpr int KS = BIL_TRI_K, KW = 2 * KS + 1;
    float const scale_x = (float)A.dm_w / (float)A.w;
    float const scale_y = (float)A.dm_h / (float)A.h;
    long long const npix = (long long)A.w * A.h;
    for (long long pix = (long long)blockIdx.x * BIL_TRI_THREADS + threadIdx.x; pix < npix;
        pix += (long long)gridDim.x * BIL_TRI_THREADS) {
        int const y = (int)(pix / A.w), x = (int)(pix - (long long)y * A.w);
        unsigned centre[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
            centre[c] = ci8[(size_t)pix * C + c];
        // the columns of the window: byte offset in a row of the guidance image,
        // column of the depth map
        int col_ci[KW], col_dm[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            int const ci_x = min(max(x + k - KS, 0), A.w - 1);
            float fx = scale_x * (float)ci_x;
            fx = fminf(fmaxf(fx, 0.f), (float)A.dm_w - 1.f);
            col_ci[k] = ci_x * C;
            col_dm[k] = (int)fx;
        }
        float acc_v = 0.0f, acc_w = 0.0f;
#pragma unroll 1
        for (int ky = -KS; ky <= KS; ++ky) {
            int const ci_y = min(max(y + ky, 0), A.h - 1);
            float fy = scale_y * (float)ci_y;
            fy = fminf(fmaxf(fy, 0.f), (float)A.dm_h - 1.f);
            const float *dm_row = A.dm + (size_t)(int)fy * A.dm_w;
            const uint8_t *ci_row = ci8 + (size_t)ci_y * A.w * C;
            const float *sw = S.w + (ky + KS) * KW;
            // every load of a window row is issued before the first is used, the
            // bytes of a tap without depth included (a tap was: depth -> branch ->
            // bytes -> table, three round trips in a row, eleven times per row)
            float dv[KW];
            unsigned bytes[KW][C];
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                dv[k] = dm_row[col_dm[k]];
#pragma unroll
                for (int c = 0; c < C; ++c)
                    bytes[k][c] = ci_row[col_ci[k] + c];
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                float weight = 1.0f;
                weight *= sw[k];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    unsigned const b = bytes[k][c];
                    unsigned const lo = min(b, centre[c]), hi = max(b, centre[c]);
                    // byte offset 4 (hi (hi + 1) / 2 + lo) = (2 hi) hi + 2 hi + 4 lo:
                    // a shift, a 24-bit multiply-add, a shift-add
                    unsigned const h2 = hi << 1;
                    unsigned t;
                    asm("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(t) : "v"(h2), "v"(hi));
                    weight *= *reinterpret_cast<const float *>(
                        reinterpret_cast<const char *>(tri) + (t + (lo << 2)));
                }
                // a tap without depth is skipped by the reference: it adds +0 to
                // both sums here, which leaves them as they are to the bit (the
                // sums start at +0 and the weights are positive: never -0)
                acc_v += dv[k] * weight;
                acc_w += dv[k] == 0.0f ? 0.0f : weight;
            }
        }
        A.out[pix] = acc_w > 0 ? acc_v / acc_w : 0.0f;
    }
}

// The triangle of colour weights (index hi (hi + 1) / 2 + lo), or nullptr when
// some pair is not symmetric to the bit (it always is, see above; checked
// because the bits are the host libm's).
static const float *
bilateral_colour_triangle(void)
{
#pragma clang fp contract(off)
    static std::vector<float> table;
    static bool usable = false;
    static std::once_flag once;
    std::call_once(once, []() {
        std::vector<float> t((size_t)BIL_TRI_FLOATS, 0.0f);
        bool ok = true;
        for (int a = 0; a < 256 && ok; ++a)
            for (int b = 0; b < 256; ++b) {
                // gaussian(tap - centre, 0.1) as the reference evaluates it on floats
                float const diff = (float)b / 255.0f - (float)a / 255.0f;
                float const wgt = expf(-(diff * diff) / (2.0f * 0.1f * 0.1f));
                int const lo = a < b ? a : b, hi = a < b ? b : a;
                size_t const at = (size_t)hi * (size_t)(hi + 1) / 2 + (size_t)lo;
                if (a <= b) {
                    t[at] = wgt;
                } else {
                    // (a > b: the mirrored pair has been stored)
                    if (std::memcmp(&t[at], &wgt, sizeof(float)) != 0) {
                        ok = false;
                        break;
                    }
                }
            }
        usable = ok;
        table.swap(t);
    });
    return usable ? table.data() : nullptr;
}

// ------------------------------------------------------- L/R check + merge
// SGMStereo::reconstruct, sgm_stereo.cc:64-91: the main view's depth is kept
// where its correspondence in the neighbour (integer pixel coordinates, no
// +0.5; Correspondence in double from the float M, t) lies inside the 3 %
// border and the neighbour's own depth agrees within a factor 0.8; truncating
// lookup.  Operation order of correspondence.cc:20-51, contraction off.
struct LrArgs {
    float *d_main;
    const float *d_neig;
    int w, h, nw, nh, cut;
    double M[9], t[3];
};

__global__ void __launch_bounds__(256)
sgm_lr_check_kernel(LrArgs A)
{
#pragma clang fp contract(off)
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= A.w)
        return;
    size_t const o = (size_t)y * A.w + x;
    float const dm = A.d_main[o];
    if (dm == 0.0f)
        return;
    double const u = (double)x, v = (double)y, wd = (double)dm;
    double const p = A.M[0] * u + A.M[1] * v + A.M[2];
    double const q = A.M[3] * u + A.M[4] * v + A.M[5];
    double const r = A.M[6] * u + A.M[7] * v + A.M[8];
    double const a = wd * p + A.t[0];
    double const b = wd * q + A.t[1];
    double const d = wd * r + A.t[2];
    double const cx = a / d, cy = b / d;
    if (cx < (double)A.cut || cx >= (double)(A.nw - A.cut)
        || cy < (double)A.cut || cy >= (double)(A.nh - A.cut)) {
        A.d_main[o] = 0.0f;
        return;
    }
    float const cdepth = (float)d;
    float const ndepth = A.d_neig[(size_t)(int)cy * A.nw + (size_t)(int)cx];
    float const ratio = fminf(cdepth, ndepth) / fmaxf(cdepth, ndepth);
    if (ndepth == 0.0f || (double)ratio < 0.8)
        A.d_main[o] = 0.0f;
}

// app/smvsrecon.cc:366-377: average where both maps are valid
__global__ void __launch_bounds__(256)
sgm_merge_kernel(float *__restrict__ d1, const float *__restrict__ d2, size_t n)
{
#pragma clang fp contract(off)
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float const b = d2[i];
    if (b == 0.0f)
        return;
    float const a = d1[i];
    d1[i] = a == 0.0f ? b : (a + b) * 0.5f;
}

// Optional per-kernel timing of the front end (smvs_sgm_profile): HIP events
// on the workspace's stream around every launch of a call, read back when the
// call has synchronised.  Off by default: no events, no overhead.
static std::mutex g_sgm_prof_mutex;
static bool g_sgm_prof_on = false;
static double g_sgm_prof_ms[SMVS_SGM_K_COUNT] = { 0 };
static long long g_sgm_prof_launches[SMVS_SGM_K_COUNT] = { 0 };

struct SgmProfile {
    struct Pending { int cls; hipEvent_t a, b; };
    std::vector<Pending> pending;
    bool on;
    SgmProfile()
    {
        std::lock_guard<std::mutex> guard(g_sgm_prof_mutex);
        on = g_sgm_prof_on;
    }
    // (called when the stream is idle)
    ~SgmProfile()
    {
        if (pending.empty())
            return;
        std::lock_guard<std::mutex> guard(g_sgm_prof_mutex);
        for (auto &p : pending) {
            float ms = 0.f;
            if (hipEventSynchronize(p.b) == hipSuccess
                && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                g_sgm_prof_ms[p.cls] += ms;
                g_sgm_prof_launches[p.cls] += 1;
            }
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    }
};

struct SgmKernelTimer {
    SgmProfile *prof;
    hipStream_t stream;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    SgmKernelTimer(SgmProfile *p, hipStream_t s, int c) : prof(p), stream(s), cls(c)
    {
        if (prof == nullptr || !prof->on)
            return;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, stream);
    }
    ~SgmKernelTimer()
    {
        if (a == nullptr)
            return;
        (void)hipEventRecord(b, stream);
        prof->pending.push_back({ cls, a, b });
    }
};

// Slots of a pooled workspace (pool.hip) used by this file.
enum {
    WS_DEPTHS = 0, WS_CENSUS, WS_WARPED, WS_COST, WS_SGM, WS_ARGMIN,   // one run_sgm
    WS_MAIN, WS_NBR0, WS_NBR1, WS_FWD0, WS_FWD1, WS_BWD, WS_COST16,   // a view's front end
    WS_RAW, WS_RAW0, WS_RAW1,                                         // raw u8 images + scratch
    WS_BIL_DM, WS_BIL_CI, WS_BIL_OUT,                                 // bilateral upsample
    WS_DELTA                                                          // eight path-byte volumes
};

// Work buffers of one run_sgm inside a pooled workspace; reused by the runs of
// a view (the runs are ordered on the workspace's stream).  Every run has its
// own depth table.
struct SgmWorkspace {
    static constexpr int MAX_RUNS = 4;
    Workspace *ws;
    SgmProfile *prof = nullptr;
    float *depths = nullptr;
    unsigned long long *census = nullptr;
    uint8_t *warped = nullptr, *cost = nullptr;
    uint16_t *sgm = nullptr;     // S: only when the caller wants it or the DELTA form does not apply
    uint8_t *delta = nullptr;    // the eight path-byte volumes of the DELTA form
    int32_t *argmin = nullptr;
    int runs = 0;
    bool want_sgm = false;       // smvs_sgm_run hands the S volume to its caller
    explicit SgmWorkspace(Workspace *w) : ws(w) {}
    // penalty2 <= 255: L - C fits a byte; planes in fours: the u32 accesses of
    // sgm_sum_wta_kernel
    static bool delta_form(int num_steps, unsigned penalty2)
    {
        return (num_steps % 4) == 0 && penalty2 <= 255u;
    }
    int ensure(size_t npix, int num_steps, unsigned penalty2)
    {
        size_t const vol = npix * (size_t)num_steps;
        bool const df = delta_form(num_steps, penalty2);
        int rc;
        if ((rc = ws->ensure(WS_DEPTHS, (size_t)128 * MAX_RUNS, &depths))
            || (rc = ws->ensure(WS_CENSUS, npix, &census))
            || (rc = ws->ensure(WS_WARPED, vol, &warped))
            || (rc = ws->ensure(WS_COST, vol, &cost))
            || (rc = ws->ensure(WS_ARGMIN, npix, &argmin)))
            return rc;
        if (df && (rc = ws->ensure(WS_DELTA, 8 * vol, &delta)))
            return rc;
        if ((!df || want_sgm) && (rc = ws->ensure(WS_SGM, vol, &sgm)))
            return rc;
        return SMVS_OK;
    }
};

static int
check_sgm_options(int num_steps, float min_depth, float max_depth,
    unsigned penalty1, unsigned penalty2)
{
    SMVS_REQUIRE(num_steps >= 2 && num_steps <= 128,
        "num_steps must be in [2, 128]");
    SMVS_REQUIRE(min_depth > 0.f && max_depth > min_depth, "bad depth range");
    SMVS_REQUIRE(penalty2 >= penalty1, "penalty2 must not be below penalty1");
    // The all-paths kernel adds the eight path costs into S with u32 atomics
    // on packed u16 pairs: exact only while no 16-bit lane can carry into its
    // neighbour, i.e. while S stays below 2^16.  Per path L <= 255 + P2 (Q20),
    // border pixels add C up to 4 times more (Q19).  The reference wraps every
    // u16 lane on its own (_mm_add_epi16), which this bound never reaches.
    SMVS_REQUIRE(8u * (255u + penalty2) + 4u * 255u < 65536u,
        "penalty2 too large for the u16 aggregation volume");
    return SMVS_OK;
}

// SGMStereo::run_sgm (sgm_stereo.cc:98-124) on device images; the depth map
// (and optionally argmin) stay on the device.  Asynchronous on `stream`.
static int
sgm_run_device(SgmWorkspace &B, const uint8_t *d_main,
    int w, int h, const uint8_t *d_nbr, int nw, int nh, const float *M,
    const float *t, float min_depth, float max_depth, int num_steps,
    uint16_t penalty1, uint16_t penalty2, float *d_depth)
{
    int rc = check_sgm_options(num_steps, min_depth, max_depth, penalty1,
        penalty2);
    if (rc != SMVS_OK)
        return rc;
    SMVS_REQUIRE(B.runs < SgmWorkspace::MAX_RUNS, "too many runs on one workspace");
    hipStream_t const stream = B.ws->stream;
    size_t const npix = (size_t)w * h;
    size_t const vol = npix * num_steps;
    if ((rc = B.ensure(npix, num_steps, penalty2)) != SMVS_OK)
        return rc;
    // sgm_stereo.cc:195-203: inverse-depth planes by repeated float addition
    float depths[128];
    {
#pragma clang fp contract(off)
        float inv_depth = 1.0f / max_depth;
        float const increment = (1.0f / min_depth - inv_depth) / (num_steps - 1);
        for (int i = 0; i < num_steps; ++i) {
            depths[i] = 1.0f / inv_depth;
            inv_depth += increment;
        }
    }
    float *d_depths = B.depths + 128 * B.runs;
    B.runs += 1;
    if ((rc = B.ws->upload(d_depths, depths, sizeof(float) * num_steps)) != SMVS_OK)
        return rc;

    {
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_CENSUS);
        hipLaunchKernelGGL(census_main_kernel, dim3((w + 255) / 256, h), dim3(256),
            0, stream, d_main, w, h, B.census);
    }
    WarpArgs W;
    W.neighbor = d_nbr;
    W.nw = nw;
    W.nh = nh;
    memcpy(W.M, M, sizeof(float) * 9);
    memcpy(W.t, t, sizeof(float) * 3);
    W.depths = d_depths;
    W.D = num_steps;
    W.w = w;
    W.h = h;
    W.warped = B.warped;
    {
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_WARP);
        hipLaunchKernelGGL(warp_kernel, dim3((unsigned)((w + WARP_TILE - 1) / WARP_TILE),
            (unsigned)h, (unsigned)((num_steps + WARP_TILE - 1) / WARP_TILE)),
            dim3(WARP_TILE), 0, stream, W);
    }
    {
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_COST);
        int const tiles = ((w + CT_W - 1) / CT_W) * ((h + CT_H - 1) / CT_H);
        // (SMVS_SGM_COST=tiled: the one-plane-per-lane kernel for every plane count)
        static bool const force_tiled = [] {
            const char *e = std::getenv("SMVS_SGM_COST");
            return e != nullptr && e[0] == 't';
        }();
        // (SMVS_SGM_XCD=0: tiles in plain order, rounds 1-5; A/B)
        static bool const bands = [] {
            const char *e = std::getenv("SMVS_SGM_XCD");
            return !(e != nullptr && e[0] == '0');
        }();
        if ((num_steps & 3) == 0 && !force_tiled)
            hipLaunchKernelGGL(cost_packed_kernel,
                dim3(bands ? ((tiles + 7) / 8) * 8 : tiles, (num_steps + CP_D - 1) / CP_D),
                dim3(256), 0, stream, B.warped, B.census, w, h, num_steps, B.cost,
                bands ? 1 : 0);
        else
            hipLaunchKernelGGL(cost_tiled_kernel,
                dim3(tiles, (num_steps + CT_D - 1) / CT_D), dim3(256), 0, stream,
                B.warped, B.census, w, h,
                num_steps, B.cost);
    }
    SMVS_HIP_CHECK(hipGetLastError());

    // the eight paths in the reference's order: ->, <-, then the three
    // top-to-bottom paths, then the three bottom-to-top paths
    static const int dirs[8][2] = { { 1, 0 }, { -1, 0 }, { 0, 1 }, { 1, 1 },
        { -1, 1 }, { 0, -1 }, { 1, -1 }, { -1, -1 } };
    PathArgs P;
    P.cost = B.cost;
    P.sgm = B.sgm;
    P.w = w;
    P.h = h;
    P.D = num_steps;
    P.p1 = penalty1;
    P.p2 = penalty2;
    P.last = 0;
    P.delta = B.delta;
    P.vol = vol;
    bool const df = SgmWorkspace::delta_form(num_steps, penalty2);
    // (SMVS_SGM_PATHS=wave: a wave per line, two planes per lane -- rounds 3-5)
    static bool const wave_per_line = [] {
        const char *e = std::getenv("SMVS_SGM_PATHS");
        return e != nullptr && e[0] == 'w';
    }();
    if (df && (num_steps & 3) == 0 && num_steps <= 128 && !wave_per_line) {
        P.dx = P.dy = 0;
        P.first = 0;
        int const nd = w + h - 1;
        int const pairs = 2 * ((h + 1) / 2) + 2 * ((w + 1) / 2) + 4 * ((nd + 1) / 2);
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_PATHS);
        if (num_steps == 128)
            hipLaunchKernelGGL((sgm_paths2_kernel<8, true>), dim3(pairs), dim3(64), 0, stream, P);
        else
            hipLaunchKernelGGL((sgm_paths2_kernel<8, false>), dim3(pairs), dim3(64), 0, stream, P);
    } else if (df) {
        P.dx = P.dy = 0;
        P.first = 0;
        int const lines = 2 * h + 2 * w + 4 * (w + h - 1);
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_PATHS);
        hipLaunchKernelGGL((sgm_all_paths_kernel<16, true>), dim3(lines), dim3(64), 0,
            stream, P);
    } else if ((num_steps % 2) == 0) {
        SMVS_HIP_CHECK(hipMemsetAsync(B.sgm, 0, sizeof(uint16_t) * vol, stream));
        P.dx = P.dy = 0;
        P.first = 0;
        int const lines = 2 * h + 2 * w + 4 * (w + h - 1);
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_PATHS);
        hipLaunchKernelGGL((sgm_all_paths_kernel<16, false>), dim3(lines), dim3(64), 0,
            stream, P);
    } else {
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_PATHS);
        // odd plane counts: one launch per direction, scalar accesses
        for (int k = 0; k < 8; ++k) {
            P.dx = dirs[k][0];
            P.dy = dirs[k][1];
            P.first = k == 0 ? 1 : 0;
            int const lines = P.dy == 0 ? h : (P.dx == 0 ? w : w + h - 1);
            hipLaunchKernelGGL(sgm_path_kernel, dim3(lines), dim3(64), 0, stream,
                P);
        }
    }
    SMVS_HIP_CHECK(hipGetLastError());
    {
        SgmKernelTimer timer(B.prof, stream, SMVS_SGM_K_WTA);
        if (df)
            hipLaunchKernelGGL(sgm_sum_wta_kernel,
                dim3((unsigned)((npix * 32 + 255) / 256)), dim3(256), 0, stream,
                B.cost, B.delta, vol, d_main, d_depths, npix, num_steps, d_depth,
                B.argmin, B.want_sgm ? B.sgm : nullptr);
        else
            hipLaunchKernelGGL(wta_rows_kernel,
                dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, stream,
                B.sgm, d_main, d_depths, npix, num_steps, d_depth,
                B.argmin);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    return SMVS_OK;
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_sgm_run(int device, const uint8_t *main_img, int w, int h,
    const uint8_t *neighbor_img, int nw, int nh, const float *M,
    const float *t, float min_depth, float max_depth, int num_steps,
    uint16_t penalty1, uint16_t penalty2, float *depth, int32_t *argmin,
    uint16_t *cost, uint16_t *sgm)
{
    SMVS_REQUIRE(main_img && neighbor_img && M && t, "null argument");
    SMVS_REQUIRE(w > 10 && h > 8 && nw > 1 && nh > 1, "image too small");
    int rc = check_sgm_options(num_steps, min_depth, max_depth, penalty1,
        penalty2);
    if (rc != SMVS_OK)
        return rc;
    WorkspaceLease lease(device);
    if (lease.w == nullptr)
        return SMVS_ERR_HIP;
    Workspace &ws = *lease.w;
    size_t const npix = (size_t)w * h, nnpix = (size_t)nw * nh;
    size_t const vol = npix * num_steps;
    SgmProfile prof;
    SgmWorkspace B(&ws);
    B.prof = &prof;
    B.want_sgm = sgm != nullptr;
    uint8_t *d_main = nullptr, *d_nbr = nullptr;
    float *d_depth = nullptr;
    if ((rc = ws.ensure(WS_MAIN, npix, &d_main)) || (rc = ws.ensure(WS_NBR0, nnpix, &d_nbr))
        || (rc = ws.ensure(WS_FWD0, npix, &d_depth))
        || (rc = ws.upload(d_main, main_img, npix))
        || (rc = ws.upload(d_nbr, neighbor_img, nnpix)))
        return rc;
    if ((rc = sgm_run_device(B, d_main, w, h, d_nbr, nw, nh, M, t, min_depth,
            max_depth, num_steps, penalty1, penalty2, d_depth)) != SMVS_OK)
        return rc;
    if (depth != nullptr
        && (rc = ws.download(depth, d_depth, sizeof(float) * npix)) != SMVS_OK)
        return rc;
    if (argmin != nullptr
        && (rc = ws.download(argmin, B.argmin, sizeof(int32_t) * npix)) != SMVS_OK)
        return rc;
    if (sgm != nullptr
        && (rc = ws.download(sgm, B.sgm, sizeof(uint16_t) * vol)) != SMVS_OK)
        return rc;
    if (cost != nullptr) {
        uint16_t *d_cost16 = nullptr;
        if ((rc = ws.ensure(WS_COST16, vol, &d_cost16)))
            return rc;
        hipLaunchKernelGGL(widen_u8_kernel, dim3((unsigned)((vol + 255) / 256)),
            dim3(256), 0, ws.stream, B.cost, d_cost16, vol);
        SMVS_HIP_CHECK(hipGetLastError());
        if ((rc = ws.download(cost, d_cost16, sizeof(uint16_t) * vol)) != SMVS_OK)
            return rc;
    }
    SMVS_HIP_CHECK(hipStreamSynchronize(ws.stream));
    return SMVS_OK;
}

// StereoView::get_byte_image (desaturate<uint8_t>, stereo_view.cc:86-95
// [MVE-unverified]: 0.21 r + 0.72 g + 0.07 b + 0.5, truncated) on the device
__global__ void __launch_bounds__(256)
sgm_desaturate_kernel(const uint8_t *__restrict__ in, size_t npix, int channels,
    uint8_t *__restrict__ out)
{
#pragma clang fp contract(off)
    size_t const p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix)
        return;
    if (channels < 3) {
        out[p] = in[p * channels];
        return;
    }
    float const v = (float)in[p * channels] * 0.21f + (float)in[p * channels + 1] * 0.72f
        + (float)in[p * channels + 2] * 0.07f + 0.5f;
    out[p] = (uint8_t)v;
}

// mve::image::rescale_half_size<uint8_t> (sgm_stereo.cc:31-39 [MVE-unverified]):
// mean of the 2 x 2 block (odd sizes repeat the last row / column), + 0.5, truncated
__global__ void __launch_bounds__(256)
sgm_half_size_kernel(const uint8_t *__restrict__ in, int w, int h,
    uint8_t *__restrict__ out)
{
#pragma clang fp contract(off)
    int const ow = (w + 1) >> 1, oh = (h + 1) >> 1;
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= ow || y >= oh)
        return;
    int const x0 = 2 * x, x1 = min(2 * x + 1, w - 1);
    int const y0 = 2 * y, y1 = min(2 * y + 1, h - 1);
    float const v = (float)in[(size_t)y0 * w + x0] * 0.25f
        + (float)in[(size_t)y0 * w + x1] * 0.25f
        + (float)in[(size_t)y1 * w + x0] * 0.25f
        + (float)in[(size_t)y1 * w + x1] * 0.25f;
    out[(size_t)y * ow + x] = (uint8_t)(v + 0.5f);
}

// One view's SGM input image on the device: upload (raw: interleaved u8 of
// `channels`; otherwise already at SGM scale, one channel), desaturate and
// `halvings` half-size steps.  *out (slot `slot_out`) receives the image,
// *ow / *oh its size.
static int
sgm_prepare_image(Workspace &ws, const uint8_t *host, int w, int h, int channels,
    int halvings, int slot_out, int slot_tmp, uint8_t **out, int *ow, int *oh)
{
    int rc;
    size_t const npix = (size_t)w * h;
    uint8_t *a = nullptr, *b = nullptr;
    if (channels == 1 && halvings == 0) {
        if ((rc = ws.ensure(slot_out, npix, &a)) || (rc = ws.upload(a, host, npix)))
            return rc;
        *out = a;
        *ow = w;
        *oh = h;
        return SMVS_OK;
    }
    // raw bytes into the scratch slot, results ping-pong between the two
    if ((rc = ws.ensure(slot_tmp, npix * (size_t)channels + npix, &b))
        || (rc = ws.ensure(slot_out, npix, &a))
        || (rc = ws.upload(b, host, npix * (size_t)channels)))
        return rc;
    uint8_t *grey = b + npix * (size_t)channels;   // behind the raw bytes
    hipLaunchKernelGGL(sgm_desaturate_kernel, dim3((unsigned)((npix + 255) / 256)),
        dim3(256), 0, ws.stream, b, npix, channels, halvings % 2 == 0 ? a : grey);
    uint8_t *cur = halvings % 2 == 0 ? a : grey;
    uint8_t *other = halvings % 2 == 0 ? grey : a;
    int cw = w, ch = h;
    for (int i = 0; i < halvings; ++i) {
        int const nw = (cw + 1) >> 1, nh = (ch + 1) >> 1;
        hipLaunchKernelGGL(sgm_half_size_kernel, dim3((nw + 255) / 256, nh), dim3(256), 0,
            ws.stream, cur, cw, ch, other);
        uint8_t *t = cur; cur = other; other = t;
        cw = nw;
        ch = nh;
    }
    SMVS_HIP_CHECK(hipGetLastError());
    // (an even number of swaps ends in `a` when it started there, an odd one
    // when it started in `grey`: cur == a by construction)
    *out = cur;
    *ow = cw;
    *oh = ch;
    return SMVS_OK;
}

// reconstruct_sgm_depth_for_view on prepared device images
static int
sgm_depth_for_view_impl(int device, const uint8_t *main_img, int w, int h,
    int main_channels, const smvs_sgm_neighbor *neighbors,
    const int *neighbor_channels, int n_neighbors, int halvings, int num_steps,
    uint16_t penalty1, uint16_t penalty2, float *depth)
{
    SMVS_REQUIRE(main_img && neighbors && depth, "null argument");
    SMVS_REQUIRE(n_neighbors >= 1 && n_neighbors <= 2,
        "one or two neighbours (app/smvsrecon.cc:360-365)");
    SMVS_REQUIRE(halvings >= 0 && halvings <= 8, "halvings out of range");
    SMVS_REQUIRE((w >> halvings) > 10 && (h >> halvings) > 8, "image too small");
    for (int k = 0; k < n_neighbors; ++k)
        SMVS_REQUIRE(neighbors[k].image && (neighbors[k].width >> halvings) > 10
            && (neighbors[k].height >> halvings) > 8, "bad neighbour image");
    int rc;
    WorkspaceLease lease(device);
    if (lease.w == nullptr)
        return SMVS_ERR_HIP;
    Workspace &ws = *lease.w;
    hipStream_t const stream = ws.stream;
    SgmProfile prof;
    SgmWorkspace B(&ws);
    B.prof = &prof;
    // the SGM-scale images (every buffer before the first SGM launch: growing
    // one waits for the stream)
    uint8_t *d_main = nullptr, *d_nbr[2] = { nullptr, nullptr };
    int mw = 0, mh = 0, nw[2] = { 0, 0 }, nh[2] = { 0, 0 };
    if ((rc = sgm_prepare_image(ws, main_img, w, h, main_channels, halvings, WS_MAIN,
             WS_RAW, &d_main, &mw, &mh)) != SMVS_OK)
        return rc;
    for (int k = 0; k < n_neighbors; ++k)
        if ((rc = sgm_prepare_image(ws, neighbors[k].image, neighbors[k].width,
                 neighbors[k].height, neighbor_channels != nullptr ? neighbor_channels[k] : 1,
                 halvings, k == 0 ? WS_NBR0 : WS_NBR1, k == 0 ? WS_RAW0 : WS_RAW1,
                 &d_nbr[k], &nw[k], &nh[k])) != SMVS_OK)
            return rc;
    size_t const npix = (size_t)mw * mh;
    float *d_fwd[2] = { nullptr, nullptr }, *d_bwd = nullptr;
    size_t max_nnpix = 0;
    for (int k = 0; k < n_neighbors; ++k) {
        size_t const nnpix = (size_t)nw[k] * nh[k];
        max_nnpix = nnpix > max_nnpix ? nnpix : max_nnpix;
        if ((rc = ws.ensure(k == 0 ? WS_FWD0 : WS_FWD1, npix, &d_fwd[k])))
            return rc;
    }
    if ((rc = ws.ensure(WS_BWD, max_nnpix, &d_bwd))
        || (rc = B.ensure(npix > max_nnpix ? npix : max_nnpix, num_steps, penalty2)))
        return rc;
    for (int k = 0; k < n_neighbors; ++k) {
        smvs_sgm_neighbor const &N = neighbors[k];
        // SGMStereo::reconstruct, sgm_stereo.cc:46-62: main -> neighbour,
        // then neighbour -> main with the neighbour's own depth range
        if ((rc = sgm_run_device(B, d_main, mw, mh, d_nbr[k], nw[k], nh[k],
                N.M_fwd, N.t_fwd, N.range_main[0], N.range_main[1], num_steps,
                penalty1, penalty2, d_fwd[k])) != SMVS_OK)
            return rc;
        if ((rc = sgm_run_device(B, d_nbr[k], nw[k], nh[k], d_main, mw, mh,
                N.M_bwd, N.t_bwd, N.range_neighbor[0], N.range_neighbor[1],
                num_steps, penalty1, penalty2, d_bwd)) != SMVS_OK)
            return rc;
        LrArgs L;
        L.d_main = d_fwd[k];
        L.d_neig = d_bwd;
        L.w = mw;
        L.h = mh;
        L.nw = nw[k];
        L.nh = nh[k];
        L.cut = (int)(0.03 * (double)(nw[k] > nh[k] ? nw[k] : nh[k]));
        for (int i = 0; i < 9; ++i)
            L.M[i] = (double)N.M_fwd[i];
        for (int i = 0; i < 3; ++i)
            L.t[i] = (double)N.t_fwd[i];
        {
            SgmKernelTimer timer(&prof, stream, SMVS_SGM_K_LR_CHECK);
            hipLaunchKernelGGL(sgm_lr_check_kernel, dim3((mw + 255) / 256, mh),
                dim3(256), 0, stream, L);
        }
        SMVS_HIP_CHECK(hipGetLastError());
    }
    if (n_neighbors > 1) {
        SgmKernelTimer timer(&prof, stream, SMVS_SGM_K_MERGE);
        hipLaunchKernelGGL(sgm_merge_kernel, dim3((unsigned)((npix + 255) / 256)),
            dim3(256), 0, stream, d_fwd[0], d_fwd[1], npix);
        SMVS_HIP_CHECK(hipGetLastError());
    }
    return ws.download(depth, d_fwd[0], sizeof(float) * npix);
}

extern "C" int
smvs_sgm_depth_for_view(int device, const uint8_t *main_img, int w, int h,
    const smvs_sgm_neighbor *neighbors, int n_neighbors, int num_steps,
    uint16_t penalty1, uint16_t penalty2, float *depth)
{
    return sgm_depth_for_view_impl(device, main_img, w, h, 1, neighbors, nullptr,
        n_neighbors, 0, num_steps, penalty1, penalty2, depth);
}

extern "C" int
smvs_sgm_depth_for_view_raw(int device, const uint8_t *main_img, int w, int h,
    int channels, const smvs_sgm_neighbor *neighbors, const int *neighbor_channels,
    int n_neighbors, int halvings, int num_steps, uint16_t penalty1,
    uint16_t penalty2, float *depth)
{
    SMVS_REQUIRE(channels == 1 || channels == 3, "1 or 3 channels");
    SMVS_REQUIRE(neighbor_channels != nullptr, "null argument");
    for (int k = 0; k < n_neighbors && k < 2; ++k)
        SMVS_REQUIRE(neighbor_channels[k] == 1 || neighbor_channels[k] == 3,
            "1 or 3 channels");
    return sgm_depth_for_view_impl(device, main_img, w, h, channels, neighbors,
        neighbor_channels, n_neighbors, halvings, num_steps, penalty1, penalty2, depth);
}

extern "C" int
smvs_bilateral_upsample(int device, const float *dm, int dm_w, int dm_h,
    const float *ci, int w, int h, int channels, float sigma, int kernel_size,
    float *out)
{
    SMVS_REQUIRE(dm && ci && out, "null argument");
    SMVS_REQUIRE(dm_w > 0 && dm_h > 0 && w > 0 && h > 0 && channels > 0
        && kernel_size >= 0 && sigma > 0.f, "bad argument");
    WorkspaceLease lease(device);
    if (lease.w == nullptr)
        return SMVS_ERR_HIP;
    Workspace &ws = *lease.w;
    float *d_dm = nullptr, *d_ci = nullptr, *d_out = nullptr;
    int rc;
    size_t const n = (size_t)w * h;
    if ((rc = ws.ensure(WS_BIL_DM, (size_t)dm_w * dm_h, &d_dm))
        || (rc = ws.ensure(WS_BIL_CI, n * channels, &d_ci))
        || (rc = ws.ensure(WS_BIL_OUT, n, &d_out))
        || (rc = ws.upload(d_dm, dm, sizeof(float) * dm_w * dm_h))
        || (rc = ws.upload(d_ci, ci, sizeof(float) * n * channels)))
        return rc;
    BilateralArgs A;
    A.dm = d_dm;
    A.ci = d_ci;
    A.out = d_out;
    A.dm_w = dm_w;
    A.dm_h = dm_h;
    A.w = w;
    A.h = h;
    A.channels = channels;
    A.kernel_size = kernel_size;
    A.sigma = sigma;
    SgmProfile prof;
    {
        SgmKernelTimer timer(&prof, ws.stream, SMVS_SGM_K_BILATERAL);
        hipLaunchKernelGGL(bilateral_kernel, dim3((w + 255) / 256, h), dim3(256), 0,
            ws.stream, A);
    }
    SMVS_HIP_CHECK(hipGetLastError());
    return ws.download(out, d_out, sizeof(float) * n);
}

// The same filter for a view whose context already holds the main image
// (smvs_ctx_upload_image): guided by that image, and the full-size result
// stays on the device as the depth map the visibility tests of
// smvs_topology_subviews compare with (lib/depth_optimizer.cc:35-51 hands the
// filtered map to both).  Saves the upload of the float image (25 MB at
// 1920 x 1080 x 3) and of the result, once per topology pass.
// The low-resolution SGM map from page-locked host memory (read over the bus)
// to the device.  from_mve: the map is the view's "smvs-sgm" embedding as
// StereoView::write_depth_to_view stored it (MVE's ray-length convention) and is
// turned into z-depth on the way -- mve::image::depthmap_convert_conventions
// with the float operations of host/stereo_view.cc (StereoView::get_sgm_depth,
// stereo_view.h:121-135), so the same bits as the host conversion.
struct SgmMapUpload {
    const float *src;
    float *dst;
    int w, h;
    int from_mve;
    float invproj[9];
};

__global__ void __launch_bounds__(256)
sgm_map_upload_kernel(SgmMapUpload A)
{
#pragma clang fp contract(off)
    size_t const i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)A.w * A.h)
        return;
    float d = A.src[i];
    if (A.from_mve != 0) {
        int const y = (int)(i / (size_t)A.w), x = (int)(i - (size_t)y * A.w);
        float const px = (float)x + 0.5f, py = (float)y + 0.5f;
        float v[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            v[r] = A.invproj[3 * r] * px + A.invproj[3 * r + 1] * py + A.invproj[3 * r + 2];
        float const len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        // `double len = px.norm(); dm *= 1.0 / len` [MVE-unverified, M10]
        double const len_d = (double)len;
        // (from_mve == 2: the map is still the z-depth the SGM front end produced;
        // the view would store it as ray length first -- write_depth_to_view,
        // `dm *= len` -- and the reference reads it back through that embedding)
        if (A.from_mve == 2)
            d = (float)((double)d * len_d);
        d = (float)((double)d * (1.0 / len_d));
    }
    A.dst[i] = d;
}

static int
sgm_init_depth(smvs_ctx *ctx, const float *dm, int dm_w, int dm_h, const float *inv_calibration9,
    int dm_is_z_depth, float sigma, int kernel_size, float *out);

extern "C" int
smvs_ctx_sgm_init_depth(smvs_ctx *ctx, const float *dm, int dm_w, int dm_h,
    float sigma, int kernel_size, float *out)
{
    return sgm_init_depth(ctx, dm, dm_w, dm_h, nullptr, 0, sigma, kernel_size, out);
}

extern "C" int
smvs_ctx_sgm_init_depth_mve(smvs_ctx *ctx, const float *dm, int dm_w, int dm_h,
    const float *inv_calibration9, int dm_is_z_depth, float sigma, int kernel_size, float *out)
{
    SMVS_REQUIRE(dm == nullptr || inv_calibration9 != nullptr, "null argument");
    return sgm_init_depth(ctx, dm, dm_w, dm_h, inv_calibration9, dm_is_z_depth != 0 ? 1 : 0, sigma,
        kernel_size, out);
}

static int
sgm_init_depth(smvs_ctx *ctx, const float *dm, int dm_w, int dm_h, const float *inv_calibration9,
    int dm_is_z_depth, float sigma, int kernel_size, float *out)
{
    SMVS_REQUIRE(ctx != nullptr, "null argument");
    if (dm == nullptr) {   // forget the resident map
        ctx->sgm_resident = false;
        return SMVS_OK;
    }
    SMVS_REQUIRE(dm_w > 0 && dm_h > 0 && kernel_size >= 0 && sigma > 0.f,
        "bad argument");
    if ((ctx->image_ok & 1u) == 0u) {
        set_error("smvs_ctx_sgm_init_depth: no main image (smvs_ctx_upload_image)");
        return SMVS_ERR_STATE;
    }
    if (ctx->images[0].w != ctx->width || ctx->images[0].h != ctx->height) {
        set_error("smvs_ctx_sgm_init_depth: main image size differs from the context");
        return SMVS_ERR_INVALID;
    }
    SMVS_HIP_CHECK(set_device(ctx->device));
    int rc;
    // (the guide image may still be on its way: smvs_ctx_upload_image_async)
    if ((rc = ctx_materialise_images(ctx, 1u)) != SMVS_OK)
        return rc;
    size_t const n = (size_t)ctx->width * ctx->height;
    size_t const n_low = (size_t)dm_w * dm_h;
    if (ctx->sgm_lowres_cap < n_low) {
        if ((rc = device_alloc(&ctx->sgm_lowres, n_low)) != SMVS_OK)
            return rc;
        ctx->sgm_lowres_cap = n_low;
    }
    if (ctx->topo_sgm_cap < n) {
        if ((rc = device_alloc(&ctx->topo_sgm, n)) != SMVS_OK)
            return rc;
        ctx->topo_sgm_cap = n;
    }
    ctx->sgm_resident = false;
    {
        // The map goes to the device through a kernel that reads page-locked
        // memory over the bus, not as a DMA: at this moment the nine images of the
        // view are on their way (smvs_ctx_upload_image_async) and a tenth transfer
        // queues behind them, with the host waiting for it before it can launch
        // the filter (round 6, profiles/r6_upload_overlap.txt).
        if (ctx->sgm_pin_busy) {
            SMVS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            ctx->sgm_pin_busy = false;
        }
        if (ctx->sgm_pin_cap < n_low) {
            if (ctx->sgm_pin != nullptr)
                (void)hipHostFree(ctx->sgm_pin);
            ctx->sgm_pin = nullptr;
            ctx->sgm_pin_cap = 0;
            SMVS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->sgm_pin),
                sizeof(float) * n_low, hipHostMallocDefault));
            ctx->sgm_pin_cap = n_low;
        }
        std::memcpy(ctx->sgm_pin, dm, sizeof(float) * n_low);
        SgmMapUpload U;
        U.src = ctx->sgm_pin;
        U.dst = ctx->sgm_lowres;
        U.w = dm_w;
        U.h = dm_h;
        U.from_mve = inv_calibration9 != nullptr ? (dm_is_z_depth ? 2 : 1) : 0;
        for (int i = 0; i < 9; ++i)
            U.invproj[i] = inv_calibration9 != nullptr ? inv_calibration9[i] : 0.0f;
        hipLaunchKernelGGL(sgm_map_upload_kernel, dim3((unsigned)((n_low + 255) / 256)),
            dim3(256), 0, ctx->stream, U);
        SMVS_HIP_CHECK(hipGetLastError());
        ctx->sgm_pin_busy = true;
    }
    BilateralArgs A;
    A.dm = ctx->sgm_lowres;
    A.ci = ctx->images[0].data;
    A.out = ctx->topo_sgm;
    A.dm_w = dm_w;
    A.dm_h = dm_h;
    A.w = ctx->width;
    A.h = ctx->height;
    A.channels = ctx->images[0].c;
    A.kernel_size = kernel_size;
    A.sigma = sigma;
    SgmProfile prof;
    const uint32_t *host_table = nullptr;
    if (kernel_size <= BIL_MAX_K && (A.channels == 1 || A.channels == 3))
        host_table = bilateral_colour_table();
    bool const tabled = host_table != nullptr;
    // the triangle of all pairs in LDS (round 6), unless SMVS_BILATERAL=compressed
    const float *host_triangle = nullptr;
    if (tabled && kernel_size == BIL_TRI_K) {
        const char *form = std::getenv("SMVS_BILATERAL");
        if (!(form != nullptr && std::strcmp(form, "compressed") == 0))
            host_triangle = bilateral_colour_triangle();
    }
    BilateralSpatial S = {};
    size_t const n_img = n * (size_t)A.channels;
    if (tabled) {
#pragma clang fp contract(off)
        if (host_triangle != nullptr) {
            if (ctx->bil_tri == nullptr) {
                if ((rc = device_alloc(&ctx->bil_tri, BIL_TRI_FLOATS)) != SMVS_OK
                    || (rc = ctx_upload(ctx, ctx->bil_tri, host_triangle,
                            BIL_TRI_FLOATS * sizeof(float))) != SMVS_OK) {
                    (void)device_alloc(&ctx->bil_tri, 0);
                    return rc;
                }
            }
        } else if (ctx->bil_lut == nullptr) {
            if ((rc = device_alloc(&ctx->bil_lut, BIL_TABLE_WORDS)) != SMVS_OK
                || (rc = ctx_upload(ctx, ctx->bil_lut, host_table,
                        BIL_TABLE_WORDS * sizeof(uint32_t))) != SMVS_OK)
                return rc;
        }
        // (the byte staging buffer of the image uploads is free between them)
        if (ctx->byte_stage_cap < n_img) {
            if ((rc = device_alloc(&ctx->byte_stage, n_img)) != SMVS_OK) {
                ctx->byte_stage_cap = 0;
                return rc;
            }
            ctx->byte_stage_cap = n_img;
        }
        // math::gaussian_2d as bilateral_kernel evaluates it, with the host's expf
        int const kw = 2 * kernel_size + 1;
        for (int ky = -kernel_size; ky <= kernel_size; ++ky)
            for (int kx = -kernel_size; kx <= kernel_size; ++kx)
                S.w[(ky + kernel_size) * kw + (kx + kernel_size)]
                    = expf(-((float)kx * (float)kx / (2.0f * sigma * sigma)
                        + (float)ky * (float)ky / (2.0f * sigma * sigma)));
    }
    {
        SgmKernelTimer timer(&prof, ctx->stream, SMVS_SGM_K_BILATERAL);
        dim3 const grid((ctx->width + 255) / 256, ctx->height);
        if (tabled) {
            hipLaunchKernelGGL(float_to_byte_kernel, dim3((unsigned)((n_img + 255) / 256)),
                dim3(256), 0, ctx->stream, ctx->images[0].data, ctx->byte_stage, n_img);
            const uint32_t *table = reinterpret_cast<const uint32_t *>(ctx->bil_lut);
            if (host_triangle != nullptr) {
                // one workgroup per CU, the table in its LDS for the whole image
                size_t const lds = BIL_TRI_FLOATS * sizeof(float);
                const void *k = A.channels == 3
                    ? reinterpret_cast<const void *>(&bilateral_triangle_kernel<3>)
                    : reinterpret_cast<const void *>(&bilateral_triangle_kernel<1>);
                if ((rc = allow_dynamic_lds(ctx->device, k, lds)) != SMVS_OK)
                    return rc;
                int cus = 0;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                        physical_device(ctx->device)) != hipSuccess || cus <= 0)
                    cus = 256;
                unsigned const blocks = (unsigned)std::min<size_t>((size_t)cus,
                    (n + BIL_TRI_THREADS - 1) / BIL_TRI_THREADS);
                if (A.channels == 3)
                    hipLaunchKernelGGL(bilateral_triangle_kernel<3>, dim3(blocks),
                        dim3(BIL_TRI_THREADS), lds, ctx->stream, A, ctx->byte_stage, ctx->bil_tri, S);
                else
                    hipLaunchKernelGGL(bilateral_triangle_kernel<1>, dim3(blocks),
                        dim3(BIL_TRI_THREADS), lds, ctx->stream, A, ctx->byte_stage, ctx->bil_tri, S);
            } else if (A.channels == 3)
                hipLaunchKernelGGL(bilateral_table_kernel<3>, grid, dim3(256), 0, ctx->stream,
                    A, ctx->byte_stage, table, S);
            else
                hipLaunchKernelGGL(bilateral_table_kernel<1>, grid, dim3(256), 0, ctx->stream,
                    A, ctx->byte_stage, table, S);
        } else {
            hipLaunchKernelGGL(bilateral_kernel, grid, dim3(256), 0, ctx->stream, A);
        }
    }
    SMVS_HIP_CHECK(hipGetLastError());
    // (no wait when the caller does not want the filtered map back: whatever
    // reads it next runs behind the filter on the context's stream)
    if (out != nullptr) {
        if ((rc = ctx_download(ctx, out, ctx->topo_sgm, sizeof(float) * n)) != SMVS_OK)
            return rc;
        ctx->sgm_pin_busy = false;
    }
    ctx->sgm_resident = true;
    return SMVS_OK;
}

extern "C" int
smvs_sgm_profile(int enable, double *ms, long long *launches)
{
    std::lock_guard<std::mutex> guard(g_sgm_prof_mutex);
    for (int i = 0; i < SMVS_SGM_K_COUNT; ++i) {
        if (ms != nullptr)
            ms[i] = g_sgm_prof_ms[i];
        if (launches != nullptr)
            launches[i] = g_sgm_prof_launches[i];
    }
    if (enable >= 0) {
        // (switching it on or off starts a new measurement)
        g_sgm_prof_on = enable != 0;
        for (int i = 0; i < SMVS_SGM_K_COUNT; ++i) {
            g_sgm_prof_ms[i] = 0.0;
            g_sgm_prof_launches[i] = 0;
        }
    }
    return SMVS_OK;
}
