// Cross-view consistency cut of the depth maps on gfx950.
//
// Replaces MeshGenerator::cut_depth_maps (reference: lib/mesh_generator.cc:24-158)
// together with the normal preparation of generate_mesh (:189-208) and
// MeshGenerator::ViewProjection (:300-342).  Every pixel of every view is an
// independent unit: its 3-D point is reprojected into all other views and
// kept when enough of them see a compatible surface there.  One thread per
// pixel walks the other views in the reference's order (the consistency sum
// and its early exit are order dependent); all maps stay resident on the
// device, so N views cost one upload and one download of N maps.
//
// Arithmetic: float, in the reference's operation order, FMA contraction off
// (hipcc's float division and square root are correctly rounded), so the cut
// maps are bit-identical to the CPU path.  MVE pieces (pixel_3dpos,
// fill_cam_to_world, fill_camera_pos, depthmap_convert_conventions) follow the
// assumptions listed in tests/golden/README.md [MVE-unverified].
#include "common.h"

#include <cmath>
#include <vector>

namespace smvs_hip {

struct MeshViewDev {
    int w, h;
    float invproj[9];   // CameraInfo::fill_inverse_calibration
    float KR[9];        // ViewProjection::KR = K * R
    float t[3];         // ViewProjection::t = KR * camera position
    float rot[9];       // world -> camera rotation
    float c2w_t[3];     // translation column of the cam-to-world matrix
    float *depth_z;     // depthmaps[i] after depthmap_convert_conventions(false)
    float *depth_ray;   // cutmaps_j[i]: the input ray-length depth
    float *cut;         // cutmaps[i]
    float *normals;     // world space after the preparation pass
};

__device__ __forceinline__ float
dot3(const float *a, const float *b)
{
#pragma clang fp contract(off)
    float s = 0.0f;
    s += a[0] * b[0];
    s += a[1] * b[1];
    s += a[2] * b[2];
    return s;
}

// mve::geom::pixel_3dpos followed by Matrix4f::mult(pos, 1) with the
// cam-to-world matrix (mesh_generator.cc:80-82, 120-123)
__device__ __forceinline__ void
world_point(MeshViewDev const &V, int x, int y, float depth, float *pos)
{
#pragma clang fp contract(off)
    float const px = (float)x + 0.5f, py = (float)y + 0.5f;
    float v[3];
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
        s += V.invproj[3 * r] * px;
        s += V.invproj[3 * r + 1] * py;
        s += V.invproj[3 * r + 2] * 1.0f;
        v[r] = s;
    }
    float const len = sqrtf(dot3(v, v));
    float pc[3];
    for (int r = 0; r < 3; ++r)
        pc[r] = v[r] / len * depth;
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
        s += V.rot[r] * pc[0];
        s += V.rot[3 + r] * pc[1];
        s += V.rot[6 + r] * pc[2];
        pos[r] = s + V.c2w_t[r] * 1.0f;
    }
}

// ViewProjection::get_surface_power, mesh_generator.cc:321-342
__device__ __forceinline__ float
surface_power(MeshViewDev const &V, const float *pos, const float *normal)
{
#pragma clang fp contract(off)
    const float *KR = V.KR;
    float const u = dot3(KR + 0, pos) - V.t[0];
    float const v = dot3(KR + 3, pos) - V.t[1];
    float const w = dot3(KR + 6, pos) - V.t[2];
    float const denom = w * w;
    float u_dx[3], v_dx[3];
    for (int k = 0; k < 3; ++k) {
        u_dx[k] = (KR[k] * w - KR[6 + k] * u) / denom;
        v_dx[k] = (KR[3 + k] * w - KR[6 + k] * v) / denom;
    }
    float cr[3];
    cr[0] = u_dx[1] * v_dx[2] - u_dx[2] * v_dx[1];
    cr[1] = u_dx[2] * v_dx[0] - u_dx[0] * v_dx[2];
    cr[2] = u_dx[0] * v_dx[1] - u_dx[1] * v_dx[0];
    return -dot3(normal, cr);
}

// generate_mesh :197-208 (normals to world space) and cut_depth_maps :31-47
// (ray-length copies, z-depth conversion) for one view
__global__ void __launch_bounds__(256)
mesh_prepare_kernel(const MeshViewDev *views, int i)
{
#pragma clang fp contract(off)
    MeshViewDev const V = views[i];
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= V.w)
        return;
    size_t const p = (size_t)y * V.w + x;
    float const n[3] = { V.normals[3 * p], -V.normals[3 * p + 1],
        -V.normals[3 * p + 2] };
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
        s += V.rot[r] * n[0];
        s += V.rot[3 + r] * n[1];
        s += V.rot[6 + r] * n[2];
        V.normals[3 * p + r] = s;
    }
    float const d = V.depth_ray[p];
    V.cut[p] = d;
    float const px = (float)x + 0.5f, py = (float)y + 0.5f;
    float v[3];
    for (int r = 0; r < 3; ++r)
        v[r] = V.invproj[3 * r] * px + V.invproj[3 * r + 1] * py
            + V.invproj[3 * r + 2];
    float const len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    // (depthmap_convert_conventions: `double len = px.norm(); dm *= 1.0 / len`,
    // tests/golden/README.md M10)
    V.depth_z[p] = (float)((double)d * (1.0 / (double)len));
}

// cut_depth_maps :61-150 for view i
__global__ void __launch_bounds__(256)
mesh_cut_kernel(const MeshViewDev *__restrict__ views, int n_views, int i)
{
#pragma clang fp contract(off)
    MeshViewDev const V = views[i];
    int const x = blockIdx.x * blockDim.x + threadIdx.x;
    int const y = blockIdx.y;
    if (x >= V.w)
        return;
    size_t const p = (size_t)y * V.w + x;
    float const d = V.depth_ray[p];
    if (d == 0.0f)
        return;
    float pos[3];
    world_point(V, x, y, d, pos);
    float const normal[3] = { V.normals[3 * p], V.normals[3 * p + 1],
        V.normals[3 * p + 2] };
    float const power = surface_power(V, pos, normal);
    bool cut = power < 0;
    float consistency = 0;
    for (int j = 0; j < n_views; ++j) {
        if (j == i)
            continue;
        MeshViewDev const &N = views[j];
        float proj[3];
        proj[0] = dot3(N.KR + 0, pos) - N.t[0];
        proj[1] = dot3(N.KR + 3, pos) - N.t[1];
        proj[2] = dot3(N.KR + 6, pos) - N.t[2];
        if (proj[2] < 0)
            continue;
        int const xj = (int)(proj[0] / proj[2]);
        int const yj = (int)(proj[1] / proj[2]);
        if (xj < 0 || xj >= N.w || yj < 0 || yj >= N.h)
            continue;
        size_t const pj = (size_t)yj * N.w + xj;
        float const dm_j = N.depth_z[pj];
        if (dm_j == 0.0f)
            continue;
        float const power_j = surface_power(N, pos, normal);
        float pos_j[3];
        world_point(N, xj, yj, N.depth_ray[pj], pos_j);
        float const normal_j[3] = { N.normals[3 * pj], N.normals[3 * pj + 1],
            N.normals[3 * pj + 2] };
        float const power_j_j = surface_power(N, pos_j, normal_j);
        if ((double)dm_j * 1.01 < (double)proj[2])
            continue;
        if ((double)dm_j * 0.997 > (double)proj[2]) {
            if ((double)power_j_j > 0.5 * (double)power)
                consistency -= power_j_j;
            continue;
        }
        if ((double)power_j_j > 2.0 * (double)power
            || (double)power_j > 2.0 * (double)power) {
            cut = true;
            break;
        }
        consistency += power_j_j;
    }
    if (consistency <= 0)
        cut = true;
    if (cut)
        V.cut[p] = 0.0f;
}

static void
mat3_mul_f(const float *A, const float *B, float *C)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k)
                s += A[3 * r + k] * B[3 * k + c];
            C[3 * r + c] = s;
        }
}

} // namespace smvs_hip

using namespace smvs_hip;

extern "C" int
smvs_cut_depth_maps(int device, smvs_mesh_view *views, int n_views)
{
    SMVS_REQUIRE(views != nullptr && n_views >= 1, "no views");
    SMVS_REQUIRE(n_views <= 4096, "too many views");
    for (int i = 0; i < n_views; ++i)
        SMVS_REQUIRE(views[i].width > 0 && views[i].height > 0
            && views[i].depth != nullptr && views[i].normals != nullptr
            && views[i].flen > 0.0f, "bad view");
    // one slab of a pooled workspace (pool.hip) holds every view's maps and
    // the view table: no allocation per call
    WorkspaceLease lease(device);
    if (lease.w == nullptr)
        return SMVS_ERR_HIP;
    Workspace &ws = *lease.w;
    size_t total = 0;
    auto carve = [&](size_t bytes) {
        size_t const at = total;
        total += (bytes + 255) & ~(size_t)255;
        return at;
    };
    std::vector<size_t> offsets((size_t)n_views * 4);
    for (int i = 0; i < n_views; ++i) {
        size_t const npix = (size_t)views[i].width * views[i].height;
        offsets[4 * i + 0] = carve(sizeof(float) * npix);       // depth_z
        offsets[4 * i + 1] = carve(sizeof(float) * npix);       // depth_ray
        offsets[4 * i + 2] = carve(sizeof(float) * npix);       // cut
        offsets[4 * i + 3] = carve(sizeof(float) * 3 * npix);   // normals
    }
    size_t const table_at = carve(sizeof(MeshViewDev) * n_views);
    char *slab = nullptr;
    int rc;
    if ((rc = ws.ensure(0, total, &slab)) != SMVS_OK)
        return rc;
    hipStream_t const stream = ws.stream;

    std::vector<MeshViewDev> table(n_views);
    for (int i = 0; i < n_views; ++i) {
        smvs_mesh_view const &in = views[i];
        MeshViewDev &V = table[i];
        V.w = in.width;
        V.h = in.height;
        // CameraInfo::fill_calibration / fill_inverse_calibration (ppoint =
        // 0.5, paspect = 1) at the depth map's size
        float const fw = (float)in.width, fh = (float)in.height;
        float const dim = fw > fh ? fw : fh;
        float const ax = in.flen * dim, ay = in.flen * dim;
        float const K[9] = { ax, 0, fw * 0.5f, 0, ay, fh * 0.5f, 0, 0, 1 };
        float const Ki[9] = { 1.0f / ax, 0, -fw * 0.5f / ax, 0, 1.0f / ay,
            -fh * 0.5f / ay, 0, 0, 1 };
        memcpy(V.invproj, Ki, sizeof(Ki));
        memcpy(V.rot, in.rot, sizeof(V.rot));
        mat3_mul_f(K, in.rot, V.KR);
        // fill_camera_pos = fill_cam_to_world's translation = -R^T t
        float pos[3];
        for (int r = 0; r < 3; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k)
                s += -in.rot[3 * k + r] * in.trans[k];
            pos[r] = s;
            V.c2w_t[r] = s;
        }
        for (int r = 0; r < 3; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k)
                s += V.KR[3 * r + k] * pos[k];
            V.t[r] = s;
        }
        size_t const npix = (size_t)in.width * in.height;
        V.depth_z = reinterpret_cast<float *>(slab + offsets[4 * i + 0]);
        V.depth_ray = reinterpret_cast<float *>(slab + offsets[4 * i + 1]);
        V.cut = reinterpret_cast<float *>(slab + offsets[4 * i + 2]);
        V.normals = reinterpret_cast<float *>(slab + offsets[4 * i + 3]);
        if ((rc = ws.upload(V.depth_ray, in.depth, sizeof(float) * npix))
            || (rc = ws.upload(V.normals, in.normals, sizeof(float) * 3 * npix)))
            return rc;
    }
    MeshViewDev *d_table = reinterpret_cast<MeshViewDev *>(slab + table_at);
    if ((rc = ws.upload(d_table, table.data(), sizeof(MeshViewDev) * n_views)))
        return rc;
    for (int i = 0; i < n_views; ++i)
        hipLaunchKernelGGL(mesh_prepare_kernel,
            dim3((table[i].w + 255) / 256, table[i].h), dim3(256), 0, stream,
            d_table, i);
    SMVS_HIP_CHECK(hipGetLastError());
    // a single depth map is returned unchanged (mesh_generator.cc:211)
    if (n_views > 1)
        for (int i = 0; i < n_views; ++i)
            hipLaunchKernelGGL(mesh_cut_kernel,
                dim3((table[i].w + 255) / 256, table[i].h), dim3(256), 0, stream,
                d_table, n_views, i);
    SMVS_HIP_CHECK(hipGetLastError());
    for (int i = 0; i < n_views; ++i) {
        size_t const npix = (size_t)table[i].w * table[i].h;
        if ((rc = ws.download(views[i].depth, table[i].cut, sizeof(float) * npix))
            || (rc = ws.download(views[i].normals, table[i].normals,
                    sizeof(float) * 3 * npix)))
            return rc;
    }
    return SMVS_OK;
}
