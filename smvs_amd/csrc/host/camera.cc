#include "image.h"

namespace smvs_amd {

void
CameraInfo::fill_calibration(float* mat, float width, float height) const
{
    float dim_aspect = width / height;
    float image_aspect = dim_aspect * paspect;
    float ax, ay;
    if (image_aspect < 1.0f) {  // portrait
        ax = flen * height / paspect;
        ay = flen * height;
    } else {                    // landscape
        ax = flen * width;
        ay = flen * width * paspect;
    }
    mat[0] = ax;   mat[1] = 0.0f; mat[2] = width * ppoint[0];
    mat[3] = 0.0f; mat[4] = ay;   mat[5] = height * ppoint[1];
    mat[6] = 0.0f; mat[7] = 0.0f; mat[8] = 1.0f;
}

void
CameraInfo::fill_inverse_calibration(float* mat, float width,
    float height) const
{
    float dim_aspect = width / height;
    float image_aspect = dim_aspect * paspect;
    float ax, ay;
    if (image_aspect < 1.0f) {
        ax = flen * height / paspect;
        ay = flen * height;
    } else {
        ax = flen * width;
        ay = flen * width * paspect;
    }
    mat[0] = 1.0f / ax; mat[1] = 0.0f;      mat[2] = -width * ppoint[0] / ax;
    mat[3] = 0.0f;      mat[4] = 1.0f / ay; mat[5] = -height * ppoint[1] / ay;
    mat[6] = 0.0f;      mat[7] = 0.0f;      mat[8] = 1.0f;
}

static void
mul3(float const* A, float const* B, float* C)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k)
                s += A[3 * r + k] * B[3 * k + c];
            C[3 * r + c] = s;
        }
}

void
CameraInfo::fill_reprojection(CameraInfo const& dst, float sw, float sh,
    float dw, float dh, float* mat, float* vec) const
{
    float Ks_inv[9], Kd[9], RsT[9], Rrel[9], tmp[9];
    this->fill_inverse_calibration(Ks_inv, sw, sh);
    dst.fill_calibration(Kd, dw, dh);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            RsT[3 * r + c] = this->rot[3 * c + r];
    mul3(dst.rot, RsT, Rrel);
    mul3(Kd, Rrel, tmp);
    mul3(tmp, Ks_inv, mat);
    float v[3];
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += Rrel[3 * r + k] * this->trans[k];
        v[r] = dst.trans[r] - s;
    }
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k)
            s += Kd[3 * r + k] * v[k];
        vec[r] = s;
    }
}

} // namespace smvs_amd
