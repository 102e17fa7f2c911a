// Host mirror of lib/stereo_view.cc.  The scale-space blur and the 3x3
// quadratic-fit gradients are the step *before* the device path
// (SURVEY.md 8(f)-1); they run on the host here.  MVE image semantics are
// recalled, not read [MVE-unverified].
#include "stereo_view.h"

#include <cmath>

namespace smvs_amd {

namespace imgtools {

FloatImage::Ptr
blur_gaussian(FloatImage::ConstPtr in, float sigma)
{
    if (std::fabs(sigma) < 0.1f)
        return in->duplicate();
    int const w = in->width(), h = in->height(), c = in->channels();
    int const ks = (int)std::ceil(sigma * 2.884f);
    std::vector<float> kernel(ks + 1);
    for (int i = 0; i <= ks; ++i)
        kernel[i] = std::exp(-((float)i * (float)i) / (2.0f * sigma * sigma));
    FloatImage::Ptr sep = FloatImage::create(w, h, c);
    FloatImage::Ptr out = FloatImage::create(w, h, c);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int cc = 0; cc < c; ++cc) {
                float v = 0.0f, wsum = 0.0f;
                for (int i = -ks; i <= ks; ++i) {
                    int const idx = std::min(std::max(x + i, 0), w - 1);
                    float const kw = kernel[std::abs(i)];
                    v += in->at(idx, y, cc) * kw;
                    wsum += kw;
                }
                sep->at(x, y, cc) = v / wsum;
            }
    for (int x = 0; x < w; ++x)
        for (int y = 0; y < h; ++y)
            for (int cc = 0; cc < c; ++cc) {
                float v = 0.0f, wsum = 0.0f;
                for (int i = -ks; i <= ks; ++i) {
                    int const idx = std::min(std::max(y + i, 0), h - 1);
                    float const kw = kernel[std::abs(i)];
                    v += sep->at(x, idx, cc) * kw;
                    wsum += kw;
                }
                out->at(x, y, cc) = v / wsum;
            }
    return out;
}

FloatImage::Ptr
desaturate(FloatImage::ConstPtr in)
{
    int const c = in->channels();
    FloatImage::Ptr out = FloatImage::create(in->width(), in->height(), 1);
    for (int p = 0; p < in->get_pixel_amount(); ++p)
        out->at(p) = c >= 3 ? in->at(p, 0) * 0.21f + in->at(p, 1) * 0.72f
            + in->at(p, 2) * 0.07f : in->at(p, 0);
    return out;
}

ByteImage::Ptr
desaturate(ByteImage::ConstPtr in)
{
    int const c = in->channels();
    ByteImage::Ptr out = ByteImage::create(in->width(), in->height(), 1);
    for (int p = 0; p < in->get_pixel_amount(); ++p)
        out->at(p) = c >= 3 ? (uint8_t)((float)in->at(p, 0) * 0.21f
            + (float)in->at(p, 1) * 0.72f + (float)in->at(p, 2) * 0.07f + 0.5f)
            : in->at(p, 0);
    return out;
}

ByteImage::Ptr
rescale_half_size(ByteImage::ConstPtr in)
{
    int const w = in->width(), h = in->height();
    int const ow = (w + 1) >> 1, oh = (h + 1) >> 1;
    ByteImage::Ptr out = ByteImage::create(ow, oh, 1);
    for (int y = 0; y < oh; ++y) {
        int const y0 = 2 * y, y1 = std::min(2 * y + 1, h - 1);
        for (int x = 0; x < ow; ++x) {
            int const x0 = 2 * x, x1 = std::min(2 * x + 1, w - 1);
            float const v = (float)in->at(x0, y0, 0) * 0.25f
                + (float)in->at(x1, y0, 0) * 0.25f
                + (float)in->at(x0, y1, 0) * 0.25f
                + (float)in->at(x1, y1, 0) * 0.25f;
            out->at(x, y, 0) = (uint8_t)(v + 0.5f);
        }
    }
    return out;
}

void
gradients_and_hessian(FloatImage::ConstPtr input, FloatImage::Ptr gradient,
    FloatImage::Ptr hessian)
{
    // lib/stereo_view.cc:97-188: least-squares quadratic on the 3x3 window
    //   I(a, b) ~ c_xx a^2 + c_yy b^2 + c_xy a b + c_x a + c_y b + c_0
    gradient->fill(0.0f);
    if (hessian != nullptr)
        hessian->fill(0.0f);
    double fit[6][9];
    int col = 0;
    for (int a = -1; a <= 1; ++a)
        for (int b = -1; b <= 1; ++b, ++col) {
            fit[0][col] = a == 0 ? -1.0 / 3.0 : 1.0 / 6.0;
            fit[1][col] = b == 0 ? -1.0 / 3.0 : 1.0 / 6.0;
            fit[2][col] = (double)(a * b) / 4.0;
            fit[3][col] = (double)a / 6.0;
            fit[4][col] = (double)b / 6.0;
            fit[5][col] = (a == 0 && b == 0) ? 5.0 / 9.0
                : ((a == 0 || b == 0) ? 2.0 / 9.0 : -1.0 / 9.0);
        }
    int const w = input->width(), h = input->height();
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            double win[9];
            int k = 0;
            for (int a = -1; a <= 1; ++a)
                for (int b = -1; b <= 1; ++b)
                    win[k++] = input->at(x + a, y + b, 0);
            double r[6];
            for (int q = 0; q < 6; ++q) {
                double s = 0.0;
                for (int i = 0; i < 9; ++i)
                    s += fit[q][i] * win[i];
                r[q] = s;
            }
            gradient->at(x, y, 0) = (float)r[3];
            gradient->at(x, y, 1) = (float)r[4];
            if (hessian == nullptr)
                continue;
            hessian->at(x, y, 0) = (float)(2.0 * r[0]);
            hessian->at(x, y, 1) = (float)r[2];
            hessian->at(x, y, 2) = (float)(2.0 * r[1]);
        }
}

void
depthmap_convert_conventions(FloatImage::Ptr dm, float const* invproj,
    bool to_mve)
{
    int const w = dm->width(), h = dm->height();
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float const px = (float)x + 0.5f, py = (float)y + 0.5f;
            float v[3];
            for (int r = 0; r < 3; ++r)
                v[r] = invproj[3 * r] * px + invproj[3 * r + 1] * py
                    + invproj[3 * r + 2];
            float const len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            // `double len = px.norm(); dm *= (to_mve ? len : 1.0 / len)`
            // [MVE-unverified, tests/golden/README.md M10]
            double const len_d = (double)len;
            dm->at(x, y, 0) = (float)((double)dm->at(x, y, 0)
                * (to_mve ? len_d : 1.0 / len_d));
        }
}

} // namespace imgtools

StereoView::Ptr
StereoView::create(int view_id, ByteImage::ConstPtr bytes,
    CameraInfo const& camera, bool initialize_linear, bool gamma_correction)
{
    Ptr sv(new StereoView());
    sv->view_id = view_id;
    sv->camera = camera;
    sv->bytes = bytes;
    if (initialize_linear)
        sv->initialize_linear(gamma_correction);
    return sv;
}

FloatImage::ConstPtr
StereoView::get_image(void) const
{
    std::call_once(image_once, [this]() {
        FloatImage::Ptr img = FloatImage::create_for_overwrite(bytes->width(), bytes->height(),
            bytes->channels());
        int64_t const n = (int64_t)bytes->get_pixel_amount() * bytes->channels();
        uint8_t const* src = bytes->begin();
        float* dst = img->begin();
        for (int64_t i = 0; i < n; ++i)
            dst[i] = (float)src[i] / 255.0f;
        image = img;
    });
    return image;
}

void
StereoView::set_scale(int scale, bool)
{
    // lib/stereo_view.cc:24-46
    double const sigma = 0.12 * std::pow(2.0, scale) + 0.2;
    this->scaleimage = imgtools::blur_gaussian(this->get_image(), (float)sigma);
    FloatImage::ConstPtr grey = this->scaleimage->channels() > 1
        ? FloatImage::ConstPtr(imgtools::desaturate(this->scaleimage))
        : FloatImage::ConstPtr(this->scaleimage);
    this->image_grad = FloatImage::create(grey->width(), grey->height(), 2);
    this->image_hessian = FloatImage::create(grey->width(), grey->height(), 3);
    imgtools::gradients_and_hessian(grey, this->image_grad, this->image_hessian);
}

void
StereoView::set_scale_planes(FloatImage::Ptr gradients, FloatImage::Ptr hessian)
{
    this->image_grad = gradients;
    this->image_hessian = hessian;
}

void
StereoView::initialize_linear(bool gamma_correction)
{
    // lib/stereo_view.cc:64-84
    this->linear_image = this->get_image()->duplicate();
    if (gamma_correction) {
        int64_t const n = (int64_t)linear_image->get_pixel_amount()
            * linear_image->channels();
        for (int64_t i = 0; i < n; ++i) {
            float const v = linear_image->at(i);
            linear_image->at(i) = v <= 0.04045f ? v / 12.92f
                : std::pow((v + 0.055f) / 1.055f, 2.4f);
        }
    }
    this->shading = this->linear_image->channels() > 1
        ? imgtools::desaturate(this->linear_image) : this->linear_image;
    this->shading_grad = FloatImage::create(shading->width(),
        shading->height(), 2);
    imgtools::gradients_and_hessian(this->shading, this->shading_grad, nullptr);
}

float
StereoView::get_flen(void) const
{
    float proj[9];
    camera.fill_calibration(proj, (float)get_width(), (float)get_height());
    return proj[0];
}

float
StereoView::get_inverse_flen(void) const
{
    float invproj[9];
    camera.fill_inverse_calibration(invproj, (float)get_width(),
        (float)get_height());
    return invproj[0];
}

ByteImage::ConstPtr
StereoView::get_byte_image(void) const
{
    if (bytes->channels() > 1)
        return imgtools::desaturate(bytes);
    return bytes;
}

bool
StereoView::has_embedding(std::string const& name) const
{
    return embeddings.count(name) != 0 || deferred_depth.count(name) != 0;
}

// the conversion write_depth_to_view_deferred put off
void
StereoView::store_deferred(std::string const& name) const
{
    auto it = deferred_depth.find(name);
    if (it == deferred_depth.end())
        return;
    FloatImage::Ptr mve_depth = it->second->duplicate();
    float invproj[9];
    camera.fill_inverse_calibration(invproj, (float)mve_depth->width(),
        (float)mve_depth->height());
    imgtools::depthmap_convert_conventions(mve_depth, invproj, true);
    embeddings[name] = mve_depth;
    deferred_depth.erase(it);
}

FloatImage::Ptr
StereoView::get_embedding(std::string const& name) const
{
    store_deferred(name);
    auto it = embeddings.find(name);
    return it == embeddings.end() ? nullptr : it->second;
}

std::map<std::string, FloatImage::Ptr> const&
StereoView::get_embeddings(void) const
{
    while (!deferred_depth.empty())
        store_deferred(deferred_depth.begin()->first);
    return embeddings;
}

void
StereoView::write_depth_to_view_deferred(FloatImage::Ptr depth, std::string const& name)
{
    embeddings.erase(name);
    deferred_depth[name] = depth;
}

FloatImage::Ptr
StereoView::get_deferred_depth(std::string const& name) const
{
    auto it = deferred_depth.find(name);
    return it == deferred_depth.end() ? nullptr : it->second;
}

FloatImage::Ptr
StereoView::get_sgm_depth(void) const
{
    FloatImage::Ptr stored = get_embedding("smvs-sgm");
    if (stored == nullptr)
        return nullptr;
    FloatImage::Ptr depth = stored->duplicate();
    float invproj[9];
    camera.fill_inverse_calibration(invproj, (float)depth->width(),
        (float)depth->height());
    imgtools::depthmap_convert_conventions(depth, invproj, false);
    return depth;
}

void
StereoView::write_image_to_view(FloatImage::Ptr img, std::string const& name)
{
    deferred_depth.erase(name);
    embeddings[name] = img;
}

void
StereoView::write_depth_to_view(FloatImage::Ptr depth, std::string const& name)
{
    FloatImage::Ptr mve_depth = depth->duplicate();
    float invproj[9];
    camera.fill_inverse_calibration(invproj, (float)mve_depth->width(),
        (float)mve_depth->height());
    imgtools::depthmap_convert_conventions(mve_depth, invproj, true);
    deferred_depth.erase(name);
    embeddings[name] = mve_depth;
}

} // namespace smvs_amd
