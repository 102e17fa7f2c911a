// Minimal image / camera / bundle containers of the host mirror.
// With MVE present these would be mve::Image<T>, mve::CameraInfo and
// mve::Bundle; the layout is MVE's: interleaved
// channels, row-major, index (y*w + x)*c + ch.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

namespace smvs_amd {

// std::vector<T>::resize(n) without the zero fill (assign(n, T(0)) still
// fills).
template <typename T>
struct DefaultInitAllocator : std::allocator<T>
{
    template <typename U>
    struct rebind { typedef DefaultInitAllocator<U> other; };
    DefaultInitAllocator(void) = default;
    template <typename U>
    DefaultInitAllocator(DefaultInitAllocator<U> const&) {}
    template <typename U>
    void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <typename U, typename... Args>
    void construct(U* p, Args&&... args)
    {
        ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
};

// Where large images live.  By default in a std::vector; the host library
// installs a hook (host/pinned_images.cc) that serves images of a megabyte and
// more from page-locked memory (smvs_pinned_alloc, include/smvs_hip.h): the
// views' u8 images and the depth / normal maps then cross PCIe as one DMA each
// instead of through a staging copy.  The hook may decline (returns null: no
// device, out of pinned memory); the vector is the fallback.
struct ImageMemory
{
    typedef void* (*AllocFn)(std::size_t bytes);
    typedef void (*FreeFn)(void* ptr);
    static AllocFn& alloc_hook(void) { static AllocFn f = nullptr; return f; }
    static FreeFn& free_hook(void) { static FreeFn f = nullptr; return f; }
    static constexpr std::size_t MIN_BYTES = (std::size_t)1 << 20;
};

template <typename T>
class Image
{
public:
    typedef std::shared_ptr<Image<T>> Ptr;
    typedef std::shared_ptr<Image<T> const> ConstPtr;

    static Ptr create(int width, int height, int channels)
    {
        Ptr img(new Image<T>());
        img->allocate(width, height, channels);
        std::fill(img->ptr, img->ptr + img->count(), T(0));
        return img;
    }
    // For images whose every value is written right away (a download from the
    // device, a conversion): the 33 MB of a 1920 x 1080 depth + normal map
    // pair are then touched once, not twice.
    static Ptr create_for_overwrite(int width, int height, int channels)
    {
        Ptr img(new Image<T>());
        img->allocate(width, height, channels);
        return img;
    }
    Ptr duplicate(void) const
    {
        Ptr img(new Image<T>());
        img->allocate(w, h, c);
        std::copy(ptr, ptr + count(), img->ptr);
        return img;
    }
    Image(void) = default;
    Image(Image const&) = delete;
    Image& operator=(Image const&) = delete;
    ~Image(void)
    {
        if (external != nullptr && ImageMemory::free_hook() != nullptr)
            ImageMemory::free_hook()(external);
    }

    int width(void) const { return w; }
    int height(void) const { return h; }
    int channels(void) const { return c; }
    int get_pixel_amount(void) const { return w * h; }
    void fill(T const& v) { std::fill(ptr, ptr + count(), v); }

    T& at(int64_t i) { return ptr[i]; }
    T const& at(int64_t i) const { return ptr[i]; }
    T& at(int64_t p, int64_t ch) { return ptr[p * c + ch]; }
    T const& at(int64_t p, int64_t ch) const { return ptr[p * c + ch]; }
    T& at(int64_t x, int64_t y, int64_t ch) { return ptr[(y * w + x) * c + ch]; }
    T const& at(int64_t x, int64_t y, int64_t ch) const
    {
        return ptr[(y * w + x) * c + ch];
    }
    T* begin(void) { return ptr; }
    T const* begin(void) const { return ptr; }

    // bilinear sample with clamping; float weights [MVE-unverified]
    T linear_at(float x, float y, int64_t ch) const;

private:
    std::size_t count(void) const { return (std::size_t)w * h * c; }
    void allocate(int width, int height, int channels)
    {
        w = width;
        h = height;
        c = channels;
        std::size_t const bytes = count() * sizeof(T);
        if (bytes >= ImageMemory::MIN_BYTES && ImageMemory::alloc_hook() != nullptr)
            external = ImageMemory::alloc_hook()(bytes);
        if (external != nullptr) {
            ptr = static_cast<T*>(external);
        } else {
            data.resize(count());
            ptr = data.data();
        }
    }

private:
    int w = 0, h = 0, c = 0;
    T* ptr = nullptr;
    void* external = nullptr;   // page-locked storage from the hook, or null
    std::vector<T, DefaultInitAllocator<T>> data;
};

typedef Image<float> FloatImage;
typedef Image<uint8_t> ByteImage;

template <>
inline float
Image<float>::linear_at(float x, float y, int64_t ch) const
{
    x = x < 0.0f ? 0.0f : (x > (float)(w - 1) ? (float)(w - 1) : x);
    y = y < 0.0f ? 0.0f : (y > (float)(h - 1) ? (float)(h - 1) : y);
    int const fx = (int)x, fy = (int)y;
    int const fx1 = fx + 1 < w - 1 ? fx + 1 : w - 1;
    int const fy1 = fy + 1 < h - 1 ? fy + 1 : h - 1;
    float const w1 = x - (float)fx, w0 = 1.0f - w1;
    float const w3 = y - (float)fy, w2 = 1.0f - w3;
    return at(fx, fy, ch) * (w0 * w2) + at(fx1, fy, ch) * (w1 * w2)
        + at(fx, fy1, ch) * (w0 * w3) + at(fx1, fy1, ch) * (w1 * w3);
}

// mve::CameraInfo subset
struct CameraInfo
{
    float flen = 0.0f;           // normalised by max(width, height)
    float ppoint[2] = { 0.5f, 0.5f };
    float paspect = 1.0f;
    float trans[3] = { 0, 0, 0 };
    float rot[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };

    void fill_calibration(float* mat, float width, float height) const;
    void fill_inverse_calibration(float* mat, float width, float height) const;
    // M = K_d R_d R_s^T K_s^-1, t = K_d (t_d - R_d R_s^T t_s)  (this = source)
    void fill_reprojection(CameraInfo const& destination, float src_width,
        float src_height, float dst_width, float dst_height, float* mat,
        float* vec) const;
};

// mve::Bundle subset: sparse SfM points with the views that see them
struct Bundle
{
    typedef std::shared_ptr<Bundle> Ptr;
    typedef std::shared_ptr<Bundle const> ConstPtr;
    struct Feature3D
    {
        float pos[3];
        std::vector<int> view_ids;
    };
    std::vector<Feature3D> features;
};

} // namespace smvs_amd
