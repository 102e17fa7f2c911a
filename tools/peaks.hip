// Peak-rate microbenchmarks for MI355X (gfx950): the denominators of the
// rooflines in DESIGN.md / bench.py, measured instead of assumed
// (SURVEY.md 8(d)).  Built by tools/peaks.py into tools/libpeaks.so.
//
//   HBM:   float4 copy, read-only and write-only streams over 1 GiB;
//   FP64:  v_fma_f64 (8 independent chains per lane), v_mfma_f64_16x16x4_f64,
//          v_mfma_f64_4x4x4_4b_f64 (4 accumulators each), v_rcp_f64, and a
//          512-thread workgroup whose waves 0-3 issue VALU FMAs while waves
//          4-7 issue MFMAs (one of each per SIMD): do the two share a pipe?
//   probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(expr)                                                       \
    do {                                                                  \
        hipError_t e__ = (expr);                                          \
        if (e__ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__,  \
                #expr, hipGetErrorString(e__));                           \
            return -1;                                                    \
        }                                                                 \
    } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    size_t const stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += stride)
        dst[i] = src[i];
}

__global__ void __launch_bounds__(256)
read_kernel(const float4 *__restrict__ src, float *__restrict__ sink, size_t n)
{
    size_t const stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += stride) {
        float4 const v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float const s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f)
        sink[0] = s;
}

__global__ void __launch_bounds__(256)
write_kernel(float4 *__restrict__ dst, size_t n, float v)
{
    size_t const stride = (size_t)gridDim.x * blockDim.x;
    float4 const val = make_float4(v, v, v, v);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += stride)
        dst[i] = val;
}

// 8 independent FMA chains per lane, `iters` rounds
__device__ __forceinline__ void
valu_fma_body(double &sinkv, int iters, double a, double b)
{
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5,
        x6 = a + 6, x7 = a + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a);
            x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
            x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a);
            x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a);
        }
    }
    sinkv = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void
valu_fma_kernel(double *sink, int iters, double a, double b)
{
    double s;
    valu_fma_body(s, iters, a + threadIdx.x * 1e-9, b);
    if (s == 0.12345)
        sink[0] = s;
}

// one dependent FMA chain (latency)
__global__ void
valu_chain_kernel(double *sink, int iters, double a, double b)
{
    double x = a + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u)
            x = __builtin_fma(x, b, a);
    }
    if (x == 0.12345)
        sink[0] = x;
}

__global__ void
valu_rcp_kernel(double *sink, int iters, double a)
{
    double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1);
            x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3);
        }
    }
    double const s = x0 + x1 + x2 + x3;
    if (s == 0.12345)
        sink[0] = s;
}

__device__ __forceinline__ void
mfma16_body(double &sinkv, int iters, double a, double b)
{
    double4_t c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    sinkv = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void
mfma16_kernel(double *sink, int iters, double a, double b)
{
    double s;
    mfma16_body(s, iters, a + threadIdx.x * 1e-9, b);
    if (s == 0.12345)
        sink[0] = s;
}

// one accumulator: dependent issue
__global__ void
mfma16_chain_kernel(double *sink, int iters, double a, double b)
{
    double4_t c0 = { 0, 0, 0, 0 };
    double const av = a + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, c0, 0, 0, 0);
    }
    double const s = c0[0] + c0[1] + c0[2] + c0[3];
    if (s == 0.12345)
        sink[0] = s;
}

__global__ void
mfma4_kernel(double *sink, int iters, double a, double b)
{
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    double const av = a + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c7, 0, 0, 0);
        }
    }
    double const s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s == 0.12345)
        sink[0] = s;
}

__global__ void
mfma4_chain_kernel(double *sink, int iters, double a, double b)
{
    double c0 = 0;
    double const av = a + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, c0, 0, 0, 0);
    }
    if (c0 == 0.12345)
        sink[0] = c0;
}

// 512 threads: waves 0-3 VALU, waves 4-7 MFMA (mode 0); all VALU (1);
// all MFMA (2).  One of each kind lands on every SIMD in mode 0.
__global__ void __launch_bounds__(512)
mixed_kernel(double *sink, int iters_valu, int iters_mfma, int mode, double a,
    double b)
{
    int const wave = threadIdx.x >> 6;
    bool const valu = mode == 1 || (mode == 0 && wave < 4);
    double s;
    if (valu)
        valu_fma_body(s, iters_valu, a + threadIdx.x * 1e-9, b);
    else
        mfma16_body(s, iters_mfma, a + threadIdx.x * 1e-9, b);
    if (s == 0.12345)
        sink[0] = s;
}

// lane layout of v_mfma_f64_4x4x4_4b_f64: A one-hot in lane la, B one-hot in
// lane lb -> out[la * 64 + lb] = bit mask of the lanes whose result is 1
__global__ void
probe4_kernel(unsigned long long *out)
{
    int const lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double const a = lane == la ? 1.0 : 0.0;
            double const b = lane == lb ? 1.0 : 0.0;
            double const c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            unsigned long long const m = __ballot(c != 0.0);
            if (lane == 0)
                out[la * 64 + lb] = m;
        }
}

// same for v_mfma_f64_16x16x4_f64 (4 result registers): out[(la*64+lb)*4 + r]
__global__ void
probe16_kernel(unsigned long long *out)
{
    int const lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double const a = lane == la ? 1.0 : 0.0;
            double const b = lane == lb ? 1.0 : 0.0;
            double4_t c = { 0, 0, 0, 0 };
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r) {
                unsigned long long const m = __ballot(c[r] != 0.0);
                if (lane == 0)
                    out[(la * 64 + lb) * 4 + r] = m;
            }
        }
}

template <typename F>
static int
time_ms(F launch, int reps, float *best_ms)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0, nullptr));
        launch();
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best)
            best = ms;
    }
    CHECK(hipGetLastError());
    *best_ms = best;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

// out[]: see tools/peaks.py for the meaning of every slot
extern "C" int
peaks_run(int device, double *out, int n_out, unsigned long long *probe4,
    unsigned long long *probe16)
{
    if (n_out < 32)
        return -2;
    CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, device));
    int const cus = prop.multiProcessorCount;
    double const clock_ghz = prop.clockRate * 1e-6;
    out[0] = cus;
    out[1] = clock_ghz;

    // ---- HBM ----
    size_t const bytes = (size_t)1 << 30;
    size_t const n4 = bytes / sizeof(float4);
    float4 *src = nullptr, *dst = nullptr;
    float *sinkf = nullptr;
    double *sink = nullptr;
    CHECK(hipMalloc((void **)&src, bytes));
    CHECK(hipMalloc((void **)&dst, bytes));
    CHECK(hipMalloc((void **)&sinkf, 64));
    CHECK(hipMalloc((void **)&sink, 64));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMemset(dst, 0, bytes));
    int const grid = cus * 16;
    float ms;
    if (time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0,
            nullptr, src, dst, n4); }, 10, &ms))
        return -1;
    out[2] = 2.0 * bytes / (ms * 1e-3) / 1e9;    // copy GB/s (read + write)
    if (time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0,
            nullptr, src, sinkf, n4); }, 10, &ms))
        return -1;
    out[3] = 1.0 * bytes / (ms * 1e-3) / 1e9;    // read GB/s
    if (time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0,
            nullptr, dst, n4, 1.0f); }, 10, &ms))
        return -1;
    out[4] = 1.0 * bytes / (ms * 1e-3) / 1e9;    // write GB/s
    (void)hipFree(src);
    (void)hipFree(dst);

    // ---- FP64 VALU FMA: waves per SIMD = 1, 2, 4 (64-thread blocks) ----
    int const iters = 2048;
    for (int k = 0; k < 3; ++k) {
        int const wps = 1 << k;
        int const blocks = cus * 4 * wps;
        if (time_ms([&] { hipLaunchKernelGGL(valu_fma_kernel, dim3(blocks),
                dim3(64), 0, nullptr, sink, iters, 1.0, 0.999999); }, 5, &ms))
            return -1;
        double const flops = (double)blocks * 64 * iters * 64 * 2.0;
        out[5 + k] = flops / (ms * 1e-3) / 1e12;  // TFLOP/s
    }
    // dependent chain: cycles per FMA at 1 wave per SIMD
    {
        int const blocks = cus * 4;
        if (time_ms([&] { hipLaunchKernelGGL(valu_chain_kernel, dim3(blocks),
                dim3(64), 0, nullptr, sink, iters, 1.0, 0.999999); }, 5, &ms))
            return -1;
        out[8] = ms * 1e-3 * clock_ghz * 1e9 / ((double)iters * 64);
    }
    // v_rcp_f64: cycles per instruction per SIMD (4 independent, 1 wave/SIMD)
    {
        int const blocks = cus * 4;
        if (time_ms([&] { hipLaunchKernelGGL(valu_rcp_kernel, dim3(blocks),
                dim3(64), 0, nullptr, sink, iters, 1.5); }, 5, &ms))
            return -1;
        out[9] = ms * 1e-3 * clock_ghz * 1e9 / ((double)iters * 64);
    }
    // ---- FP64 MFMA 16x16x4: 1, 2 waves per SIMD ----
    for (int k = 0; k < 2; ++k) {
        int const wps = 1 << k;
        int const blocks = cus * 4 * wps;
        if (time_ms([&] { hipLaunchKernelGGL(mfma16_kernel, dim3(blocks),
                dim3(64), 0, nullptr, sink, iters, 1.0, 0.5); }, 5, &ms))
            return -1;
        double const n_mfma = (double)blocks * iters * 16;
        out[10 + k] = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
        if (k == 0)   // cycles per MFMA per SIMD
            out[12] = ms * 1e-3 * clock_ghz * 1e9 / ((double)iters * 16);
    }
    {
        int const blocks = cus * 4;
        if (time_ms([&] { hipLaunchKernelGGL(mfma16_chain_kernel, dim3(blocks),
                dim3(64), 0, nullptr, sink, iters, 1.0, 0.5); }, 5, &ms))
            return -1;
        out[13] = ms * 1e-3 * clock_ghz * 1e9 / ((double)iters * 16);
    }
    // ---- FP64 MFMA 4x4x4 (4 blocks): 512 flop per instruction ----
    for (int k = 0; k < 2; ++k) {
        int const wps = 1 << k;
        int const blocks = cus * 4 * wps;
        if (time_ms([&] { hipLaunchKernelGGL(mfma4_kernel, dim3(blocks),
                dim3(64), 0, nullptr, sink, iters, 1.0, 0.5); }, 5, &ms))
            return -1;
        double const n_mfma = (double)blocks * iters * 16;
        out[14 + k] = n_mfma * 512.0 / (ms * 1e-3) / 1e12;
        if (k == 0)
            out[16] = ms * 1e-3 * clock_ghz * 1e9 / ((double)iters * 16);
    }
    {
        int const blocks = cus * 4;
        if (time_ms([&] { hipLaunchKernelGGL(mfma4_chain_kernel, dim3(blocks),
                dim3(64), 0, nullptr, sink, iters, 1.0, 0.5); }, 5, &ms))
            return -1;
        out[17] = ms * 1e-3 * clock_ghz * 1e9 / ((double)iters * 16);
    }
    // ---- VALU wave + MFMA wave on the same SIMD ----
    {
        // work sized so each kind alone takes about the same time
        int const iv = 2048, im = 2048 / 4;   // 64 FMA x iv  vs 16 MFMA x im
        float t_mixed, t_valu, t_mfma;
        if (time_ms([&] { hipLaunchKernelGGL(mixed_kernel, dim3(cus), dim3(512),
                0, nullptr, sink, iv, im, 0, 1.0, 0.999999); }, 5, &t_mixed))
            return -1;
        // the same VALU work alone: 4 VALU waves + 4 idle-exit MFMA waves
        if (time_ms([&] { hipLaunchKernelGGL(mixed_kernel, dim3(cus), dim3(512),
                0, nullptr, sink, iv, 0, 0, 1.0, 0.999999); }, 5, &t_valu))
            return -1;
        if (time_ms([&] { hipLaunchKernelGGL(mixed_kernel, dim3(cus), dim3(512),
                0, nullptr, sink, 0, im, 0, 1.0, 0.999999); }, 5, &t_mfma))
            return -1;
        out[18] = t_mixed;
        out[19] = t_valu;
        out[20] = t_mfma;
    }
    // ---- layout probes ----
    if (probe4 != nullptr) {
        unsigned long long *d = nullptr;
        CHECK(hipMalloc((void **)&d, 4096 * sizeof(unsigned long long)));
        hipLaunchKernelGGL(probe4_kernel, dim3(1), dim3(64), 0, nullptr, d);
        CHECK(hipMemcpy(probe4, d, 4096 * sizeof(unsigned long long),
            hipMemcpyDeviceToHost));
        (void)hipFree(d);
    }
    if (probe16 != nullptr) {
        unsigned long long *d = nullptr;
        CHECK(hipMalloc((void **)&d, 4 * 4096 * sizeof(unsigned long long)));
        hipLaunchKernelGGL(probe16_kernel, dim3(1), dim3(64), 0, nullptr, d);
        CHECK(hipMemcpy(probe16, d, 4 * 4096 * sizeof(unsigned long long),
            hipMemcpyDeviceToHost));
        (void)hipFree(d);
    }
    (void)hipFree(sinkf);
    (void)hipFree(sink);
    return 0;
}
