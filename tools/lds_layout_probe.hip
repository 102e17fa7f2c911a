// Does a four-double vector per lane cost more LDS time as one 32-byte record per lane
// (two ds_read_b128 whose lanes are 32 bytes apart) than as two planes of 16-byte halves
// (lanes 16 bytes apart)?  512 threads, every thread reads / rewrites its own vector and
// two neighbours' per round -- the access pattern of the resident PCG's direction tile.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_layout_probe tools/lds_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_r __attribute__((ext_vector_type(4)));
typedef double double2_r __attribute__((ext_vector_type(2)));
constexpr int N = 512, ROUNDS = 2000;

template <int MODE>
__global__ void __launch_bounds__(512) probe(double *out, long long *ticks)
{
    __shared__ __attribute__((aligned(16))) double buf[N * 4 + 64];
    int const t = threadIdx.x;
    for (int i = t; i < N * 4 + 64; i += 512)
        buf[i] = 1.0 / (1 + i);
    __syncthreads();
    double4_r acc = { 0, 0, 0, 0 };
    long long const t0 = (long long)wall_clock64();
    for (int r = 0; r < ROUNDS; ++r) {
        int const a = t, b = (t + 1) & (N - 1), c = (t + 31) & (N - 1);
        if (MODE == 0) {
            const double4_r *p = reinterpret_cast<const double4_r *>(buf);
            double4_r const x = p[a], y = p[b], z = p[c];
            acc += x + y + z;
            reinterpret_cast<double4_r *>(buf)[a] = acc * 1e-9 + x;
        } else {
            const double2_r *p = reinterpret_cast<const double2_r *>(buf);
            double2_r const x0 = p[a], x1 = p[N + a], y0 = p[b], y1 = p[N + b], z0 = p[c],
                            z1 = p[N + c];
            acc += (double4_r){ x0.x + y0.x + z0.x, x0.y + y0.y + z0.y, x1.x + y1.x + z1.x,
                x1.y + y1.y + z1.y };
            reinterpret_cast<double2_r *>(buf)[a] = (double2_r){ acc.x * 1e-9 + x0.x, acc.y * 1e-9 + x0.y };
            reinterpret_cast<double2_r *>(buf)[N + a] = (double2_r){ acc.z * 1e-9 + x1.x, acc.w * 1e-9 + x1.y };
        }
        __syncthreads();
    }
    long long const t1 = (long long)wall_clock64();
    if (t == 0)
        ticks[MODE] = t1 - t0;
    out[t] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    double *out;
    long long *ticks;
    hipMalloc(&out, 512 * sizeof(double));
    hipMalloc(&ticks, 2 * sizeof(long long));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<0>, dim3(1), dim3(512), 0, 0, out, ticks);
        hipLaunchKernelGGL(probe<1>, dim3(1), dim3(512), 0, 0, out, ticks);
        long long h[2];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        std::printf("one record of 32 bytes per lane: %.1f ns per round; two planes of 16 bytes: %.1f ns per round\n",
            h[0] * 10.0 / ROUNDS, h[1] * 10.0 / ROUNDS);
    }
    return 0;
}
