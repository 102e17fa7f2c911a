// Probe for the XCD-aware all-reduce of the resident PCG (cg_resident.hip):
//  (A) is workgroup b of a 512-thread, one-per-CU launch placed on XCD b % 8?
//      (s_getreg_b32 HW_REG_XCC_ID, MI355X_MICROARCH.md "Workgroup dispatch")
//  (B) what does a one-way hand-off of a tagged 16-byte granule pair cost between
//      two workgroups of the SAME XCD through its L2 (plain store, sc1 load) and
//      between XCDs (sc1 store, sc1 load), with every CU of the chip taking part?
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_probe tools/xcd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int uint4_r __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ unsigned xcc_id(void)
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

__global__ void __launch_bounds__(512)
placement_kernel(int *xcc)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1.0;
        xcc[blockIdx.x] = (int)xcc_id();
    }
}

__global__ void __launch_bounds__(512)
spin_kernel(long long *sink, long long ticks)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) {
        lds[0] = 0.0;
        long long const t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < ticks)
            __builtin_amdgcn_s_sleep(8);
        if (ticks < 0)
            *sink = t0;
    }
}

// Ping-pong between workgroup b and its partner: `rounds` exchanges; the
// initiator (lower id) stores tag k, the partner answers with tag k; the
// initiator measures the round trip.  mode 0: sc1 store + sc1 load; mode 1:
// plain store + sc1 load (correct only when both share an XCD's L2).
__global__ void __launch_bounds__(512)
pingpong_kernel(unsigned *slots, int partner_stride, int mode, int rounds, long long *cycles,
    int *fails)
{
    extern __shared__ double lds[];
    if (threadIdx.x != 0)
        return;
    lds[0] = 0.0;
    int const b = blockIdx.x;
    int const group = b / (2 * partner_stride), within = b % (2 * partner_stride);
    bool const initiator = within < partner_stride;
    int const partner = initiator ? b + partner_stride : b - partner_stride;
    (void)group;
    __amdgpu_buffer_rsrc_t const buf = __builtin_amdgcn_make_buffer_rsrc(slots, 0,
        (int)(gridDim.x * 64), 0x00020000);
    auto store = [&](int who, unsigned tag) {
        uint4_r const w = { tag, tag, tag, tag };
        if (mode == 0)
            __builtin_amdgcn_raw_buffer_store_b128(w, buf, who * 64, 0, AUX_SC1);
        else
            __builtin_amdgcn_raw_buffer_store_b128(w, buf, who * 64, 0, 0);
    };
    auto wait = [&](int who, unsigned tag) -> bool {
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
            uint4_r const w = __builtin_amdgcn_raw_buffer_load_b128(buf, who * 64, 0, AUX_SC1);
            if (w.x == tag && w.w == tag)
                return true;
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        return false;
    };
    long long t0 = 0;
    bool ok = true;
    for (int k = 1; k <= rounds && ok; ++k) {
        if (initiator) {
            if (k == 3)
                t0 = (long long)wall_clock64();
            store(b, (unsigned)k);            // my slot, read by the partner
            ok = wait(partner, (unsigned)k);  // its answer
        } else {
            ok = wait(partner, (unsigned)k);
            store(b, (unsigned)k);
        }
    }
    if (initiator)
        cycles[b] = ok ? (long long)wall_clock64() - t0 : -1;
    if (!ok)
        atomicAdd(fails, 1);
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    int const blocks = 256;
    size_t const lds = 100 * 1024;
    int *xcc = nullptr;
    CHECK(hipMalloc(&xcc, blocks * sizeof(int)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(placement_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pingpong_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int grid : { 256, 64, 24 }) {
        int mismatches = 0, launches = 0;
        for (int rep = 0; rep < 50; ++rep) {
            hipLaunchKernelGGL(placement_kernel, dim3(grid), dim3(512), lds, 0, xcc);
            std::vector<int> h(grid);
            CHECK(hipMemcpy(h.data(), xcc, grid * sizeof(int), hipMemcpyDeviceToHost));
            for (int b = 0; b < grid; ++b)
                mismatches += h[b] != b % 8 ? 1 : 0;
            launches += 1;
            if (rep == 0) {
                std::printf("grid %3d, first launch, XCC of blocks 0..15:", grid);
                for (int b = 0; b < 16 && b < grid; ++b)
                    std::printf(" %d", h[b]);
                std::printf("\n");
            }
        }
        std::printf("grid %3d: %d workgroups off the round-robin placement in %d launches\n", grid,
            mismatches, launches);
    }
    // (C) the round-robin pointer is not reset per launch: after a launch whose
    // grid is no multiple of 8 the next one starts on another XCD.  Is the
    // placement then still a rotation (every aligned group of 8 consecutive
    // workgroups on 8 different XCDs)?  Also with a kernel of another stream
    // holding part of the chip.
    {
        hipStream_t other;
        CHECK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
        long long *sink = nullptr;
        CHECK(hipMalloc(&sink, sizeof(long long)));
        for (int busy : { 0, 1 }) {
            int rotated = 0, broken = 0, launches = 0;
            for (int rep = 0; rep < 60; ++rep) {
                int const odd = 1 + rep % 13;
                hipLaunchKernelGGL(placement_kernel, dim3(odd), dim3(64), 1024, 0, xcc);
                if (busy)
                    hipLaunchKernelGGL(spin_kernel, dim3(40 + rep), dim3(512), lds, other, sink,
                        200000LL);
                int const grid = rep % 2 ? 256 : 64;
                hipLaunchKernelGGL(placement_kernel, dim3(grid), dim3(512), lds, 0, xcc);
                std::vector<int> h(grid);
                CHECK(hipMemcpy(h.data(), xcc, grid * sizeof(int), hipMemcpyDeviceToHost));
                CHECK(hipDeviceSynchronize());
                bool rotation = true, groups_ok = true;
                for (int b = 0; b < grid; ++b)
                    rotation = rotation && h[b] == (h[0] + b) % 8;
                for (int g = 0; g < grid / 8; ++g) {
                    unsigned seen = 0;
                    for (int j = 0; j < 8; ++j)
                        seen |= 1u << h[g * 8 + j];
                    groups_ok = groups_ok && seen == 0xFFu;
                }
                rotated += rotation && h[0] != 0 ? 1 : 0;
                broken += groups_ok ? 0 : 1;
                launches += 1;
                if (rep < 3 || !groups_ok) {
                    std::printf("  after a grid of %2d%s: XCC of blocks 0..15:", odd,
                        busy ? " (other stream busy)" : "");
                    for (int b = 0; b < 16; ++b)
                        std::printf(" %d", h[b]);
                    std::printf("%s\n", groups_ok ? "" : "  <- not a permutation per group of 8");
                }
            }
            std::printf("%s: %d launches, %d started on another XCD than 0, %d with a group of 8 "
                "that does not cover the 8 XCDs\n", busy ? "other stream busy" : "idle chip",
                launches, rotated, broken);
        }
    }
    unsigned *slots = nullptr;
    long long *cycles = nullptr;
    int *fails = nullptr;
    CHECK(hipMalloc(&slots, blocks * 64));
    CHECK(hipMalloc(&cycles, blocks * sizeof(long long)));
    CHECK(hipMalloc(&fails, sizeof(int)));
    int const rounds = 203;
    struct Case { const char *name; int stride, mode; };
    Case const cases[] = { { "cross-XCD (b, b+1), sc1 store + sc1 load", 1, 0 },
        { "same XCD (b, b+8), sc1 store + sc1 load", 8, 0 },
        { "same XCD (b, b+8), plain store + sc1 load", 8, 1 },
        { "cross-XCD (b, b+1), plain store + sc1 load (expected to fail / stall)", 1, 1 } };
    for (Case const &c : cases) {
        CHECK(hipMemset(slots, 0, blocks * 64));
        CHECK(hipMemset(fails, 0, sizeof(int)));
        CHECK(hipMemset(cycles, 0, blocks * sizeof(long long)));
        hipLaunchKernelGGL(pingpong_kernel, dim3(blocks), dim3(512), lds, 0, slots, c.stride,
            c.mode, rounds, cycles, fails);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> h(blocks);
        int f = 0;
        CHECK(hipMemcpy(h.data(), cycles, blocks * sizeof(long long), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&f, fails, sizeof(int), hipMemcpyDeviceToHost));
        std::vector<double> us;
        for (int b = 0; b < blocks; ++b)
            if (h[b] > 0)
                us.push_back((double)h[b] / 100.0 / (rounds - 2) / 2.0);   // one way
        std::sort(us.begin(), us.end());
        if (us.empty())
            std::printf("%-70s: no pair finished (%d workgroups gave up)\n", c.name, f);
        else
            std::printf("%-70s: one way min %.2f median %.2f max %.2f us (%zu pairs, %d gave up)\n",
                c.name, us.front(), us[us.size() / 2], us.back(), us.size(), f);
    }
    return 0;
}
