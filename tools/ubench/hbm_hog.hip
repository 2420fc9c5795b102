// A co-runner that keeps the L2 <-> fabric <-> HBM path busy at a chosen intensity, and its control.
//   hbm_hog <seconds> <blocks> [hbm|l2]
//     hbm : streaming copies between two 2 GiB buffers on <blocks> workgroups of 256 threads (16 B per thread and iteration): every byte
//           crosses the fabric twice (read + write).  The rate is set by the number of workgroups; the achieved TB/s is printed at the end.
//     l2  : THE SAME instruction stream (same loads, stores and address arithmetic per byte) on a private 32 KiB window per workgroup that
//           stays in the CU's L1 / the XCD's L2: the CU slots, issue cycles and most of the watts of the hog without its fabric traffic --
//           what separates "the pipeline lost throughput to the hog's bandwidth" from "... to its power and CU time" (tools/gpu/fabric_sensitivity.sh).
// Used by tools/gpu/contended_sample.sh (rounds 3-4) and tools/gpu/fabric_sensitivity.sh (round 6).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_hog.hip -o tools/ubench/hbm_hog
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *src, uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// n = 16-byte elements per workgroup and pass (the same count the hbm mode moves per workgroup); window = elements of the private region
__global__ __launch_bounds__(256) void copy_window_kernel(const uint4 *src, uint4 *dst, size_t n, size_t window) {
    const uint4 *s = src + (size_t)blockIdx.x * window;
    uint4 *d = dst + (size_t)blockIdx.x * window;
    for (size_t i = threadIdx.x; i < n; i += 256) d[i & (window - 1)] = s[i & (window - 1)];
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 60.0;
    const int blocks = argc > 2 ? atoi(argv[2]) : 512;          // 2 per CU: leaves the other processes their share of every CU
    const bool l2 = argc > 3 && !strcmp(argv[3], "l2");
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(uint4);
    uint4 *a = nullptr, *b = nullptr;
    if (hipMalloc((void **)&a, bytes) != hipSuccess || hipMalloc((void **)&b, bytes) != hipSuccess) { fprintf(stderr, "hbm_hog: hipMalloc failed\n"); return 1; }
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    const size_t window = 1024;                                  // 16 KiB read + 16 KiB written per workgroup
    const auto t0 = std::chrono::steady_clock::now();
    size_t passes = 0;
    for (;;) {
        for (int i = 0; i < 8; ++i) {
            if (l2) hipLaunchKernelGGL(copy_window_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n / blocks, window);
            else hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n);
            ++passes;
        }
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "hbm_hog: kernel failed\n"); return 1; }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt >= seconds) {
            printf("hbm_hog %s blocks %d: %zu passes of 2 GiB read + 2 GiB written in %.1f s = %.2f TB/s %s\n", l2 ? "l2" : "hbm", blocks, passes, dt,
                   passes * 2.0 * bytes / dt / 1e12, l2 ? "through the L1 / L2 (no fabric traffic)" : "through the fabric");
            break;
        }
    }
    return 0;
}
