// A co-runner that keeps the HBM busy: streaming copies between two 2 GiB buffers for <seconds> (default 60), on every CU it gets.
// Used by tools/gpu/contended_sample.sh (the round-2 probability excursion was only ever sampled with the GPU to itself).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_hog.hip -o tools/ubench/hbm_hog
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *src, uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 60.0;
    const int blocks = argc > 2 ? atoi(argv[2]) : 512;          // 2 per CU: leaves the other processes their share of every CU
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(uint4);
    uint4 *a = nullptr, *b = nullptr;
    if (hipMalloc((void **)&a, bytes) != hipSuccess || hipMalloc((void **)&b, bytes) != hipSuccess) { fprintf(stderr, "hbm_hog: hipMalloc failed\n"); return 1; }
    (void)hipMemset(a, 1, bytes);
    const auto t0 = std::chrono::steady_clock::now();
    size_t passes = 0;
    for (;;) {
        for (int i = 0; i < 8; ++i) {
            hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n);
            ++passes;
        }
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "hbm_hog: kernel failed\n"); return 1; }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt >= seconds) { printf("hbm_hog: %zu passes of 2 GiB read + 2 GiB written in %.1f s = %.2f TB/s\n", passes, dt, passes * 2.0 * bytes / dt / 1e12); break; }
    }
    return 0;
}
