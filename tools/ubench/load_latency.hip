// Latency of a fresh (never touched) global_load_dwordx4 per wave, as the recurrent kernel issues them: each workgroup walks its
// own contiguous region in 64 KiB steps; 64 workgroups x 256 threads; one 4 x 1 KiB load group per wave per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int STORES>
__global__ __launch_bounds__(256) void k(const float *src, float *dst, long long *out, int steps, size_t wg_stride, size_t step_stride) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float *p = src + blockIdx.x * wg_stride + w * 4096 + lane * 4;
    float *q = dst + blockIdx.x * wg_stride + w * 4096 + lane * 4;
    f32x4 acc = {0, 0, 0, 0};
    long long tot = 0, mx = 0;
    for (int s = 0; s < steps; ++s) {
        const float *ps = p + s * step_stride;
        long long t0 = __builtin_readcyclecounter();
        f32x4 a = *(const f32x4 *)ps, b = *(const f32x4 *)(ps + 256), c = *(const f32x4 *)(ps + 512), d = *(const f32x4 *)(ps + 768);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        long long t1 = __builtin_readcyclecounter();
        acc += a + b + c + d;
        if (STORES) { float *qs = q + s * step_stride; *(f32x4 *)qs = acc; *(f32x4 *)(qs + 256) = acc; }
        if (s > 0) { tot += t1 - t0; if (t1 - t0 > mx) mx = t1 - t0; }
        __builtin_amdgcn_s_sleep(64);   // ~4000 cycles of "compute" between steps
        __builtin_amdgcn_s_sleep(64);
    }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = tot / (steps - 1); out[blockIdx.x * 2 + 1] = mx; }
    if (acc[0] == 12345.f) dst[0] = acc[1];
}
template <int STORES> void run(const char *name, size_t wg_stride, size_t step_stride) {
    const int wgs = 64, steps = 33;
    const size_t n = (size_t)64 * 33 * 16384 * 2;
    float *src, *dst; long long *out;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&dst, n * 4); (void)hipMalloc(&out, wgs * 16);
    (void)hipMemset(src, 0, n * 4);
    hipLaunchKernelGGL((k<STORES>), dim3(wgs), dim3(256), 0, 0, src, dst, out, steps, wg_stride, step_stride);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(wgs * 2);
    (void)hipMemcpy(h.data(), out, wgs * 16, hipMemcpyDeviceToHost);
    long long a = 0, m = 0;
    for (int i = 0; i < wgs; ++i) { a += h[2 * i]; if (h[2 * i + 1] > m) m = h[2 * i + 1]; }
    printf("%-58s mean %6lld cycles, worst %6lld\n", name, a / wgs, m);
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(out);
}
int main() {
    run<0>("tile-major (wg stride 33*64 KiB, step 64 KiB), loads only", (size_t)33 * 16384, 16384);
    run<1>("tile-major, loads + 2 stores per step", (size_t)33 * 16384, 16384);
    run<0>("t-major (wg stride 64 KiB, step 64*64 KiB), loads only", 16384, (size_t)64 * 16384);
    run<1>("t-major, loads + 2 stores per step", 16384, (size_t)64 * 16384);
    return 0;
}
