// Micro-benchmark: two waves per SIMD, one issuing only fp32 MFMAs, the other only VALU/transcendental
// work.  Do they overlap (time = max) or share one datapath (time = sum)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>  // 0: partner idle, 1: partner v_fma, 2: partner v_exp
__global__ __launch_bounds__(512) void k2(float *out, int iters, long long *cyc) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    long long t0 = __builtin_readcyclecounter();
    float s = 0;
    if (wave < 4) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 32; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 7], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (KIND == 1) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 256; ++m) v[m & 7] = __builtin_fmaf(v[m & 7], 0.999f, 0.001f);
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if (KIND == 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = b * 0.01f + i * 0.001f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 64; ++m) v[m & 7] = __builtin_amdgcn_exp2f(v[m & 7]) * 0.0f + v[m & 7];
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char *name) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k2<KIND>), dim3(256), dim3(512), 0, 0, out, 10, cyc);
    hipLaunchKernelGGL((k2<KIND>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-28s MFMA wave: %8.1f cycles per 32 MFMA | partner wave: %8.1f cycles per iteration\n", name,
           (double)c[0] / iters, (double)c[4] / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("partner idle");
    run<1>("partner 256 v_fma / iter");
    run<2>("partner 64 (v_exp+v_fma) / iter");
    return 0;
}
