// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 per SIMD and the clock the chip sustains,
// for a given number of resident workgroups (1 wave per SIMD).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, long long *cyc) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv) {
    int iters = 20000;
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grids[] = {64, 128, 256, 512};
    for (int g : grids) {
        hipLaunchKernelGGL(k_mfma, dim3(g), dim3(256), 0, 0, out, 100, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma, dim3(g), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double mfmas = (double)iters * 32;
        double flops = mfmas * 2048.0 * 4 * g;
        printf("grid %4d: %.3f ms, %.1f TFLOP/s, %.2f us per 256 MFMA, counter ticks/MFMA %.2f (ticks %lld)\n", g, ms,
               flops / ms / 1e9, ms * 1e3 / (mfmas / 256), (double)c / mfmas, c);
    }
    return 0;
}
