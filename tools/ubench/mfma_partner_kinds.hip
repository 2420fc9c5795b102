// Micro-benchmark: which kind of partner-wave instruction slows an fp32-MFMA wave on the same SIMD?
// 512 threads: waves 0-3 issue MFMAs (distinct A/B registers like the LSTM kernel), waves 4-7 run KIND.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k2(float *out, float *gbuf, int iters, long long *cyc) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 0.001f;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    float s = 0;
    if (wave < 4) {
        f32x4 acc[4];
        float a[32], b[128];
        for (int i = 0; i < 32; ++i) a[i] = lds[(lane + i * 64) & 8191];
        for (int i = 0; i < 128; ++i) b[i] = lds[(lane * 3 + i * 64) & 8191];
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[g * 32 + k], acc[g], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
        for (int it = 0; it < iters; ++it) {
            if (KIND == 1) {
#pragma unroll
                for (int m = 0; m < 16; ++m) { f32x4 x = *(const f32x4 *)&lds[((lane + m * 64) * 4) & 8191]; v[m & 7] += x[0] + x[3]; }
            } else if (KIND == 2) {
#pragma unroll
                for (int m = 0; m < 16; ++m) lds[(lane + m * 64 + it) & 8191] = v[m & 7];
            } else if (KIND == 3) {
#pragma unroll
                for (int m = 0; m < 4; ++m) *(f32x4 *)&gbuf[(size_t)(((blockIdx.x * 4 + wave - 4) * 4 + m) * 64 + lane) * 4] = (f32x4){v[0], v[1], v[2], v[3]};
            } else if (KIND == 4) {
#pragma unroll
                for (int m = 0; m < 40; ++m) v[m & 7] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[m & 7] * 0.001f));
            } else if (KIND == 5) {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m & 7] += (float)(__builtin_readcyclecounter() & 1);
            }
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char *name) {
    float *out, *gbuf; long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&gbuf, (size_t)256 * 16 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 1000;
    hipLaunchKernelGGL((k2<KIND>), dim3(256), dim3(512), 0, 0, out, gbuf, 10, cyc);
    hipLaunchKernelGGL((k2<KIND>), dim3(256), dim3(512), 0, 0, out, gbuf, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-44s MFMA wave: %8.1f cycles per 128 MFMA | partner: %8.1f cycles per iteration\n", name, (double)c[0] / iters, (double)c[4] / iters);
    (void)hipFree(out); (void)hipFree(gbuf); (void)hipFree(cyc);
}
int main() {
    run<0>("partner idle");
    run<1>("partner 16 ds_read_b128 / iter");
    run<2>("partner 16 ds_write_b32 / iter");
    run<3>("partner 4 global_store_dwordx4 / iter");
    run<4>("partner 40 x (mul, exp, add, rcp) / iter");
    run<5>("partner 8 s_memtime / iter");
    return 0;
}
