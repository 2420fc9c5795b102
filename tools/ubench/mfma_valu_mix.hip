// Micro-benchmark: does VALU / transcendental work issued between fp32 MFMAs of the SAME wave cost
// matrix-pipe time?  One wave per SIMD; per 32 MFMAs insert NV v_fma_f32 and NT v_exp_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NT, int NM, int F16>
__global__ __launch_bounds__(256) void k_mix(float *out, int iters, long long *cyc) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    float v[8], t[8];
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); }
    for (int i = 0; i < 8; ++i) { v[i] = a + i; t[i] = b * 0.01f + i * 0.001f; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (m < NM) acc[m & 7] = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[m & 7], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 7], 0, 0, 0);
            if (m < NV) v[m & 7] = __builtin_fmaf(v[m & 7], 0.999f, 0.001f);
            if (m < NT) t[m & 7] = __builtin_amdgcn_exp2f(t[m & 7]) * 0.0f + t[m & 7];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + t[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NT, int NM, int F16>
void run(const char *name) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k_mix<NV, NT, NM, F16>), dim3(256), dim3(256), 0, 0, out, 10, cyc);
    hipLaunchKernelGGL((k_mix<NV, NT, NM, F16>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.1f cycles per group of (%d MFMA + %d v_fma + %d x (v_exp,v_fma))\n", name, (double)c / iters, NM, NV, NT);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0, 0, 32, 0>("fp32: 32 MFMA");
    run<32, 0, 32, 0>("fp32: 32 MFMA + 32 fma");
    run<0, 16, 32, 0>("fp32: 32 MFMA + 16 exp(+fma)");
    run<0, 0, 32, 1>("f16 : 32 MFMA");
    run<32, 0, 32, 1>("f16 : 32 MFMA + 32 fma");
    run<32, 0, 0, 1>("      32 fma alone");
    run<0, 16, 32, 1>("f16 : 32 MFMA + 16 exp(+fma)");
    run<0, 32, 32, 1>("f16 : 32 MFMA + 32 exp(+fma)");
    run<16, 8, 32, 1>("f16 : 32 MFMA + 16 fma + 8 exp(+fma)");
    return 0;
}
