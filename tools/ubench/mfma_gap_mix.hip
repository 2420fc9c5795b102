// Micro-benchmark: what one gap of the recurrent kernels costs the wave -- v_mfma_f32_32x32x16_f16 (inline asm, A in AGPRs, one accumulator chain
// as in lstm32.hip.h) followed by NE v_exp_f32 + NF v_fma_f32 + NP v_pk_fma_f32 (two fp32 per lane), all independent of each other.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_gap_mix.hip -o tools/ubench/mfma_gap_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NE, int NF, int NP, bool MFMA>
__global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc) {
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    float v[8], u[8];
    f32x2 p[4], pb = {b, a};
    f16x8 aw[8], bw[8];
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) { aw[q][i] = (_Float16)(a + i + q); bw[q][i] = (_Float16)(b - i - q); }
    for (int i = 0; i < 8; ++i) { v[i] = a * 0.01f + i * 0.001f; u[i] = v[i] + 1.0f; }
    for (int i = 0; i < 4; ++i) p[i] = (f32x2){v[i], u[i]};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(aw[m & 7]), "v"(bw[(m >> 1) & 7]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NE; ++f) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(m * NE + f) & 7]));
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(u[(m * NF + f) & 7]) : "v"(b));
#pragma unroll
            for (int f = 0; f < NP; ++f) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(m * NP + f) & 3]) : "v"(pb));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + u[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NE, int NF, int NP> void run() {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    double r[2];
    for (int mf = 0; mf < 2; ++mf) {
        if (mf) { hipLaunchKernelGGL((k<NE, NF, NP, true>), dim3(256), dim3(256), 0, 0, out, 10, cyc); hipLaunchKernelGGL((k<NE, NF, NP, true>), dim3(256), dim3(256), 0, 0, out, 2000, cyc); }
        else { hipLaunchKernelGGL((k<NE, NF, NP, false>), dim3(256), dim3(256), 0, 0, out, 10, cyc); hipLaunchKernelGGL((k<NE, NF, NP, false>), dim3(256), dim3(256), 0, 0, out, 2000, cyc); }
        (void)hipDeviceSynchronize();
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        r[mf] = (double)c / 2000 / 32;
    }
    printf("%d exp + %d fma + %d pk_fma:  alone %6.1f   after an MFMA %6.1f\n", NE, NF, NP, r[0], r[1]);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    printf("cycles per gap (s_memtime-free: readcyclecounter over 64 000 gaps), fillers alone | MFMA + fillers\n");
    run<0, 0, 0>(); run<0, 4, 0>(); run<0, 0, 2>(); run<0, 0, 4>(); run<0, 2, 1>();
    run<2, 0, 0>(); run<3, 0, 0>(); run<2, 2, 0>(); run<3, 1, 0>(); run<2, 0, 1>(); run<3, 0, 1>(); run<2, 0, 2>(); run<1, 0, 1>(); run<1, 0, 2>(); run<2, 1, 1>();
    return 0;
}
