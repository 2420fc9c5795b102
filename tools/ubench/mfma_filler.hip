// Micro-benchmark: how many VALU "fillers" hide in the shadow of one MFMA when both come from the SAME wave
// (one wave per SIMD)?  Shapes: 0 = v_mfma_f32_16x16x4_f32, 1 = v_mfma_f32_16x16x32_f16, 2 = v_mfma_f32_32x32x16_f16.
// KIND: 0 = v_fma_f32, 1 = v_exp_f32.  NF fillers are placed after every MFMA (sched_barrier keeps the order).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NF, int KIND>
__global__ __launch_bounds__(256) void k_fill(float *out, int iters, long long *cyc) {
    f32x4 acc[8];
    f32x16 big[4];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    float v[8];
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); }
    for (int i = 0; i < 8; ++i) v[i] = a * 0.01f + i * 0.001f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (SHAPE == 0) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 7], 0, 0, 0);
            if (SHAPE == 1) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[m & 7], 0, 0, 0);
            if (SHAPE == 2) big[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, big[m & 3], 0, 0, 0);
            if (SHAPE == 3) big[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, big[0], 0, 0, 0);   // one dependent chain
            if (SHAPE == 4) big[(m >> 3) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, big[(m >> 3) & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int r = (m * NF + f) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(b));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else if (KIND == 2) { if (f & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r])); else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(b)); }
                else if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[r]));
                else if (KIND == 4) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[r]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += big[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int NF, int KIND>
void run() {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k_fill<SHAPE, NF, KIND>), dim3(256), dim3(256), 0, 0, out, 10, cyc);
    hipLaunchKernelGGL((k_fill<SHAPE, NF, KIND>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf(" %6.1f", (double)c / iters / 32);
    (void)hipFree(out); (void)hipFree(cyc);
}
template <int SHAPE, int KIND> void row(const char *name) {
    printf("%-40s", name);
    run<SHAPE, 0, KIND>(); run<SHAPE, 1, KIND>(); run<SHAPE, 2, KIND>(); run<SHAPE, 3, KIND>(); run<SHAPE, 4, KIND>();
    run<SHAPE, 6, KIND>(); run<SHAPE, 8, KIND>();
    printf("\n");
}
int main() {
    printf("cycles per (MFMA + NF fillers), NF =            0      1      2      3      4      6      8\n");
    row<0, 0>("16x16x4 f32   + v_fma_f32");
    row<0, 1>("16x16x4 f32   + v_exp_f32");
    row<1, 0>("16x16x32 f16  + v_fma_f32");
    row<1, 1>("16x16x32 f16  + v_exp_f32");
    row<2, 0>("32x32x16 f16  + v_fma_f32");
    row<2, 1>("32x32x16 f16  + v_exp_f32");
    row<2, 2>("32x32x16 f16  + fma,exp alternating");
    row<2, 3>("32x32x16 f16  + v_rcp_f32");
    row<2, 4>("32x32x16 f16  + v_cvt_f16_f32");
    row<3, 0>("32x32x16 f16 one acc chain + v_fma_f32");
    row<3, 2>("32x32x16 f16 one acc chain + fma,exp");
    row<4, 2>("32x32x16 f16 chains of 8 + fma,exp");
    return 0;
}
