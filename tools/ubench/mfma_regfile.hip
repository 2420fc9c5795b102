// Micro-benchmark: does the register file an MFMA operand lives in (architectural VGPR "v" vs accumulation
// VGPR "a") change how many VALU instructions hide in the MFMA's shadow?  v_mfma_f32_32x32x16_f16, one wave per
// SIMD, one dependent accumulator chain, NF fillers (alternating v_fma_f32 / v_exp_f32) after every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NF>
__global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc) {
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    float v[8];
    f16x8 ah, bh, aw[8], bw[8];
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); }
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) { aw[q][i] = (_Float16)(a + i + q); bw[q][i] = (_Float16)(b - i - q); }
    for (int i = 0; i < 8; ++i) v[i] = a * 0.01f + i * 0.001f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(ah), "v"(bh));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bh));
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(ah), "v"(bh));
            if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(ah), "v"(bh));
            if (MODE == 5) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(aw[m & 7]), "v"(bw[(m >> 1) & 7]));
            if (MODE == 6) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(aw[m & 7]), "v"(bw[(m >> 1) & 7]));
            if (MODE == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(ah), "a"(bh));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int r = (m * NF + f) & 7;
                if (f & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int NF> void run() {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL((k<MODE, NF>), dim3(256), dim3(256), 0, 0, out, 10, cyc);
    hipLaunchKernelGGL((k<MODE, NF>), dim3(256), dim3(256), 0, 0, out, 2000, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf(" %6.1f", (double)c / 2000 / 32);
    (void)hipFree(out); (void)hipFree(cyc);
}
template <int MODE> void row(const char *name) {
    printf("%-44s", name);
    run<MODE, 0>(); run<MODE, 2>(); run<MODE, 4>(); run<MODE, 5>(); run<MODE, 6>();
    printf("\n");
}
int main() {
    printf("cycles per (MFMA + NF fillers), NF =                0      2      4      5      6\n");
    row<0>("acc AGPR, A VGPR, B VGPR");
    row<1>("acc VGPR, A VGPR, B VGPR");
    row<2>("acc VGPR, A AGPR, B VGPR");
    row<3>("acc AGPR, A AGPR, B VGPR");
    row<4>("acc AGPR, A AGPR, B AGPR");
    row<5>("acc VGPR, A 8 AGPR quads, B 8 VGPR quads");
    row<6>("acc VGPR, A 8 VGPR quads, B 8 VGPR quads");
    return 0;
}
