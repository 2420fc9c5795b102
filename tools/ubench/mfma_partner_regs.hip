// Micro-benchmark: does the partner wave's VALU throughput depend on how many distinct VGPRs the MFMA
// wave's operands come from?  waves 0-3: 128 MFMAs/iter with NA distinct A registers and NB distinct B
// registers per accumulator; waves 4-7: 256 independent v_fma per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NA, int NB, int NACC>
__global__ __launch_bounds__(512) void k2(float *out, int iters, long long *cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = __builtin_readcyclecounter();
    float s = 0;
    if (wave < 4) {
        f32x4 acc[NACC];
        float a[NA], b[NB * NACC];
        for (int i = 0; i < NA; ++i) a[i] = lane * 0.001f + i;
        for (int i = 0; i < NB * NACC; ++i) b[i] = lane * 0.002f + i;
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 128 / NACC; ++k)
#pragma unroll
                for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k % NA], b[g * NB + (k % NB)], acc[g], 0, 0, 0);
            asm volatile("" : "+v"(acc[0]));
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 256; ++m) v[m & 7] = __builtin_fmaf(v[m & 7], 0.999f, 0.001f);
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NA, int NB, int NACC>
void run() {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 1000;
    hipLaunchKernelGGL((k2<NA, NB, NACC>), dim3(256), dim3(512), 0, 0, out, 10, cyc);
    hipLaunchKernelGGL((k2<NA, NB, NACC>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("A regs %3d, B regs/acc %3d, accumulators %d : MFMA wave %7.1f cycles per 128 MFMA | partner %7.1f cycles per 256 v_fma\n",
           NA, NB, NACC, (double)c[0] / iters, (double)c[4] / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<1, 1, 8>();
    run<1, 1, 4>();
    run<32, 1, 4>();
    run<1, 32, 4>();
    run<32, 32, 4>();
    run<32, 16, 8>();
    return 0;
}
