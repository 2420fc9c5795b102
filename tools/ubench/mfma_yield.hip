// Micro-benchmark: what makes a wave that streams fp32 MFMAs give issue slots to its SIMD partner?
// The MFMA wave inserts a "yield" instruction every EVERY MFMAs; the partner loops over 64 v_fma and
// polls an LDS flag.  Reported: MFMA-wave cycles per MFMA and partner instructions retired per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int YIELD, int EVERY>
__global__ __launch_bounds__(512) void k2(float *out, int iters, long long *res) {
    __shared__ volatile int flag;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    float s = 0;
    if (wave < 4) {
        f32x4 acc[4];
        float a = lane * 0.001f, b = lane * 0.002f;
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 128; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k & 3], 0, 0, 0);
                if ((k % EVERY) == EVERY - 1) {
                    if (YIELD == 1) asm volatile("s_nop 0");
                    if (YIELD == 2) asm volatile("s_sleep 1");
                    if (YIELD == 3) { asm volatile("s_setprio 0"); }
                    if (YIELD == 4) asm volatile("s_branch 1f\n1:");
                    if (YIELD == 5) asm volatile("s_nop 7\n\ts_nop 7");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        long long t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (lane == 0) { flag = 1; res[blockIdx.x * 16 + wave] = t1 - t0; }
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
        long long count = 0;
        while (!flag) {
#pragma unroll
            for (int m = 0; m < 64; ++m) v[m & 7] = __builtin_fmaf(v[m & 7], 0.999f, 0.001f);
            count += 64;
        }
        for (int i = 0; i < 8; ++i) s += v[i];
        if (lane == 0) res[blockIdx.x * 16 + wave] = count;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int YIELD, int EVERY>
void run(const char *name) {
    float *out; long long *res;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&res, 256 * 16 * 8);
    const int iters = 200;
    hipLaunchKernelGGL((k2<YIELD, EVERY>), dim3(256), dim3(512), 0, 0, out, iters, res);
    (void)hipDeviceSynchronize();
    long long c[16]; (void)hipMemcpy(c, res, 128, hipMemcpyDeviceToHost);
    printf("%-34s every %3d MFMA: MFMA wave %6.2f cycles/MFMA | partner %.2f v_fma per MFMA\n", name, EVERY,
           (double)c[0] / (iters * 128.0), (double)c[4] / (iters * 128.0));
    (void)hipFree(out); (void)hipFree(res);
}
int main() {
    run<0, 128>("no yield");
    run<1, 8>("s_nop 0");
    run<2, 8>("s_sleep 1");
    run<2, 32>("s_sleep 1");
    run<3, 8>("s_setprio 0");
    run<4, 8>("s_branch next");
    run<4, 32>("s_branch next");
    run<5, 8>("2 x s_nop 7");
    return 0;
}
