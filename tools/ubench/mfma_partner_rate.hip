// Micro-benchmark: issue rate available to the SIMD partner of a wave that streams fp32 MFMAs, per
// instruction kind.  waves 0-3 stream MFMAs for a fixed count then raise an LDS flag; waves 4-7 loop over
// a block of N instructions of one kind until they see the flag; rate = instructions retired / MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int PRIO, int F16>
__global__ __launch_bounds__(512) void k2(float *out, float *gbuf, int iters, long long *res) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    __shared__ volatile int flag;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 0.001f;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    float s = 0;
    if (wave < 4) {
        f32x4 acc[4];
        float a = lane * 0.001f, b = lane * 0.002f;
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            if (F16) {
                f16x8 ah, bh;
                for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); }
#pragma unroll
                for (int k = 0; k < 128; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[k & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 128; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k & 3], 0, 0, 0);
            }
        }
        long long t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (lane == 0) { flag = 1; res[blockIdx.x * 16 + wave] = t1 - t0; }
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
        long long count = 0;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        int sidx = __builtin_amdgcn_readfirstlane(blockIdx.x);
        while (!flag) {
            if (KIND == 0) {
#pragma unroll
                for (int m = 0; m < 64; ++m) v[m & 7] = __builtin_fmaf(v[m & 7], 0.999f, 0.001f);
            } else if (KIND == 1) {
#pragma unroll
                for (int m = 0; m < 64; ++m) v[m & 7] = __builtin_amdgcn_exp2f(v[m & 7]);
            } else if (KIND == 2) {
#pragma unroll
                for (int m = 0; m < 64; ++m) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sidx));
            } else if (KIND == 3) {
#pragma unroll
                for (int m = 0; m < 64; ++m) v[m & 7] += lds[(lane + m * 64) & 8191];
            } else if (KIND == 4) {
#pragma unroll
                for (int m = 0; m < 64; ++m) lds[(lane + m * 64) & 4095] = v[m & 7];
            } else if (KIND == 5) {
#pragma unroll
                for (int m = 0; m < 64; ++m) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sidx));
            }
            count += 64;
        }
        for (int i = 0; i < 8; ++i) s += v[i];
        s += sidx;
        if (lane == 0) res[blockIdx.x * 16 + wave] = count;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int PRIO, int F16>
void run(const char *name) {
    float *out, *gbuf; long long *res;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&gbuf, 1 << 20); (void)hipMalloc(&res, 256 * 16 * 8);
    const int iters = 200;
    hipLaunchKernelGGL((k2<KIND, PRIO, F16>), dim3(256), dim3(512), 0, 0, out, gbuf, iters, res);
    (void)hipDeviceSynchronize();
    long long c[16]; (void)hipMemcpy(c, res, 128, hipMemcpyDeviceToHost);
    printf("%-28s MFMA wave %6.1f cycles/MFMA | partner retired %.2f instr per MFMA (%.1f cycles each)\n", name,
           (double)c[0] / (iters * 128.0), (double)c[4] / (iters * 128.0), (double)c[0] / (double)c[4]);
    (void)hipFree(out); (void)hipFree(gbuf); (void)hipFree(res);
}
int main() {
    run<0, 0, 0>("fp32 MFMA | partner v_fma");
    run<0, 0, 1>("f16 MFMA  | partner v_fma");
    run<1, 0, 1>("f16 MFMA  | partner v_exp");
    run<3, 0, 1>("f16 MFMA  | partner ds_read(+add)");
    run<4, 0, 1>("f16 MFMA  | partner ds_write");
    run<0, 3, 1>("f16 MFMA  | partner v_fma prio3");
    return 0;
}
