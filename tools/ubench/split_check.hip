// Check the 2-way fp16 split of y = s*v as hipcc may emit it: (A) v_mul + v_cvt_f16_f32 + v_cvt_f32_f16 + v_sub + v_cvt_f16_f32,
// (B) v_fma_mixlo_f16 (hi) + v_fma_mix_f32 (residual) + v_cvt_f16_f32.  Reports max |hi + lo - y| / |y| per magnitude decade.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float *v, float s, float *outA, float *outB, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = v[i];
    {
        float y, back, d; unsigned hi, lo;
        asm volatile("v_mul_f32 %0, %5, %6\n\tv_cvt_f16_f32 %1, %0\n\ts_nop 1\n\tv_cvt_f32_f16 %2, %1\n\ts_nop 1\n\tv_sub_f32 %3, %0, %2\n\ts_nop 1\n\tv_cvt_f16_f32 %4, %3"
                     : "=&v"(y), "=&v"(hi), "=&v"(back), "=&v"(d), "=&v"(lo) : "v"(x), "v"(s));
        _Float16 h, l; unsigned short hs = hi & 0xffff, ls = lo & 0xffff;
        __builtin_memcpy(&h, &hs, 2); __builtin_memcpy(&l, &ls, 2);
        outA[i] = (float)((double)(float)h + (double)(float)l);
    }
    {
        float d; unsigned hi = 0, lo;
        asm volatile("v_fma_mixlo_f16 %0, %3, %4, 0\n\tv_fma_mix_f32 %1, %3, %4, -%0 op_sel_hi:[0,0,1]\n\ts_nop 1\n\tv_cvt_f16_f32 %2, %1"
                     : "+v"(hi), "=&v"(d), "=&v"(lo) : "v"(x), "v"(s));
        _Float16 h, l; unsigned short hs = hi & 0xffff, ls = lo & 0xffff;
        __builtin_memcpy(&h, &hs, 2); __builtin_memcpy(&l, &ls, 2);
        outB[i] = (float)((double)(float)h + (double)(float)l);
    }
}
int main() {
    const int n = 1 << 16;
    std::vector<float> hv(n);
    unsigned r = 12345;
    for (int i = 0; i < n; ++i) { r = r * 1664525u + 1013904223u; float u = (r >> 8) / 16777216.0f; int e = (i % 8) - 6; hv[i] = (0.5f + u) * ldexpf(1.0f, e * 2) * ((i & 1) ? -1.f : 1.f); }
    float *v, *a, *b; (void)hipMalloc(&v, n * 4); (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
    (void)hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice);
    const float s = 1.0507009873554804934193349852946f;
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, v, s, a, b, n);
    std::vector<float> ha(n), hb(n);
    (void)hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    for (int e = 0; e < 8; ++e) {
        double ma = 0, mb = 0;
        for (int i = e; i < n; i += 8) { double y = (double)hv[i] * (double)s; ma = fmax(ma, fabs(ha[i] - y) / fabs(y)); mb = fmax(mb, fabs(hb[i] - y) / fabs(y)); }
        printf("|v| ~ 2^%3d: rel err  cvt/sub/cvt %.2e   fma_mixlo/fma_mix/cvt %.2e\n", (e - 6) * 2, ma, mb);
    }
    return 0;
}
