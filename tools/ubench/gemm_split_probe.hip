// Where does the LSTM2 input projection (gemm_split_kernel) spend its time?  Runs the product kernel and
// three amputated variants on a batch-1024 problem: no zx store / no MFMAs / no global loads after the first.
#include "gemm_split.hip.h"
#include <cstdio>
#include <vector>
using namespace clair;
template <int PROBE> float run(GemmSplitArgs a, int grid, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_split_kernel<PROBE>, dim3(grid), dim3(256), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split_kernel<PROBE>, dim3(grid), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / reps;
}
int main() {
    const int n_pad = 1024, ntiles = n_pad / 16, m_rows = 33 * n_pad;
    unsigned short *A, *B; float *bias, *C;
    hipMalloc(&A, (size_t)2 * m_rows * 256 * 2); hipMalloc(&B, (size_t)8 * 2 * 1024 * 32 * 2);
    hipMalloc(&bias, 4096); hipMalloc(&C, (size_t)m_rows * 1024 * 4);
    std::vector<unsigned short> h((size_t)2 * m_rows * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3000 + (i * 2654435761u >> 22 & 0x3ff);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)8 * 2 * 1024 * 32 * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 4096);
    GemmSplitArgs a{A, B, bias, C, n_pad, ntiles, m_rows};
    const int grid = (((m_rows + 127) / 128 + 7) / 8) * 64;
    printf("gemm_split batch 1024: full %.1f us | no store %.1f | no mfma %.1f | no loads %.1f\n",
           run<0>(a, grid, 50), run<1>(a, grid, 50), run<2>(a, grid, 50), run<3>(a, grid, 50));
    // phase timestamps of every workgroup (s_memtime ticks), one launch
    long long *st; hipMalloc(&st, (size_t)grid * 6 * 8); hipMemset(st, 0, (size_t)grid * 6 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(gemm_probe_stamps), &st, sizeof(st));
    hipLaunchKernelGGL(gemm_split_kernel<4>, dim3(grid), dim3(256), 0, 0, a);
    hipDeviceSynchronize();
    std::vector<long long> hs((size_t)grid * 6);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    FILE *f = fopen("gpurun_out/gemm_stamps.txt", "w");
    if (f) {
        for (int w = 0; w < grid; ++w)
            if (hs[w * 6]) fprintf(f, "%d %lld %lld %lld %lld %llx %llx\n", w, hs[w * 6], hs[w * 6 + 1], hs[w * 6 + 2], hs[w * 6 + 3], hs[w * 6 + 4], hs[w * 6 + 5]);
        fclose(f);
    }
    return 0;
}
