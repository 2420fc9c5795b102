// Stand-alone timing of the LSTM2 projection GEMM (gemm_split.hip.h) on a batch-1024 problem, by workgroup-group count.
#include "probed/gemm_split.hip.h"   // frozen round-1 copy with the GEMM_PROBE_* modes
#include <cstdio>
#include <vector>
using namespace clair;
int main() {
    const int n_pad = 1024, ntiles = n_pad / 32, m_rows = 33 * n_pad;
    unsigned short *A, *B; float *bias, *C;
    (void)hipMalloc(&A, (size_t)2 * m_rows * 256 * 2); (void)hipMalloc(&B, (size_t)8 * 2 * 2 * 16 * 2 * 64 * 8 * 2);
    (void)hipMalloc(&bias, 4096); (void)hipMalloc(&C, (size_t)m_rows * 1024 * 4);
    std::vector<unsigned short> h((size_t)2 * m_rows * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3000 + (i * 2654435761u >> 22 & 0x3ff);
    (void)hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, h.data(), (size_t)8 * 2 * 2 * 16 * 2 * 64 * 8 * 2, hipMemcpyHostToDevice);
    (void)hipMemset(bias, 0, 4096);
    for (int groups : {1, 2, 3, 4, 8}) {
        GemmSplitArgs a{A, B, bias, C, n_pad, ntiles, m_rows, groups};
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_split_kernel, dim3(64 * groups), dim3(256), 0, 0, a);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(gemm_split_kernel, dim3(64 * groups), dim3(256), 0, 0, a);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("gemm_split batch 1024, %d groups per XCD (%d workgroups): %.1f us\n", groups, 64 * groups, ms * 1000.f / 30);
    }
    return 0;
}
