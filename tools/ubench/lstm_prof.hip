// Issue-timeline probe of lstm_rec2_kernel: s_memtime stamps of workgroup 0 / wave 0 at step 8.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLSTM_PROFILE -I clair_amd/csrc tools/ubench/lstm_prof.hip -o tools/ubench/lstm_prof
#include "lstm.hip.h"
#include <cstdio>
#include <vector>
using namespace clair;
int main(int argc, char **argv) {
    const int n_pad = argc > 1 ? atoi(argv[1]) : 1024, ntiles = n_pad / 16;
    size_t zx_n = (size_t)2 * 33 * ntiles * 4 * 8 * 256, wh_n = (size_t)2 * 4 * 8 * 8 * 64 * 4, ao_n = (size_t)33 * n_pad * 256;
    float *zx, *wh, *ao; long long *prof;
    (void)hipMalloc(&zx, zx_n * 4); (void)hipMalloc(&wh, wh_n * 4); (void)hipMalloc(&ao, ao_n * 4); (void)hipMalloc(&prof, 64 * 8);
    std::vector<float> h(zx_n); for (size_t i = 0; i < zx_n; ++i) h[i] = ((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    (void)hipMemcpy(zx, h.data(), zx_n * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(wh_n); for (size_t i = 0; i < wh_n; ++i) hw[i] = (((i * 40503u) % 2001) * 1e-3f - 1.0f) * 0.1f;
    (void)hipMemcpy(wh, hw.data(), wh_n * 4, hipMemcpyHostToDevice);
    (void)hipMemset(prof, 0, 64 * 8);
    LstmArgs a{zx, wh, ao, n_pad, ntiles, prof};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(lstm_rec2_kernel, dim3(ntiles), dim3(256), 0, 0, a);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.1f us (%d workgroups)\n", rep, ms * 1e3, ntiles);
    }
    long long t[64]; (void)hipMemcpy(t, prof, 64 * 8, hipMemcpyDeviceToHost);
    printf("phase A: entry->mfma-start %lld | chunks:", t[1] - t[0]);
    for (int k = 0; k < 8; ++k) printf(" %lld", t[2 + k] - t[1 + k]);
    printf(" | last chunk->phase B entry (barrier) %lld\n", t[10] - t[9]);
    printf("phase B: entry->mfma-start %lld | chunks:", t[11] - t[10]);
    for (int k = 0; k < 8; ++k) printf(" %lld", t[12 + k] - t[11 + k]);
    printf(" | last chunk->after barrier %lld\n", t[20] - t[19]);
    printf("step total %lld ticks (pure MFMA floor 2 x 8192)\n", t[20] - t[0]);
    return 0;
}
