// Phase timing of the recurrent kernels (lstm32.hip.h) from inside: s_memtime stamps of workgroup 0 / wave 0
// at step start, after each of the four blocks, after the exposed gate tail and after the barrier.
#define L32_PROBE 1
#include "probed/lstm32.hip.h"   // frozen round-1 copy of the recurrent kernels WITH the probe build modes (production headers carry none)
#include <cstdio>
#include <vector>
using namespace clair;
template <bool FIRST> void run(int n_pad) {
    const int ntiles = n_pad / 32;
    unsigned short *whs, *wxs, *a1; float *zx, *bq, *a2, *x;
    (void)hipMalloc(&whs, (size_t)2 * 4 * 4 * 8 * 2 * 64 * 16); (void)hipMalloc(&wxs, (size_t)2 * 4 * 4 * 2 * 2 * 64 * 16);
    (void)hipMalloc(&a1, (size_t)2 * 33 * n_pad * 256 * 2); (void)hipMalloc(&a2, (size_t)33 * n_pad * 256 * 4);
    (void)hipMalloc(&zx, (size_t)33 * n_pad * 1024 * 4); (void)hipMalloc(&bq, 4096); (void)hipMalloc(&x, (size_t)n_pad * 33 * 32 * 4);
    std::vector<unsigned short> h((size_t)2 * 4 * 4 * 8 * 2 * 64 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x2000 + (i * 2654435761u >> 22 & 0x3ff) + ((i & 1) << 15);
    (void)hipMemcpy(whs, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(wxs, h.data(), (size_t)2 * 4 * 4 * 2 * 2 * 64 * 16, hipMemcpyHostToDevice);
    (void)hipMemset(zx, 0, (size_t)33 * n_pad * 1024 * 4); (void)hipMemset(bq, 0, 4096); (void)hipMemset(x, 0, (size_t)n_pad * 33 * 32 * 4);
    long long *st; (void)hipMalloc(&st, 33 * 16 * 8); (void)hipMemset(st, 0, 33 * 16 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(l32_stamps), &st, sizeof(st));
    Lstm32Args a{x, wxs, bq, zx, whs, a1, a2, n_pad, ntiles, -1};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lstm32_kernel<FIRST>), dim3(ntiles * 2), dim3(256), 0, 0, a);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((lstm32_kernel<FIRST>), dim3(ntiles * 2), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hs(33 * 16);
    (void)hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    printf("lstm32<%s> n_pad %d: %.1f us per launch", FIRST ? "first" : "second", n_pad, ms * 1000 / 20);
#ifdef L32_PROBE_NOSTAMP
    printf("   step loop %lld cycles = %.0f per step", hs[1] - hs[0], (double)(hs[1] - hs[0]) / 33);
#endif
    printf("\n");
#ifdef L32_PROBE_NOSTAMP
    return;
#endif
    for (int s : {1, 5, 16, 17, 31}) {
        const long long *t = &hs[s * 16];
        printf("  step %2d: head %5lld | block0 %5lld | block1 %5lld | block2 %5lld | block3 %5lld | gate tail %5lld | barrier %5lld | total %5lld (next step starts +%lld)\n", s,
               t[7] - t[0], t[1] - t[7], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[6] - t[0], hs[(s + 1) * 16] - t[6]);
        printf("           block0 MFMA 0,3,6,..21 at +: ");
        for (int i = 8; i < 16; ++i) printf("%lld ", t[i] - t[7]);
        printf("\n");
    }
}
int main() {
    run<false>(1024);
    run<true>(1024);
    return 0;
}
