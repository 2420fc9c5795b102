// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask enable?  One workgroup per CU (64 KiB of LDS each), each records
// XCC_ID and the HW_ID fields; the host prints, per mask, the set of (xcc, se, cu) the workgroups ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned *out) {
    __shared__ char pad[65536 - 256];
    pad[threadIdx.x] = 1;
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    for (volatile int i = 0; i < 20000; ++i) { }     // stay resident so that the next workgroup goes elsewhere
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw + pad[0] - 1; }
}
int main() {
    unsigned *d; (void)hipMalloc(&d, 8 * 1024);
    std::vector<unsigned> h(2048);
    auto run = [&](const char *name, std::vector<uint32_t> mask, int wgs) {
        hipStream_t s;
        if (mask.empty()) (void)hipStreamCreate(&s);
        else if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
        (void)hipMemsetAsync(d, 0xff, 8192, s);
        hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, s, d);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per;   // xcc -> set of (se, cu)
        for (int i = 0; i < wgs; ++i) per[h[2 * i] & 0xf].insert((h[2 * i + 1] >> 8) & 0xff);   // HW_ID bits 8..11 cu, 12..15 sh/se (layout varies)
        printf("%-28s %3d workgroups:", name, wgs);
        for (auto &kv : per) printf(" xcc%u:%zu", kv.first, kv.second.size());
        printf("   first 16 xcc ids:");
        for (int i = 0; i < 16 && i < wgs; ++i) printf(" %u", h[2 * i] & 0xf);
        printf("\n");
        if (wgs <= 64) {
            std::set<unsigned> cus;
            for (int i = 0; i < wgs; ++i) cus.insert(((h[2 * i] & 0xf) << 8) | ((h[2 * i + 1] >> 8) & 0xff));
            printf("      (xcc, se/sh/cu) set:");
            for (unsigned c : cus) printf(" %u:%02x", c >> 8, c & 0xff);
            printf("\n");
        }
        (void)hipStreamDestroy(s);
    };
    run("no mask", {}, 256);
    run("bits 0-63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}, 64);
    run("bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}, 32);
    run("bits 32-63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0}, 32);
    run("bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}, 8);
    run("every 8th bit (0,8,..)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}, 32);
    run("bits 64-127", {0, 0, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}, 64);
    run("bits 0-63, 256 workgroups", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}, 256);
    return 0;
}
