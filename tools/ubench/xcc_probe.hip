// Which XCD does block b of a launch run on when several launches are in flight on several streams?
// (lstm2_fused.hip.h wanted b % 8; that holds for a launch alone on the chip only.)  One workgroup per CU (100 KB LDS), ~40 us of
// spinning per block; three streams, 12 launches each; per launch: the offset (xcc(0) - 0) mod 8, whether xcc(b) = (b + offset) mod 8
// for all b, and the count of blocks per XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
    __shared__ unsigned char pad[100 * 1024];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[threadIdx.x] = (unsigned char)xcc;
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) | (pad[1] << 8 & 0);
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 192, streams = argc > 2 ? atoi(argv[2]) : 3, launches = 12;
    std::vector<hipStream_t> st(streams);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * grid * streams * launches);
    for (int l = 0; l < launches; ++l)
        for (int s = 0; s < streams; ++s)
            hipLaunchKernelGGL(probe, dim3(grid + 8 * ((l + s) % 3)), dim3(256), 0, st[s], d + (size_t)(l * streams + s) * grid, 4000 + 1000 * s);   // 100 MHz wall clock: 40-60 us
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)grid * streams * launches);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int l = 0; l < launches; ++l)
        for (int s = 0; s < streams; ++s) {
            const unsigned *p = h.data() + (size_t)(l * streams + s) * grid;
            int off = p[0] & 7, rot = 1, cnt[8] = {0};
            for (int b = 0; b < grid; ++b) { rot &= (int)(p[b] & 7) == ((b + off) & 7); cnt[p[b] & 7]++; }
            printf("launch %2d stream %d: offset %d  rotation-consistent %d  blocks per XCD %d %d %d %d %d %d %d %d   first 16:", l, s, off, rot,
                   cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7]);
            for (int b = 0; b < 16; ++b) printf(" %u", p[b] & 7);
            printf("\n");
        }
    return 0;
}
