// What does a per-step exchange of the recurrent state between the CUs of one XCD cost?  (LSTM2 computing its own input
// projection with the gate rows of a (tile, direction) split over P CUs needs every CU's slice of h_t at every other CU of the group before
// the h-part of step t + 1 -- 33 times per forward pass.  This measures the hand-off itself, and how much of it hides behind MFMA work that does
// not depend on h: the x-part of the next step.)
//
//   xcu_exchange <P = CUs per group: 2 | 4> <groups per XCD> <steps> <mfma per step> <release: 0 = wait for the stores' acks, 1 = + buffer_wbl2>
//
// A group is P workgroups (256 threads, one per CU: 100 KB of LDS keeps them apart) that find themselves on the same XCD (HW_REG_XCC_ID; a launch
// alone on the chip places block b on XCD b % 8 -- checked, mismatches are counted).  Per step every member
//   1. stores its slice (32 candidates x 128 / P hidden units x two fp16 planes = 16 KB / P) into the group's buffer of that step parity,
//   2. waits for the stores' acknowledgements (vmcnt(0); variant 1: then buffer_wbl2 sc1, the release the HIP memory model names),
//   3. publishes "step s done" in its flag word (relaxed, agent scope: written through to the L2),
//   4. issues <mfma per step> independent v_mfma_f32_32x32x16_f16 (what the x-part of the next step would be doing meanwhile),
//   5. polls the other members' flags, invalidates its L1 (acquire), loads their slices and checks every byte against what they must have written.
// "tagged" protocol: no flag and no wait for acknowledgements -- every 16-byte word carries the step it belongs to (12 bytes of payload + a tag) and
// the reader polls the words it needs, past its L1, until the tag is this step's: ~1.5 L2 round trips on the critical path instead of ~3.5.
// Output: ns per step (wall clock), for the exchange alone, the MFMAs alone and both -- the exposed part of the exchange is (both - MFMAs alone).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/xcu_exchange.hip -o tools/ubench/xcu_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

struct Args {
    u32x4 *buf;          // [groups][2 parities][P members][slice of 16-byte words]
    unsigned *flags;     // [groups][P], zeroed
    unsigned *bad;       // [0] data mismatches, [1] workgroups not on the XCD their id implies, [2] waits that ran out
    long long *ticks;    // [blocks] wall-clock ticks (100 MHz) of the timed loop
    int P, groups_per_xcd, steps, mfma, release, exchange;
    unsigned salt;       // differs from launch to launch: a slice left over from an earlier launch is not what this one expects
};

__device__ __forceinline__ unsigned pattern(unsigned salt, int group, int member, int step, int word) {
    return salt ^ (unsigned)(group * 2654435761u) ^ (unsigned)(member * 40503u) ^ (unsigned)(step * 2246822519u) ^ (unsigned)word;
}

__global__ __launch_bounds__(256) void exchange_kernel(Args a) {
    __shared__ unsigned char pad[100 * 1024];           // one workgroup per CU
    const int tid = threadIdx.x;
    pad[tid] = (unsigned char)tid;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int group_on_xcd = local / a.P, member = local % a.P;
    const int group = group_on_xcd * 8 + xcd;
    if (tid == 0 && (int)(xcc & 7u) != xcd) atomicAdd(a.bad + 1, 1u);
    const int slice_words = (16 * 1024 / a.P) / 16;      // 16-byte words per member and step
    const int per_thread = slice_words / 256;            // 4 KB / P ... : 1 (P = 4) or 2 (P = 2) words per thread
    u32x4 *gbuf = a.buf + (size_t)group * 2 * a.P * slice_words;
    unsigned *gflags = a.flags + (size_t)group * a.P;

    f32x16 acc0, acc1, acc2, acc3;
    for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = acc2[j] = acc3[j] = 0.0f;
    f16x8 wa, wb;
    for (int j = 0; j < 8; ++j) { wa[j] = (_Float16)(0.001f * (tid & 7)); wb[j] = (_Float16)(0.002f * (j + 1)); }

    __syncthreads();
    const long long t0 = wall_clock64();
    for (int s = 0; s < a.steps; ++s) {
        if (a.exchange == 2) {       // TAGGED: every 16-byte word carries the step it belongs to -- no flag, no wait for acknowledgements
            u32x4 *mine = gbuf + ((size_t)(s & 1) * a.P + member) * slice_words;
            for (int k = 0; k < per_thread; ++k) {
                const int wd = k * 256 + tid;
                const unsigned v = pattern(a.salt, group, member, s, wd);
                __builtin_nontemporal_store((u32x4){v, v + 1, v + 2, (unsigned)(s + 1) ^ a.salt}, mine + wd);
            }
        }
        if (a.exchange == 1) {
            u32x4 *mine = gbuf + ((size_t)(s & 1) * a.P + member) * slice_words;
            for (int k = 0; k < per_thread; ++k) {
                const int wd = k * 256 + tid;
                const unsigned v = pattern(a.salt, group, member, s, wd);
                mine[wd] = (u32x4){v, v + 1, v + 2, v + 3};
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every store of this wave acknowledged by the L2 (gfx9: stores count in vmcnt)
            if (a.release) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                                                      // ... and of the other three waves
            if (tid == 0) __hip_atomic_store(gflags + member, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int m = 0; m < a.mfma; m += 4) {            // four independent accumulator chains, registers fixed at compile time
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(wa), "v"(wb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(wa), "v"(wb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(wa), "v"(wb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc3) : "v"(wa), "v"(wb));
        }
        if (a.exchange == 2) {       // the reader polls the words it needs until they carry this step's tag (16-byte stores land whole)
            unsigned wrong = 0;
            for (int o = 1; o < a.P; ++o) {
                const int other = (member + o) % a.P;
                const u32x4 *theirs = gbuf + ((size_t)(s & 1) * a.P + other) * slice_words;
                for (int k = 0; k < per_thread; ++k) {
                    const int wd = k * 256 + tid;
                    const unsigned want = pattern(a.salt, group, other, s, wd), tag = (unsigned)(s + 1) ^ a.salt;
                    u32x4 v;
                    const long long deadline = wall_clock64() + 20000000;
                    for (;;) {
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + wd) : "memory");     // past this CU's L1
                        if (v[3] == tag) break;
                        if (wall_clock64() > deadline) { atomicAdd(a.bad + 2, 1u); break; }
                    }
                    wrong += (v[0] != want) + (v[1] != want + 1) + (v[2] != want + 2);
                }
            }
            if (wrong) atomicAdd(a.bad, wrong);
            __syncthreads();         // the real kernel's h buffer barrier
        }
        if (a.exchange == 1) {
            if (tid < a.P && tid != member) {
                const long long deadline = wall_clock64() + 20000000;                             // 0.2 s
                while (__hip_atomic_load(gflags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(s + 1)) {
                    if (wall_clock64() > deadline) { atomicAdd(a.bad + 2, 1u); break; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                    // nothing of the peers' slices may come from this CU's L1
            unsigned wrong = 0;
            for (int o = 1; o < a.P; ++o) {
                const int other = (member + o) % a.P;
                const u32x4 *theirs = gbuf + ((size_t)(s & 1) * a.P + other) * slice_words;
                for (int k = 0; k < per_thread; ++k) {
                    const int wd = k * 256 + tid;
                    const u32x4 v = __builtin_nontemporal_load(theirs + wd);
                    const unsigned want = pattern(a.salt, group, other, s, wd);
                    wrong += (v[0] != want) + (v[1] != want + 1) + (v[2] != want + 2) + (v[3] != want + 3);
                }
            }
            if (wrong) atomicAdd(a.bad, wrong);
            // a slice of parity s & 1 is overwritten at step s + 2: by then every member has published s + 1, i.e. has read step s
        }
    }
    const long long t1 = wall_clock64();
    const float sink = acc0[0] + acc1[0] + acc2[0] + acc3[0];
    if (tid == 0) a.ticks[blockIdx.x] = (t1 - t0) + (sink == 12345.678f ? 1 : 0) + (pad[17] == 99 ? 1 : 0);
}

static double run(Args a, int blocks, const char *what) {
    static unsigned launches = 0;
    a.salt = 0x9e3779b9u * ++launches;
    hipMemset(a.flags, 0, sizeof(unsigned) * 8 * a.groups_per_xcd * a.P);
    hipMemset(a.bad, 0, 3 * sizeof(unsigned));
    hipLaunchKernelGGL(exchange_kernel, dim3(blocks), dim3(256), 0, 0, a);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); exit(1); }
    std::vector<long long> t(blocks);
    unsigned bad[3];
    hipMemcpy(t.data(), a.ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    hipMemcpy(bad, a.bad, sizeof bad, hipMemcpyDeviceToHost);
    long long mx = 0;
    double mean = 0;
    for (long long v : t) { mx = v > mx ? v : mx; mean += (double)v; }
    mean /= blocks;
    const double ns = mean * 10.0 / a.steps, ns_max = mx * 10.0 / a.steps;      // 100 MHz wall clock
    printf("%-34s %8.0f ns per step (slowest workgroup %8.0f)   wrong bytes %u  misplaced workgroups %u  waits run out %u\n", what, ns, ns_max, bad[0], bad[1], bad[2]);
    return ns;
}

int main(int argc, char **argv) {
    Args a{};
    a.P = argc > 1 ? atoi(argv[1]) : 4;
    a.groups_per_xcd = argc > 2 ? atoi(argv[2]) : 8;
    a.steps = argc > 3 ? atoi(argv[3]) : 3300;
    a.mfma = argc > 4 ? atoi(argv[4]) : 48;
    a.release = argc > 5 ? atoi(argv[5]) : 0;
    if (a.P != 2 && a.P != 4) { fprintf(stderr, "P must be 2 or 4\n"); return 1; }
    const int groups = 8 * a.groups_per_xcd, blocks = groups * a.P;
    hipMalloc((void **)&a.buf, (size_t)groups * 2 * 16 * 1024);
    hipMalloc((void **)&a.flags, sizeof(unsigned) * groups * a.P);
    hipMalloc((void **)&a.bad, 3 * sizeof(unsigned));
    hipMalloc((void **)&a.ticks, blocks * sizeof(long long));
    hipMemset(a.buf, 0, (size_t)groups * 2 * 16 * 1024);
    printf("# %d CUs per group, %d groups per XCD (%d workgroups), %d steps, %d MFMAs per step, release %s; slice %d KB per member and step\n", a.P, a.groups_per_xcd, blocks, a.steps,
           a.mfma, a.release ? "buffer_wbl2 sc1" : "vmcnt(0) only", 16 / a.P);
    Args w = a;                   // warm-up
    w.steps = 100; w.exchange = 1;
    run(w, blocks, "(warm-up)");
    a.exchange = 0;
    const double t_mfma = run(a, blocks, "MFMAs alone");
    const int mf = a.mfma;
    for (int mode = 1; mode <= 2; ++mode) {
        const char *name = mode == 1 ? "flag" : "tagged";
        char what[64];
        a.exchange = mode;
        a.mfma = 0;
        snprintf(what, sizeof what, "exchange alone (%s)", name);
        const double t_x = run(a, blocks, what);
        a.mfma = mf;
        snprintf(what, sizeof what, "exchange (%s) + MFMAs", name);
        const double t_both = run(a, blocks, what);
        printf("=> %s: exchange alone %.0f ns; beside %d MFMAs (%.0f ns) a step takes %.0f ns: %.0f ns of the exchange exposed\n", name, t_x, mf, t_mfma, t_both, t_both - t_mfma);
    }
    return 0;
}
