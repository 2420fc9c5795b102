// Recurrent half of one BiLSTM layer (both directions), fp32-exact on MFMA.
//
// Reference semantics: CudnnCompatibleLSTMCell(128) under stack_bidirectional_dynamic_rnn
// (clair/model.py:299-312, 423-451): per step z = [x_t, h_{t-1}].W + b, gates (i, c~, f, o),
// c_t = sig(f) c_{t-1} + sig(i) tanh(c~), h_t = sig(o) tanh(c_t); the backward direction walks
// t = 32..0; zero initial state.  The x-part (x_t.Wx + b) arrives precomputed from the
// projection GEMM in fragment-major order (gemm.hip.h: zx_block_offset), so each step only
// adds h_{t-1}.Wh, a [16,128]x[128,512] product per 16-candidate tile.
//
// Mapping to the CU: one 256-thread workgroup (one wave per SIMD) owns TILES tiles of 16
// candidates of one direction for all 33 steps.  Wave w owns hidden units 32w..32w+31 of all
// four gates (8 column blocks of 16), so the gate non-linearities are lane-local in the MFMA
// C layout.  Its [128 x 128] slice of Wh stays in 256 VGPRs for the whole kernel (the register
// file is the only on-chip store big enough for the 256 KiB fp32 Wh); c_t stays in registers;
// h_t is exchanged between the four waves through a double-buffered LDS tile, one barrier per
// step.  K is visited in the order k = q*32 + kk (q = lane>>4) so that a lane's A operands for
// 4 consecutive MFMAs are one ds_read_b128.
#pragma once
#include "common.hip.h"

namespace clair {

constexpr int H_LDS_ROW = HID + 4;  // 132 floats: rows 16 B apart in bank space -> conflict-free b128 reads

struct LstmArgs {
    const float *zx;   // fragment-major x-projection [2][33][ntiles][4][8][64][4]
    const float *whp;  // packed recurrent weights [2][4][8][8][64][4]  (dir, wave, nb, kk/4, lane, kk%4)
    float *aout;       // [33][n_pad][256]  (fw -> cols 0..127, bw -> 128..255)
    int n_pad;
    int ntiles;
    long long *prof;   // optional issue-timeline probe (tools/ubench/lstm_prof.hip); nullptr in production
};

template <int TILES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_rec_kernel(LstmArgs p) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][TILES][16][H_LDS_ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int d = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * TILES;

    // resident weights: Bw[nb][kk] = Wh[k = lq*32 + kk][col = g*128 + 32w + 16hh + li], nb = g*2+hh
    float Bw[8][32];
    {
        const f32x4 *wp = (const f32x4 *)p.whp + (size_t)(d * 4 + w) * (8 * 8 * 64) + lane;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                f32x4 v = wp[(nb * 8 + k4) * 64];
                Bw[nb][k4 * 4 + 0] = v[0];
                Bw[nb][k4 * 4 + 1] = v[1];
                Bw[nb][k4 * 4 + 2] = v[2];
                Bw[nb][k4 * 4 + 3] = v[3];
            }
    }

    float cst[TILES][8];
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int e = 0; e < 8; ++e) cst[tl][e] = 0.0f;

    // zx fragments of this wave: block (t, tile) is 4 waves x 8 nb x 256 floats
    auto zx_ptr = [&](int t, int tile) {
        return (const f32x4 *)(p.zx + ((((size_t)(d * T_POS + t) * p.ntiles + tile) * 4 + w) * 8) * 256) + lane;
    };

    f32x4 znext[TILES][8];
    {
        const int t = d ? T_POS - 1 : 0;
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            const f32x4 *z = zx_ptr(t, tile0 + tl);
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) znext[tl][nb] = z[nb * 64];
        }
    }

    for (int s = 0; s < T_POS; ++s) {
        const int t = d ? T_POS - 1 - s : s;
        f32x4 acc[TILES][8];
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[tl][nb] = znext[tl][nb];
        if (s + 1 < T_POS) {  // prefetch next step's x-projection under this step's MFMAs
            const int tn = d ? t - 1 : t + 1;
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) {
                const f32x4 *z = zx_ptr(tn, tile0 + tl);
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) znext[tl][nb] = z[nb * 64];
            }
        }
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            if (s > 0) {
                const float *hrow = &hbuf[(s - 1) & 1][tl][li][lq * 32];
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    const f32x4 a = *(const f32x4 *)(hrow + k4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nb = 0; nb < 8; ++nb) acc[tl][nb] = mfma16(a[j], Bw[nb][k4 * 4 + j], acc[tl][nb]);
                }
            }
            // gates: element (row = 4*lq + r, unit = 32w + 16hh + li)
            float *orow = p.aout + ((size_t)t * p.n_pad + (size_t)(tile0 + tl) * 16 + lq * 4) * (2 * HID) + d * HID + w * 32 + li;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ig = sigmoid_f(acc[tl][0 + hh][r]);
                    const float gg = tanh_f(acc[tl][2 + hh][r]);
                    const float fg = sigmoid_f(acc[tl][4 + hh][r]);
                    const float og = sigmoid_f(acc[tl][6 + hh][r]);
                    const float c = fg * cst[tl][hh * 4 + r] + ig * gg;
                    cst[tl][hh * 4 + r] = c;
                    const float h = og * tanh_f(c);
                    hbuf[s & 1][tl][lq * 4 + r][w * 32 + hh * 16 + li] = h;
                    orow[(size_t)r * (2 * HID) + hh * 16] = h;
                }
        }
        __syncthreads();
    }
}

// ---- two-tile variant: MFMA of one tile overlaps the gate math of the other -------------------
// A workgroup owns tiles (2p, 2p+1).  Each step is two phases separated by one barrier:
//   phase A(s): MFMAs of tile 0 for step s   ||  gates of tile 1 for step s-1  (VALU/transcendental)
//   phase B(s): MFMAs of tile 1 for step s   ||  gates of tile 0 for step s
// The two instruction streams inside a phase are independent, so the matrix pipe stays busy while the
// VALU evaluates the non-linearities.  Per tile, single LDS buffers suffice (written in one phase,
// read in the next, rewritten two barriers later):
//   hbuf[tile]  h_t, 16 x 128 (+pad): written by the gate stream, read as MFMA A fragments and, at
//               the start of the next phase, copied out to aout as whole 512-byte rows;
//   zlds[tile]  the x-projection fragments of the step whose MFMAs are running, fetched by LDS-DMA
//               (global_load_lds, 1 KiB per wave-instruction, no VGPRs) at the start of the phase and
//               consumed by the gate stream one phase later -- a full phase (~3.5 us) of latency cover.
// All VMEM of a phase (2 row stores + 8 DMA pieces per wave) is issued right after the barrier, so the
// vmcnt(0) hipcc places in front of the next barrier never waits on a fresh operation.
#ifdef LSTM_PROFILE
#define LSTM_PROF(slot) do { if (p.prof && blockIdx.x == 0 && tid == 0 && prof_on) p.prof[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define LSTM_PROF(slot) do { } while (0)
#endif

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base, lds_base + 1 KiB).
// Issued from inline asm on purpose: hipcc then neither counts it nor fences later ds_reads of the
// same __shared__ array behind it with vmcnt(0) (the data is only read after phase_barrier()).
// M0 carries the LDS base and is compiler-reserved, so it is saved/restored inside the statement.
__device__ __forceinline__ void glds16(const f32x4 *gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// End of a phase: hipcc does not track LDS-DMA completion, so drain this wave's VMEM queue (the DMA
// pieces and row stores issued at the START of the phase, long since landed) before the barrier that
// publishes LDS to the other waves.
__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_rec2_kernel(LstmArgs p) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][16][H_LDS_ROW];
    __shared__ __attribute__((aligned(16))) float zlds[2][4][8][256];   // [tile][wave][nb][lane*4 + r]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int d = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * 2;

    float Bw[8][32];
    {
        const f32x4 *wp = (const f32x4 *)p.whp + (size_t)(d * 4 + w) * (8 * 8 * 64) + lane;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                f32x4 v = wp[(nb * 8 + k4) * 64];
                Bw[nb][k4 * 4 + 0] = v[0];
                Bw[nb][k4 * 4 + 1] = v[1];
                Bw[nb][k4 * 4 + 2] = v[2];
                Bw[nb][k4 * 4 + 3] = v[3];
            }
    }
    float cst[2][8];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int e = 0; e < 8; ++e) cst[tl][e] = 0.0f;

    // LDS-DMA of this wave's 8 x-projection fragments of (step s, tile tl) into zlds[tl][w]
    auto fetch_zx = [&](int s, int tl) {
        const int t = d ? T_POS - 1 - s : s;
        const f32x4 *src = (const f32x4 *)(p.zx + ((((size_t)(d * T_POS + t) * p.ntiles + tile0 + tl) * 4 + w) * 8) * 256) + lane;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)&zlds[tl][w][0][0]);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) glds16(src + nb * 64, lds0 + nb * 1024);
    };
    // one gate element e = hh*4 + r of tile tl: (row 4*lq + r, unit 32w + 16hh + li);
    // pre-activation = recurrent product (acc) + x-projection fragment (zq, bias included)
    auto gate_elem = [&](const f32x4 (&acc)[8], const f32x4 (&zq)[4], int e, int tl) {
        const int hh = e >> 2, r = e & 3;
        const float ig = sigmoid_f(acc[0 + hh][r] + zq[0][r]);
        const float gg = tanh_f(acc[2 + hh][r] + zq[1][r]);
        const float fg = sigmoid_f(acc[4 + hh][r] + zq[2][r]);
        const float og = sigmoid_f(acc[6 + hh][r] + zq[3][r]);
        const float c = fg * cst[tl][e] + ig * gg;
        cst[tl][e] = c;
        const float h = og * tanh_f(c);
        hbuf[tl][lq * 4 + r][w * 32 + hh * 16 + li] = h;
    };
    auto load_zq = [&](f32x4 (&zq)[4], int hh, int tl) {   // the four gates' fragments of half hh
#pragma unroll
        for (int g = 0; g < 4; ++g) zq[g] = *(const f32x4 *)&zlds[tl][w][g * 2 + hh][lane * 4];
    };
    // h_t of tile tl (complete in LDS after the barrier) -> aout[t][tile rows][d*128 ..] as whole
    // 512-byte rows: thread f covers row f/32, 16-byte column f%32 for f = tid and tid+256.
    auto store_h = [&](int s, int tl) {
        const int t = d ? T_POS - 1 - s : s;
        float *base = p.aout + ((size_t)t * p.n_pad + (size_t)(tile0 + tl) * 16) * (2 * HID) + d * HID;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int f = h2 * 256 + tid, row = f >> 5, c4 = f & 31;
            *(f32x4 *)(base + (size_t)row * (2 * HID) + c4 * 4) = *(const f32x4 *)&hbuf[tl][row][c4 * 4];
        }
    };
    auto gates_only = [&](const f32x4 (&acc)[8], int tl) {
        f32x4 zq[4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            load_zq(zq, hh, tl);
#pragma unroll
            for (int r = 0; r < 4; ++r) gate_elem(acc, zq, hh * 4 + r, tl);
        }
    };
    // MFMAs of tile tm with the gate stream of tile tg threaded through them by hand.  Chunk k4 =
    // 32 MFMAs (one ds_read_b128 worth of K) + gate element k4, cut into eight stages of <= 6
    // VALU/transcendental instructions; one stage follows every fourth MFMA and a sched_barrier pins
    // it there.  (Left to itself the scheduler clusters 10-50 VALU instructions between MFMAs and the
    // matrix pipe idles: measured 1360-1480 cycles per chunk instead of 1024.)
    struct GateRegs { float p[4], e4[4], c, t; };
    auto gate_stage = [&](GateRegs &q, const f32x4 (&acc)[8], const f32x4 (&zq)[4], int e, int tl, int stage) {
        const int hh = e >> 2, r = e & 3;
        constexpr float L2E = 1.44269504088896340736f;
        switch (stage) {
            case 0:  // pre-activations = recurrent product + x-projection
#pragma unroll
                for (int g = 0; g < 4; ++g) q.p[g] = acc[2 * g + hh][r] + zq[g][r];
                break;
            case 1:  // exp2 arguments: sigmoid uses e^-x, tanh (g gate) uses e^2x
                q.p[0] *= -L2E; q.p[1] *= 2.0f * L2E; q.p[2] *= -L2E; q.p[3] *= -L2E;
                break;
            case 2:
#pragma unroll
                for (int g = 0; g < 4; ++g) q.e4[g] = __builtin_amdgcn_exp2f(q.p[g]);
                break;
            case 3:
#pragma unroll
                for (int g = 0; g < 4; ++g) q.e4[g] = 1.0f + q.e4[g];
                break;
            case 4:
#pragma unroll
                for (int g = 0; g < 4; ++g) q.p[g] = fast_rcp(q.e4[g]);   // sig(i), 1/(1+e^2g), sig(f), sig(o)
                break;
            case 5: {
                const float gg = 1.0f - 2.0f * q.p[1];                      // tanh(g)
                q.c = q.p[2] * cst[tl][e] + q.p[0] * gg;
                cst[tl][e] = q.c;
                q.t = __builtin_amdgcn_exp2f(q.c * (2.0f * L2E));
                break;
            }
            case 6:
                q.t = fast_rcp(1.0f + q.t);
                break;
            default: {
                const float h = q.p[3] * (1.0f - 2.0f * q.t);               // sig(o) * tanh(c)
                hbuf[tl][lq * 4 + r][w * 32 + hh * 16 + li] = h;
                break;
            }
        }
    };
    auto fused_phase = [&](f32x4 (&accm)[8], int tm, const f32x4 (&accg)[8], int tg, bool prof_on, int pbase) {
        (void)prof_on; (void)pbase;
        LSTM_PROF(pbase + 0);
        f32x4 afr[8], zq[4];
        const float *hrow = &hbuf[tm][li][lq * 32];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) afr[k4] = *(const f32x4 *)(hrow + k4 * 4);
        load_zq(zq, 0, tg);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) accm[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        LSTM_PROF(pbase + 1);
        GateRegs gr;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            if (k4 == 4) load_zq(zq, 1, tg);          // elements 4..7 use the hh = 1 fragments
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const int j = m >> 3, nb = m & 7;
                accm[nb] = mfma16(afr[k4][j], Bw[nb][k4 * 4 + j], accm[nb]);
                if ((m & 3) == 3) {
                    gate_stage(gr, accg, zq, k4, tg, m >> 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            LSTM_PROF(pbase + 2 + k4);
        }
    };

    f32x4 acc0[8], acc1[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc0[nb] = acc1[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // step 0 (h = 0): the pre-activation is the x-projection alone
    fetch_zx(0, 0);
    fetch_zx(0, 1);
    phase_barrier();
    gates_only(acc0, 0);
    phase_barrier();
    for (int s = 1; s < T_POS; ++s) {
        // phase A: tile 0 matrix work for step s, tile 1 gates for step s-1
        store_h(s - 1, 0);
        fetch_zx(s, 0);
        fused_phase(acc0, 0, acc1, 1, s == 8, 0);
        phase_barrier();
        // phase B: tile 1 matrix work for step s, tile 0 gates for step s
        store_h(s - 1, 1);
        fetch_zx(s, 1);
        fused_phase(acc1, 1, acc0, 0, s == 8, 10);
        phase_barrier();
        { const bool prof_on = s == 8; (void)prof_on; LSTM_PROF(20); }
    }
    store_h(T_POS - 1, 0);
    gates_only(acc1, 1);
    phase_barrier();
    store_h(T_POS - 1, 1);
}

}  // namespace clair
