// BiLSTM layer 2 as ONE launch: the input-projection GEMM (gemm_split.hip.h) and the recurrence (lstm32.hip.h) of a batch run side
// by side, the recurrent workgroups consuming each zx block as soon as the projection has written it (clair/model.py:443-450).
// What it buys is TIME on handles with one or two slots (104 us instead of 47 + 78 at batch 1024); the bytes still make the trip:
// the L2 does not allocate a full-line store that misses, so the block goes to memory and the reader's first touch brings it back
// (fetch 208 MB + write 174 MB per 1024-batch against 186 + 173 for the two launches; DESIGN.md section 6,
// profiles/r02_lstm2_fused.txt).
//
//   * workgroups [0, P) are projection workgroups, [P, P + C) recurrent ones.  Workgroups reach the CUs in id order and the
//     projection never waits for anyone, so every block a recurrent workgroup waits for comes from a workgroup that is running or
//     done: no co-residency assumption, no deadlock, whatever else is on the chip.
//   * a launch's blocks go round the eight XCDs: block b runs on XCD (b + r) % 8, with a rotation r that depends on the queue and
//     on what it dispatched before (tools/ubench/xcc_probe.hip; alone on the chip r = 0, MI355X_MICROARCH.md; HIP promises
//     nothing).  So a workgroup takes its place from where it IS: XCD x = HW_REG_XCC_ID and its row blockIdx / 8 give the logical
//     id 8 * (blockIdx / 8) + x -- the eight blocks of a row land on eight different XCDs and rows arrive in order on each.  Pair q
//     of candidate tiles belongs to XCD q % 8: its projection items and its four recurrent workgroups (2 tiles x 2 directions) all
//     carry logical ids = q mod 8, so the zx block written by a projection wave is read by a recurrent wave of the same XCD
//     (non-temporal loads: L1 bypassed), i.e. through the SAME L2 its stores went through: that L2 is what orders the block's
//     stores (retired before the ticket is written) against the reader's loads (issued after the ticket is seen).
//     Every workgroup CLAIMS its logical id (atomic exchange of the pass's ticket): a second claimant means the placement rule
//     does not hold, *error is raised and the engine refuses the result; every wait is bounded (a block whose producer never
//     came raises the same word after ~50 ms instead of hanging the GPU).
//   * per (direction, tile, t) block there are eight ticket words, one per producing wave (two projection workgroups x four
//     waves); a wave writes the forward pass's ticket (relaxed, agent scope: written through) after its stores of the block have
//     retired (s_waitcnt on its own in-order counter).  Recurrent wave w needs exactly words 2w and 2w+1: one 8-byte relaxed
//     agent load per step, issued a step ahead, spun on only when the producer is behind.
//   * both directions' projection workgroups walk time in their own direction (t = 0..32 / 32..0), a step's items spread over the
//     XCD's groups, so production runs just ahead of consumption.
#pragma once

#include "gemm_split.hip.h"
#include "lstm32.hip.h"

namespace clair {

struct Lstm2FusedArgs {
    GemmSplitArgs g;
    Lstm32Args l;
    FuseArgs f;
    int producers;   // P = 32 * g.groups
};

// recurrent workgroup c (0-based behind the producers): XCD c % 8 = pair % 8
__device__ __forceinline__ void fused_consumer_coords(int c, int &d, int &tile) {
    const int x = c & 7, j = c >> 3, r = j & 3, q = (j >> 2) * 8 + x;
    tile = 2 * q + (r >> 1);
    d = r & 1;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm2_fused_kernel(Lstm2FusedArgs a) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int block = (int)(blockIdx.x & ~7u) + (int)(xcc & 7u);
    if (threadIdx.x == 0 && atomicExch(a.f.claims + block, a.f.ticket) == a.f.ticket) *a.f.error = 1u;
    if (block < a.producers) {
        gemm_split_body<true>(a.g, block, a.f);
    } else {
        int d, tile;
        fused_consumer_coords(block - a.producers, d, tile);
        if (tile >= a.l.ntiles) return;
        lstm32_body<false, true>(a.l, d, tile, a.f);
    }
}

}  // namespace clair
