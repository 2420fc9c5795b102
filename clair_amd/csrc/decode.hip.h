// Device decode: probabilities -> call records (include/clair_call.h), one wavefront per candidate.
//
// The reference turns a candidate's four softmax vectors into a variant call by forming ten outcome families -- 1 179 float32
// products (clair/call_var.py:589-690, possible_outcome_probabilites_from) -- and picking the best with an iterative arg-max whose
// membership tests are EXACT float equalities (output_from, :693-947: `max(...) in family`), falling through outcomes it cannot
// write down; genotype, depth, supporting reads and the probability behind QUAL follow from the pick (output_with, :1002-1166).
// This kernel does all of that where the probabilities already are, so the 360 bytes per candidate need not cross the host link and
// the host is left with text (clair_host_format_calls).  Its twin is clair_host_resolve_calls (clair_amd/hostsrc/host_decode.cpp):
// the two write the same 32 bytes bit for bit, which is what tests/test_decode_gpu.py checks.
//
// Exactness.  Every product is formed by plain float32 multiplies in the reference's operand order ((a*b)*c is two roundings, as
// NumPy's element-wise products are; nothing here is of the a*b+c shape, and contraction is switched off regardless); sums follow
// NumPy's pairwise order for eight elements; float32 denormals are kept (the default on gfx9; the test feeds products that
// underflow); equality tests are exact.  The 64 lanes share the 1 179 products (19 slots of 64, families aligned to slot
// boundaries: element i of a family lives in lane i % 64 of slot i / 64, so "the first index equal to the best" is the lowest set bit
// of the lowest non-empty ballot); everything else is wave-uniform control flow that all lanes execute alike.
#pragma once
#include "common.hip.h"
#include "../../include/clair_call.h"

namespace clair {

struct DecodeArgs {
    const float *x;            // [n_pad][33][8][4] network input (channels 1..3 minus channel 0)
    const float *probs;        // [n][90]  gt21 (21) | genotype (3) | len1 (33) | len2 (33)
    const unsigned char *centre;   // [n][2]  reference window: centre character, min(length, 255)
    clair_call_t *calls;       // [n]
    int n;
};

constexpr int DEC_SLOTS = 19;   // small families | ACGT_INS | ACGT_DEL | INSINS x4 | DELDEL x4 | INSDEL x8

__device__ __forceinline__ float dec_wave_max(float v) {
#pragma unroll
    for (int sh = 32; sh; sh >>= 1) v = fmaxf(v, __shfl_xor(v, sh));
    return v;
}
__device__ __forceinline__ int dec_first(unsigned long long m) { return m ? (int)__builtin_ctzll(m) : -1; }

__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs p) {
#pragma clang fp contract(off)
    __shared__ float pr_s[4][96];
    __shared__ float xs_s[4][17 * 32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cand = blockIdx.x * 4 + w;
    if (cand >= p.n) return;                    // whole waves leave; nothing below synchronises across waves
    float *pr = pr_s[w], *xs = xs_s[w];
    {   // stage: the candidate's 90 probabilities and window positions 16..32 of its input (544 floats)
        const float *src = p.probs + (size_t)cand * OUT_FLOATS;
        pr[lane] = src[lane];
        if (lane + 64 < OUT_FLOATS) pr[lane + 64] = src[lane + 64];
        const float *xsrc = p.x + (size_t)cand * (T_POS * F_IN) + 16 * F_IN;
        for (int i = lane; i < 17 * 32; i += 64) xs[i] = xsrc[i];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);              // the wave's own LDS writes are visible to its own lanes after the wait
    const float *g = pr, *z = pr + 21, *l1 = pr + 24, *l2 = pr + 57;
    auto xat = [&](int pos, int row, int ch) { return xs[((pos - 16) * 8 + row) * 4 + ch]; };   // pos 16..32; ch 0 ref, 1 ins, 2 del, 3 snp
    auto sum_rows = [&](int pos, int ch) {      // NumPy's pairwise order for eight elements
        return ((xat(pos, 0, ch) + xat(pos, 1, ch)) + (xat(pos, 2, ch) + xat(pos, 3, ch))) + ((xat(pos, 4, ch) + xat(pos, 5, ch)) + (xat(pos, 6, ch) + xat(pos, 7, ch)));
    };

    clair_call_t c;
    c.status = 0; c.family = 0; c.index = 0; c.flags = 0; c.gt = 255; c.gi = 255; c.alt_b0 = 255; c.alt_b1 = 255; c.ins_avail = 0; c.reserved0 = 0;
    c.ins_code = 0; c.depth = 0.0f; c.support = 0.0f; c.p_call = 0.0f; c.rounds = 0;
    const unsigned char ref0 = p.centre[2 * (size_t)cand];
    const int seq_len = p.centre[2 * (size_t)cand + 1];
    const int ref_num = ref0 == 'A' ? 0 : ref0 == 'C' ? 1 : ref0 == 'G' ? 2 : (ref0 == 'T' || ref0 == 'U') ? 3 : -1;   // shared/utils.py:19-23 on ACGTU
    float depth = 0.0f;
    if (ref_num >= 0) {                          // call_var.py:1018, :1022-1024
        float d[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = xat(16, r, 2) + xat(16, r, 0);
        depth = ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
    }
    if (ref_num < 0 || depth == 0.0f) {
        if (lane == 0) p.calls[cand] = c;
        return;
    }
    c.depth = depth;
    c.status = CLAIR_CALL_RESOLVED;
    {   // the tensor's vote on the inserted base at positions 17..32 (:428-447): lane k < 16 votes for position 17 + k
        const int pos = 17 + (lane & 15);
        float v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = (xat(pos, b, 1) + xat(pos, b + 4, 1)) - (xat(pos, b, 3) + xat(pos, b + 4, 3));
        int best = 0;
        float bv = v[0];
#pragma unroll
        for (int b = 1; b < 4; ++b)
            if (v[b] > bv) { bv = v[b]; best = b; }
        if (0.0f > bv) best = 0;                 // entries 4..7 of the vote vector are zero: the first of them wins, and 4 % 4 is base 0
        const unsigned long long b0 = __ballot((best & 1) && lane < 16), b1 = __ballot((best & 2) && lane < 16);
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) code |= (unsigned)((b0 >> k) & 1) << (2 * k) | (unsigned)((b1 >> k) & 1) << (2 * k + 1);
        c.ins_code = code;
        // an insertion of 16 or more: positions 17..31 always, 32 under the read-support condition (:487-497)
        float v32[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) v32[b] = (xat(32, b, 1) + xat(32, b + 4, 1)) - (xat(32, b, 3) + xat(32, b + 4, 3));
        const float s8 = ((v32[0] + v32[1]) + (v32[2] + v32[3])) + ((0.0f + 0.0f) + (0.0f + 0.0f));
        c.ins_avail = (unsigned char)(15 + ((double)s8 >= 0.125 * (double)sum_rows(32, 0) ? 1 : 0));
    }

    // ---- the ten outcome families (:589-690), products left to right as the reference writes them ----
    const int HOMO_IDX[4] = {0, 4, 7, 9}, HET_IDX[6] = {1, 2, 3, 5, 6, 8};
    const float p_ref = z[0], p_hom = z[1], p_het = z[2];
    const float z1 = l1[16], z2 = l2[16], zero = z1 * z2;
    const float e_homins = p_hom * g[15], e_insins = p_het * g[15], e_homdel = p_hom * g[10], e_deldel = p_het * g[10], e_insdel = p_het * g[20];
    float v[DEC_SLOTS];
    unsigned alive = 0;                          // bit s: this lane's element of slot s is a live outcome
    {
        float s0 = 0.0f;
        bool live = true;
        if (lane == 0) s0 = (zero * p_ref) * g[HOMO_IDX[ref_num]];
        else if (lane <= 4) s0 = (zero * p_hom) * g[HOMO_IDX[lane - 1]];
        else if (lane <= 10) s0 = (zero * p_het) * g[HET_IDX[lane - 5]];
        else if (lane <= 26) { const int i = lane - 11; s0 = (l1[17 + i] * l2[17 + i]) * e_homins; }
        else if (lane <= 42) { const int i = lane - 27; s0 = (l1[15 - i] * l2[15 - i]) * e_homdel; }
        else live = false;
        v[0] = s0;
        alive |= live ? 1u : 0u;
        const int i = lane >> 2, k = lane & 3;
        const float a = z1 * l2[17 + i], b = l1[17 + i] * z2, one_ins = a > b ? a : b;       // np.maximum
        const float cc = z1 * l2[15 - i], dd = l1[15 - i] * z2, one_del = cc > dd ? cc : dd;
        v[1] = (one_ins * g[16 + k]) * p_het;
        v[2] = (one_del * g[11 + k]) * p_het;
        alive |= 6u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {           // INSINS: index (i, j) = 16 i + j
            const int idx = s * 64 + lane;
            v[3 + s] = (l1[17 + (idx >> 4)] * l2[17 + (idx & 15)]) * e_insins;
            alive |= 1u << (3 + s);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {           // DELDEL: pairs (i, j), j != i, in list order: index 15 i + (j < i ? j : j - 1)
            const int idx = s * 64 + lane, ii = idx / 15, jj = idx - ii * 15, j = jj < ii ? jj : jj + 1;
            const bool in = idx < 240;
            v[7 + s] = in ? (l1[15 - ii] * l2[15 - j]) * e_deldel : 0.0f;
            alive |= in ? 1u << (7 + s) : 0u;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {           // INSDEL: ((i, j), which): even = insertion i with deletion j, odd = deletion i with insertion j
            const int idx = s * 64 + lane, pair = idx >> 1, ii = pair >> 4, j = pair & 15;
            v[11 + s] = (idx & 1) ? (l1[15 - ii] * l2[17 + j]) * e_insdel : (l1[17 + ii] * l2[15 - j]) * e_insdel;
            alive |= 1u << (11 + s);
        }
    }
    const float v_ref = __shfl(v[0], 0);

    // ---- iterative arg-max with exact-equality membership (:693-947) ----
    auto deletion_length = [&](int length) {     // characters of reference behind the centre a deletion of `length` can show
        const int a = 17 < seq_len ? 17 : seq_len, b = 17 + length < seq_len ? 17 + length : seq_len;
        return b > a ? b - a : 0;
    };
    unsigned flags = 0;
    int family = CLAIR_F_REF, index = 0;
    bool consulted = false, multi = false, same = false;
    for (;;) {
        ++c.rounds;
        float m = 0.0f;                          // an emptied family counts as 0 (`max(...) if len(...) else 0`); every product is >= 0
#pragma unroll
        for (int s = 0; s < DEC_SLOTS; ++s)
            if ((alive >> s) & 1u) m = fmaxf(m, v[s]);
        const float best = dec_wave_max(m);
        if (best == v_ref) { flags = 1u << CLAIR_F_REF; family = CLAIR_F_REF; index = 0; break; }
        // who holds the best?  ballots per slot; the small families share slot 0
        unsigned long long hit[DEC_SLOTS];
#pragma unroll
        for (int s = 0; s < DEC_SLOTS; ++s) hit[s] = __ballot(((alive >> s) & 1u) && v[s] == best);
        int first[CLAIR_F_COUNT];
        first[CLAIR_F_REF] = -1;
        first[CLAIR_F_HOMO_SNP] = dec_first(hit[0] & 0x1eull) - 1;
        first[CLAIR_F_HET_SNP] = dec_first(hit[0] & 0x7e0ull) - 5;
        first[CLAIR_F_HOMO_INS] = dec_first(hit[0] & 0x7fff800ull) - 11;
        first[CLAIR_F_HOMO_DEL] = dec_first(hit[0] & 0x7fff8000000ull) - 27;
        if (!(hit[0] & 0x1eull)) first[CLAIR_F_HOMO_SNP] = -1;
        if (!(hit[0] & 0x7e0ull)) first[CLAIR_F_HET_SNP] = -1;
        if (!(hit[0] & 0x7fff800ull)) first[CLAIR_F_HOMO_INS] = -1;
        if (!(hit[0] & 0x7fff8000000ull)) first[CLAIR_F_HOMO_DEL] = -1;
        first[CLAIR_F_ACGT_INS] = dec_first(hit[1]);
        first[CLAIR_F_ACGT_DEL] = dec_first(hit[2]);
        auto first_of = [&](int s0, int ns) {
            int f = -1;
#pragma unroll
            for (int s = 7; s >= 0; --s)
                if (s < ns && hit[s0 + s]) f = s * 64 + dec_first(hit[s0 + s]);
            return f;
        };
        first[CLAIR_F_INSINS] = first_of(3, 4);
        first[CLAIR_F_DELDEL] = first_of(7, 4);
        first[CLAIR_F_INSDEL] = first_of(11, 8);
        flags = 0;
        family = -1;
#pragma unroll
        for (int k = 1; k < CLAIR_F_COUNT; ++k)
            if (first[k] >= 0) { flags |= 1u << k; if (family < 0) family = k; }
        if (family < 0) {   // nothing equals the best: a NaN among the probabilities (the reference's max() is undefined there): no call
            c.status = 0; c.family = 0; c.index = 0; c.flags = 0; c.gt = 255; c.gi = 255; c.alt_b0 = 255; c.alt_b1 = 255; c.ins_avail = 0;
            c.ins_code = 0; c.depth = 0.0f; c.support = 0.0f; c.p_call = 0.0f; c.rounds = 0;
            if (lane == 0) p.calls[cand] = c;
            return;
        }
        index = first[family];
        bool have = false;
        c.alt_b0 = c.alt_b1 = 255;
        multi = false;
        int dead_slot = -1, dead_lane = 0;       // the outcome this round consumes
        if (family == CLAIR_F_HOMO_SNP) {         // :60-62
            int bi = 0;
            for (int k = 1; k < 4; ++k)
                if (g[HOMO_IDX[k]] > g[HOMO_IDX[bi]]) bi = k;
            c.alt_b0 = (unsigned char)bi;
            same = "ACGT"[bi] == (char)ref0;
            have = true;
        } else if (family == CLAIR_F_HET_SNP) {   // :65-67
            int bi = 0;
            for (int k = 1; k < 6; ++k)
                if (g[HET_IDX[k]] > g[HET_IDX[bi]]) bi = k;
            const int B1[6] = {0, 0, 0, 1, 1, 2}, B2[6] = {1, 2, 3, 2, 3, 3};      // AC AG AT CG CT GT
            const int b1 = B1[bi], b2 = B2[bi];
            const bool n1 = "ACGT"[b1] != (char)ref0, n2 = "ACGT"[b2] != (char)ref0;
            if (n1 && n2) { c.alt_b0 = (unsigned char)b1; c.alt_b1 = (unsigned char)b2; multi = true; }
            else c.alt_b0 = (unsigned char)(n1 ? b1 : b2);
            have = true;
        } else if (family == CLAIR_F_HOMO_INS) {
            dead_slot = 0; dead_lane = 11 + index;
            consulted |= index + 1 >= 16;
            have = true;
        } else if (family == CLAIR_F_ACGT_INS) {
            dead_slot = 1; dead_lane = index;
            consulted |= index / 4 + 1 >= 16;
            c.alt_b0 = (unsigned char)(index & 3);
            multi = "ACGT"[index & 3] != (char)ref0;
            have = true;
        } else if (family == CLAIR_F_INSINS) {
            dead_slot = 3 + (index >> 6); dead_lane = index & 63;
            const int i = index / 16 + 1, j = index % 16 + 1, short_ = i <= j ? i : j, long_ = i <= j ? j : i;
            consulted = true;                     // a long allele, or the second allele's look-up (:805-823)
            const int eff = long_ < 16 ? long_ : c.ins_avail;
            have = (short_ < eff ? short_ : eff) < eff;
            multi = true;
        } else if (family == CLAIR_F_HOMO_DEL) {
            dead_slot = 0; dead_lane = 27 + index;
            consulted |= index + 1 >= 16;
            have = deletion_length(index + 1) > 0;
        } else if (family == CLAIR_F_ACGT_DEL) {
            dead_slot = 2; dead_lane = index;
            const int length = index / 4 + 1;
            consulted |= length >= 16;
            have = deletion_length(length) > 0;
            c.alt_b0 = (unsigned char)(index & 3);
            multi = "ACGT"[index & 3] != (char)ref0;
        } else if (family == CLAIR_F_DELDEL) {
            dead_slot = 7 + (index >> 6); dead_lane = index & 63;
            const int i = index / 15 + 1, jj = index % 15, j = (jj < i - 1 ? jj : jj + 1) + 1, short_ = i < j ? i : j, long_ = i < j ? j : i;
            consulted |= long_ >= 16;
            have = deletion_length(long_) > short_;
            multi = true;
        } else {                                   // INSDEL
            dead_slot = 11 + (index >> 6); dead_lane = index & 63;
            const int i = (index / 2) / 16 + 1, j = (index / 2) % 16 + 1, del_len = index % 2 == 0 ? j : i, ins_len = index % 2 == 0 ? i : j;
            consulted |= ins_len >= 16 || del_len >= 16;
            have = deletion_length(del_len) > 0;
            multi = true;
        }
        if (dead_slot >= 0 && lane == dead_lane) alive &= ~(1u << dead_slot);
        if (have) break;
    }
    c.family = (unsigned char)family;
    c.index = (unsigned short)index;
    c.flags = (unsigned short)flags;
    auto flag = [&](int k) { return (flags >> k) & 1u; };
    // ---- the numeric half of output_with (:1002-1166) ----
    const bool is_ref = flag(CLAIR_F_REF);
    const bool hetero_call = flag(CLAIR_F_HET_SNP) || flag(CLAIR_F_ACGT_INS) || flag(CLAIR_F_INSINS) || flag(CLAIR_F_ACGT_DEL) || flag(CLAIR_F_DELDEL);
    int gt = 255;
    if (is_ref) gt = 0;
    else if (flag(CLAIR_F_HOMO_SNP) || flag(CLAIR_F_HOMO_INS) || flag(CLAIR_F_HOMO_DEL)) gt = 1;
    else if (hetero_call) gt = 2;
    if (multi) gt = 3;
    c.gt = (unsigned char)gt;
    c.status |= (consulted ? CLAIR_CALL_CONSULTED : 0) | (multi ? CLAIR_CALL_MULTI : 0) | (same && !is_ref ? CLAIR_CALL_SAME : 0);
    auto snp_reads = [&](int b) { return ((xat(16, b, 3) + xat(16, b + 4, 3)) + xat(16, b, 0)) + xat(16, b + 4, 0); };   // :1100-1107
    float support = 0.0f;
    if (is_ref) support = xat(16, ref_num, 0) + xat(16, ref_num + 4, 0);
    else if (flag(CLAIR_F_HOMO_SNP) || flag(CLAIR_F_HET_SNP)) {
        support = support + snp_reads(c.alt_b0);
        if (c.alt_b1 != 255) support = support + snp_reads(c.alt_b1);
    } else {
        const float ins_reads = sum_rows(17, 1) - sum_rows(17, 3);
        const float del_reads = sum_rows(17, 2);
        if (flag(CLAIR_F_HOMO_INS) || flag(CLAIR_F_INSINS)) support = ins_reads;
        else if (flag(CLAIR_F_ACGT_INS)) support = multi ? ins_reads + snp_reads(c.alt_b0) : ins_reads;
        else if (flag(CLAIR_F_HOMO_DEL) || flag(CLAIR_F_DELDEL)) support = del_reads;
        else if (flag(CLAIR_F_ACGT_DEL)) support = multi ? del_reads + snp_reads(c.alt_b0) : del_reads;
        else if (flag(CLAIR_F_INSDEL)) support = (sum_rows(17, 1) + sum_rows(17, 2)) - sum_rows(17, 3);
    }
    c.support = support;
    if (gt != 255) {   // gt21 class of the call from allele kinds (task/gt21.py:60-110): 0..3 base, 4 Ins, 5 Del, 254 not a base
        const bool has0 = gt == 0 || gt == 2;
        const int kref = is_ref ? ref_num : (ref0 == 'A' ? 0 : ref0 == 'C' ? 1 : ref0 == 'G' ? 2 : ref0 == 'T' ? 3 : 254);
        int k0, k1;
        switch (family) {
            case CLAIR_F_REF: k0 = k1 = ref_num; break;
            case CLAIR_F_HOMO_SNP: case CLAIR_F_HET_SNP:
                if (multi) { k0 = c.alt_b0; k1 = c.alt_b1; } else { k1 = c.alt_b0; k0 = has0 ? kref : k1; }
                break;
            case CLAIR_F_HOMO_INS: k1 = 4; k0 = has0 ? kref : 4; break;
            case CLAIR_F_ACGT_INS: k1 = 4; k0 = multi ? (int)c.alt_b0 : (has0 ? kref : 4); break;
            case CLAIR_F_INSINS: k0 = k1 = 4; break;
            case CLAIR_F_HOMO_DEL: k1 = 5; k0 = has0 ? kref : 5; break;
            case CLAIR_F_ACGT_DEL: if (multi) { k0 = 5; k1 = c.alt_b0; } else { k1 = 5; k0 = has0 ? kref : 5; } break;
            case CLAIR_F_DELDEL: k0 = k1 = 5; break;
            default: k0 = 5; k1 = 4; break;
        }
        const int PAIR[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
        int gi;
        if (k0 == 254 || k1 == 254) gi = 255;
        else if (k0 < 4 && k1 < 4) gi = PAIR[k0][k1];
        else if (k0 < 4 || k1 < 4) { const int base = k0 < 4 ? k0 : k1, other = k0 < 4 ? k1 : k0; gi = (other == 4 ? 16 : 11) + base; }
        else if (k0 == k1) gi = k0 == 4 ? 15 : 10;
        else gi = 20;
        c.gi = (unsigned char)gi;
        if (gi != 255) c.p_call = g[gi] * z[gt == 0 ? 0 : (gt == 1 ? 1 : 2)];
    }
    if (lane == 0) p.calls[cand] = c;
}

}  // namespace clair
