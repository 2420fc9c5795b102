// Device front end of the C ABI (include/clair_amd.h, "front end"): packed alignments (include/clair_reads.h) -> candidate sites ->
// pileup count windows [n][33][8][4] int16 that stay in HBM for clair_submit_ex.  SURVEY.md 8(f) N4, the GPU half.
//
// What the reference does per read base in two interpreters (dataPrepScripts/ExtractVariantCandidates.py:296-316: a dict of
// per-position tallies; CreateTensor.py:289-365 + :29-65: a tuple per (read base, open window), summed when the window is written)
// is done here per reference POSITION: one thread per read base adds to per-position tables with atomics (pass 1), the candidate
// filter runs over the tallies, one thread per read base counts the windows it lies in and scatters the inserted bases (pass 2), and
// the windows are assembled from 33 table columns each.  Why that is the same result -- and when it is not (a tuple budget that
// binds, unsorted input, bases outside the IUPAC alphabet: reported as CLAIR_FE_* bits, the caller then runs the sequential host
// code) -- is in oracle/frontend_np.py, the NumPy restatement these kernels are tested against, and in DESIGN.md 6b.
//
// All integer work, HBM/atomic bound: nothing here is a matrix product.  Tables are sized for the whole region and the packed reads
// stay resident (a 10 Mb region at 50x: ~1 GB of tables, ~1.3 GB of reads; a whole chromosome fits the 288 GB many times over), so
// there is no streaming window to manage and pass 2 re-reads the reads from HBM instead of from the host.
#include "../../include/clair_amd.h"
#include "../../include/clair_reads.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_fe_error;

constexpr int N_POS = 33, N_ROW = 8, N_QUAD = N_POS * N_ROW;   // 264 (position, row) quads of four channels
constexpr int WINDOW_VALUES = N_QUAD * 4;                        // 1056

// IUPAC_base_to_num_dict (shared/utils.py:24-27) and IUPAC_base_to_ACGT_base_dict through evc_base_from (:19-22, 27-28; N stays N = 6);
// 255: not a key
struct BaseTables {
    uint8_t pile[256], evc[256];
    constexpr BaseTables() : pile(), evc() {
        for (int i = 0; i < 256; ++i) { pile[i] = 255; evc[i] = 255; }
        const char keys[17] = "ACGTURYSWKMBDHVN";
        const uint8_t rows[16] = {0, 1, 2, 3, 3, 0, 1, 1, 0, 2, 0, 1, 0, 0, 0, 0};
        const char acgt[17] = "ACGTTACCAGACAAAA";
        for (int i = 0; i < 16; ++i) {
            pile[(unsigned char)keys[i]] = rows[i];
            evc[(unsigned char)keys[i]] = (uint8_t)(acgt[i] == 'A' ? 0 : acgt[i] == 'C' ? 1 : acgt[i] == 'G' ? 2 : 3);
        }
        evc[(unsigned char)'N'] = 6;
    }
};
__constant__ BaseTables BASES = BaseTables();

struct Region {          // what every kernel needs to know about the tables and the reference
    int64_t lo, n;       // tables cover 0-based positions [lo, lo + n)
    const uint8_t *ref;  // upper-cased reference bytes for [ref0, ref0 + ref_len)
    int64_t ref0, ref_len;
    uint32_t *ev;        // [n][8]  A C G T I D N -          candidate-search tallies
    uint32_t *q;         // [n][8]  read-base row + 4 * strand, M of pileup reads
    uint32_t *misc;      // [n][8]  0,1 M per strand | 2,3 D per strand (rp > POS) | 4 D at rp == POS | 5 inserted bases (rp > POS) | 6 M at rp == POS
    uint32_t *anomalies;
};

struct Slab {
    clair_read_t *reads = nullptr;
    clair_op_t *ops = nullptr;
    uint32_t *op_elem = nullptr;
    uint8_t *seq = nullptr;
    uint64_t *tuples = nullptr;      // per read, pass 2
    int64_t n_reads = 0, n_ops = 0, n_elem = 0, seq_bytes = 0;
};

struct SlabView {
    const clair_read_t *reads;
    const clair_op_t *ops;
    const uint32_t *op_elem;
    const uint8_t *seq;
    uint64_t *tuples;
    uint32_t n_ops;
    uint32_t n_elem;
};

// the read base / deleted base a thread stands for
struct Element {
    clair_read_t r;
    int64_t rp;
    uint32_t qp, k, code, read;
};

__device__ inline Element element_of(const SlabView &s, uint32_t e) {
    uint32_t a = 0, b = s.n_ops;             // largest j with op_elem[j] <= e
    while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (s.op_elem[mid] <= e) a = mid; else b = mid;
    }
    const clair_op_t op = s.ops[a];
    Element el;
    el.read = op.read;
    el.r = s.reads[op.read];
    el.k = e - s.op_elem[a];
    el.code = op.code_len & 3u;
    el.rp = el.r.pos0 + op.ref_off + (el.code == CLAIR_OP_I ? 0 : (int64_t)el.k);
    el.qp = op.q_off + (el.code == CLAIR_OP_D ? 0u : el.k);
    return el;
}

__device__ inline uint8_t ref_row(const Region &g, int64_t p) {
    const int64_t i = p - g.ref0;
    return (i >= 0 && i < g.ref_len) ? BASES.pile[g.ref[i]] : (uint8_t)255;
}

__device__ inline void flag(const Region &g, uint32_t bit) { atomicOr(g.anomalies, bit); }

// ---- pass 1: the per-position tables ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fe_tally_kernel(Region g, SlabView s) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= s.n_elem) return;
    const Element el = element_of(s, e);
    const bool evc = el.r.flags & CLAIR_READ_EVC, pile = el.r.flags & CLAIR_READ_PILE;
    const int so = (el.r.flags & CLAIR_READ_REVERSE) ? 1 : 0;
    const int64_t t = el.rp - g.lo;
    const bool inside = t >= 0 && t < g.n;
    if (el.code == CLAIR_OP_M) {
        if (el.qp >= el.r.seq_len) { flag(g, CLAIR_FE_SEQ_OVERRUN); return; }
        if (pile && inside && ref_row(g, el.rp) == 255) flag(g, CLAIR_FE_BAD_REF);
        const uint8_t base = s.seq[el.r.seq0 + el.qp];
        const uint8_t ei = BASES.evc[base];
        if (ei == 255) { flag(g, CLAIR_FE_BAD_BASE); return; }
        if (!inside) return;
        if (evc) atomicAdd(&g.ev[t * 8 + ei], 1u);
        if (pile) {
            atomicAdd(&g.q[t * 8 + BASES.pile[base] + 4 * so], 1u);
            atomicAdd(&g.misc[t * 8 + so], 1u);
            if (el.rp == el.r.pos0) atomicAdd(&g.misc[t * 8 + 6], 1u);
        }
    } else if (el.code == CLAIR_OP_I) {
        if (el.k == 0 && evc && t - 1 >= 0 && t - 1 < g.n) atomicAdd(&g.ev[(t - 1) * 8 + 4], 1u);   // once per operation, at the base before it
        if (!pile) return;
        if (el.qp >= el.r.seq_len) { flag(g, CLAIR_FE_SEQ_OVERRUN); return; }
        if (el.rp <= el.r.pos0) return;                                                              // no window is open yet (CreateTensor.py:326-341)
        if (BASES.pile[s.seq[el.r.seq0 + el.qp]] == 255) flag(g, CLAIR_FE_BAD_BASE);
        if (inside) atomicAdd(&g.misc[t * 8 + 5], 1u);
    } else {
        if (el.k == 0 && evc && t - 1 >= 0 && t - 1 < g.n) atomicAdd(&g.ev[(t - 1) * 8 + 5], 1u);
        if (!pile || !inside) return;
        if (ref_row(g, el.rp) == 255) flag(g, CLAIR_FE_BAD_REF);
        if (el.rp > el.r.pos0) atomicAdd(&g.misc[t * 8 + 2 + so], 1u);
        else atomicAdd(&g.misc[t * 8 + 4], 1u);
    }
}

// ---- the candidate filter (ExtractVariantCandidates.py:347-393) over the tallies: one flag per position -----------------------
struct CandidateRule {
    double min_depth, min_af;
    int64_t ctg_start, ctg_end;      // 1-based inclusive; -1: no range
    const int64_t *bed_start, *bed_end;
    int64_t n_bed;                   // -1: no bed file; intervals sorted, merged, 0-based half-open
};

__global__ __launch_bounds__(256) void fe_candidate_flags_kernel(Region g, CandidateRule rule, uint8_t *flags) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= g.n) return;
    const uint4 lo4 = *(const uint4 *)(g.ev + t * 8), hi4 = *(const uint4 *)(g.ev + t * 8 + 4);
    const uint32_t n[7] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z};
    uint8_t keep = 0;
    do {
        if (!(n[0] | n[1] | n[2] | n[3] | n[4] | n[5] | n[6])) break;       // not a key of the reference's dict
        const int64_t p0 = g.lo + t;
        if (rule.ctg_start >= 0 && !(rule.ctg_start <= p0 + 1 && p0 + 1 <= rule.ctg_end)) break;
        if (rule.n_bed >= 0) {
            int64_t a = 0, b = rule.n_bed;                                  // first interval starting after p0
            while (a < b) { const int64_t mid = (a + b) >> 1; if (rule.bed_start[mid] <= p0) a = mid + 1; else b = mid; }
            if (a == 0 || !(p0 < rule.bed_end[a - 1])) break;
        }
        int64_t i = p0 - g.ref0;
        if (i < 0) i += g.ref_len;                                          // Python's negative index
        if (i < 0 || i >= g.ref_len) break;
        const uint8_t rb = BASES.evc[g.ref[i]];
        if (rb == 255) break;
        int64_t depth = 0;
        for (int k = 0; k < 7; ++k) depth += n[k];
        depth -= (int64_t)n[4] + n[5];
        if ((double)depth < rule.min_depth) break;
        int first = 0;                                                      // a stable descending sort keeps dict order among equals
        for (int k = 1; k < 7; ++k) if (n[k] > n[first]) first = k;
        int second = first == 0 ? 1 : 0;
        for (int k = 0; k < 7; ++k) if (k != first && n[k] > n[second]) second = k;
        const int64_t denom = depth > 0 ? depth : 1;
        keep = (first != rb || (double)n[second] / (double)denom >= rule.min_af) ? 1 : 0;
    } while (false);
    flags[t] = keep;
}

__global__ __launch_bounds__(256) void fe_given_flags_kernel(const int64_t *positions, int64_t n_positions, int64_t lo, int64_t n, uint8_t *flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_positions) return;
    const int64_t t = positions[i] - 1 - lo;
    if (t >= 0 && t < n) flags[t] = 1;
}

// ---- flags -> exclusive prefix (uint32, n + 1 entries) and the list of flagged indices; 4096 flags per workgroup ---------------
constexpr int SCAN_ITEMS = 16, SCAN_BLOCK = 256 * SCAN_ITEMS;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *total) {   // 256 threads
    __shared__ uint32_t wave_sum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wave_sum[w] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int i = 0; i < 4; ++i) { if (i < w) before += wave_sum[i]; all += wave_sum[i]; }
    __syncthreads();
    *total = all;
    return before + x - v;
}

__global__ __launch_bounds__(256) void fe_block_count_kernel(const uint8_t *flags, int64_t n, uint32_t *block_sum) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t c = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) if (at + i < n) c += flags[at + i];
    uint32_t total;
    (void)block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void fe_scan_block_sums_kernel(uint32_t *block_sum, int64_t n_blocks, uint32_t *grand_total) {
    uint32_t carry = 0;                      // one workgroup walks the block sums 256 at a time
    for (int64_t at = 0; at < n_blocks; at += 256) {
        const int64_t i = at + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sum[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, &total);
        if (i < n_blocks) block_sum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

__global__ __launch_bounds__(256) void fe_scan_write_kernel(const uint8_t *flags, int64_t n, const uint32_t *block_sum, uint32_t *prefix,
                                                            int64_t *list, int64_t list_base) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint8_t f[SCAN_ITEMS];
    uint32_t c = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) { f[i] = at + i < n ? flags[at + i] : 0; c += f[i]; }
    uint32_t total;
    uint32_t run = block_sum[blockIdx.x] + block_exclusive_scan(c, &total);
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (at + i > n) break;
        if (prefix) prefix[at + i] = run;     // entry n = the total
        if (at + i < n && f[i]) { list[run] = list_base + at + i; ++run; }
    }
}

// ---- pass 2: windows per read base (the tuple counts), inserted bases into their windows ---------------------------------------
struct Candidates {
    const int64_t *centre;       // 1-based, ascending
    const uint32_t *before;      // before[t] = candidates with centre - 1 - lo < t; n + 1 entries
    uint32_t *ins;               // [n_candidates][33][8]
};

__device__ inline uint32_t before_at(const Region &g, const Candidates &c, int64_t t) {
    return c.before[t < 0 ? 0 : (t > g.n ? g.n : t)];
}

__global__ __launch_bounds__(256) void fe_windows_per_base_kernel(Region g, SlabView s, Candidates c) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    uint32_t read = 0xffffffffu;
    uint64_t nc = 0;
    if (e < s.n_elem) {
        const Element el = element_of(s, e);
        if (el.r.flags & CLAIR_READ_PILE) {
            // centres whose window is open while this base is offered (CreateTensor.py:296-361)
            int64_t a = -1, b = -2;
            if (el.code == CLAIR_OP_M) {
                a = el.rp > el.r.pos0 ? el.rp - 17 : el.r.pos0 - 16;
                b = el.rp + 17;
            } else if (el.rp > el.r.pos0) {
                a = el.rp - 17;
                b = el.rp + 16;
            }
            if (b >= a) nc = before_at(g, c, b - g.lo) - before_at(g, c, a - 1 - g.lo);
            read = el.read;
            if (el.code == CLAIR_OP_I && el.rp > el.r.pos0 && el.qp < el.r.seq_len) {
                const uint8_t row = BASES.pile[s.seq[el.r.seq0 + el.qp]];
                if (row != 255) {
                    const int so = (el.r.flags & CLAIR_READ_REVERSE) ? 4 : 0;
                    const uint32_t i0 = before_at(g, c, el.rp - 15 - 1 - g.lo), i1 = before_at(g, c, el.rp + 16 - g.lo);
                    for (uint32_t i = i0; i < i1; ++i) {       // generate_tensor :51-53: column min(idx + k, 32), channel 1
                        const int64_t col = el.rp - c.centre[i] + 17 + (int64_t)el.k;
                        atomicAdd(&c.ins[((size_t)i * N_POS + (col < N_POS - 1 ? col : N_POS - 1)) * N_ROW + row + so], 1u);
                    }
                }
            }
        }
    }
    // one atomic per run of equal reads in the wave (64 consecutive bases are nearly always one alignment's): 64 lanes adding to one
    // address serialise at the memory side, which made this pass four times the first one
    unsigned long long todo = __ballot(nc != 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t r = __shfl(read, leader, 64);
        const bool mine = nc != 0 && read == r;
        unsigned long long v = mine ? nc : 0;
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd((unsigned long long *)&s.tuples[r], v);
        todo &= ~__ballot(mine);
    }
}

// ---- per candidate: was its window ever opened, how many tuples did it hold, does it survive -----------------------------------
struct WindowRule {
    int min_coverage;
    int drop_non_iupac_centre;   // clair/utils.py:90-91
};

__global__ __launch_bounds__(256) void fe_window_flags_kernel(Region g, const int64_t *centre, int64_t n_candidates, WindowRule rule,
                                                              uint8_t *keep, uint64_t *window_tuples) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_candidates) return;
    const int64_t c = centre[i];
    uint64_t walked = 0, tuples = 0;
    uint32_t depth_centre = 0;
    for (int64_t rp = c - 17; rp <= c + 17; ++rp) {
        const int64_t t = rp - g.lo;
        if (t < 0 || t >= g.n) continue;
        const uint4 a = *(const uint4 *)(g.misc + t * 8), b = *(const uint4 *)(g.misc + t * 8 + 4);
        const uint64_t m = (uint64_t)a.x + a.y, d = (uint64_t)a.z + a.w;
        if (rp <= c + 16) walked += m + d + b.x;                    // M or D (incl. a read's first D) opens the window
        tuples += m;
        if (rp == c + 17) tuples -= b.z;                            // a read that STARTS there never opened this window
        if (rp >= c - 16) tuples += d + b.y;
        if (rp == c - 1) depth_centre = (uint32_t)m;
    }
    const bool opened = walked > 0;
    const int64_t nrp = c - g.ref0;
    bool ok = opened && nrp - 17 >= 0 && (int64_t)depth_centre >= (int64_t)rule.min_coverage;
    if (ok && rule.drop_non_iupac_centre) {
        const int64_t at = nrp - 17 + 16;                           // refseq[16] of the slice [nrp-17, nrp+16) clamped to the sequence
        ok = at < g.ref_len && BASES.pile[g.ref[at]] != 255;
    }
    keep[i] = ok ? 1 : 0;
    window_tuples[i] = opened ? tuples : 0;
}

// ---- assembly: 264 (position, row) quads per kept window ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fe_assemble_kernel(Region g, const int64_t *centre, const int64_t *kept, int64_t n_kept, const uint32_t *ins,
                                                          short4 *counts, int64_t *out_centre, uint8_t *out_refseq) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t o = id / N_QUAD;
    if (o >= n_kept) return;
    const int quad = (int)(id - o * N_QUAD), idx = quad >> 3, row = quad & 7, so = row >> 2, b = row & 3;
    const int64_t ci = kept[o], c = centre[ci];
    const int64_t rp = c - 17 + idx, t = rp - g.lo;
    uint32_t qv = 0, mw = 0, dw = 0;
    if (t >= 0 && t < g.n) {
        qv = g.q[t * 8 + row];
        mw = g.misc[t * 8 + so];
        dw = g.misc[t * 8 + 2 + so];
    }
    const bool is_ref = ref_row(g, rp) == b;
    const uint32_t ch0 = is_ref ? mw : 0, ch1 = qv + ins[((size_t)ci * N_POS + idx) * N_ROW + row];
    const uint32_t ch2 = is_ref ? mw + (idx >= 1 ? dw : 0) : 0, ch3 = qv;
    if ((ch0 | ch1 | ch2 | ch3) > 32767u) flag(g, CLAIR_FE_OVERFLOW);
    counts[id] = make_short4((short)ch0, (short)ch1, (short)ch2, (short)ch3);
    if (quad == 0) out_centre[o] = c;
    if (quad < 34) {                                                 // reference_sequence[nrp-17 : nrp+16], NUL-padded to 34
        const int64_t nrp = c - g.ref0, at = nrp - 17 + quad;
        out_refseq[o * 34 + quad] = (quad < 33 && at >= 0 && at < g.ref_len) ? g.ref[at] : (uint8_t)0;
    }
}

int fe_fail(clair_frontend *f, const char *fmt, ...);

}  // namespace

struct clair_frontend {
    int device = 0;
    hipStream_t stream = nullptr;
    Region g{};
    uint8_t *d_ref = nullptr;
    std::vector<Slab> slabs;
    // candidates
    uint8_t *d_flags = nullptr;          // per position
    uint32_t *d_before = nullptr;        // n + 1
    uint32_t *d_block_sum = nullptr;
    uint32_t *d_total = nullptr;         // [2]: candidates, kept windows
    int64_t *d_centre = nullptr;
    int64_t n_candidates = -1, cap_candidates = 0;
    int64_t *d_bed = nullptr;
    // windows
    uint32_t *d_ins = nullptr;
    uint8_t *d_keep = nullptr;
    uint64_t *d_window_tuples = nullptr;
    int64_t *d_kept = nullptr;
    uint32_t *d_cand_block_sum = nullptr;
    short4 *d_counts = nullptr;
    int64_t *d_out_centre = nullptr;
    uint8_t *d_out_refseq = nullptr;
    int64_t n_windows = -1;
    std::string error;

    SlabView view(const Slab &s) const {
        return SlabView{s.reads, s.ops, s.op_elem, s.seq, s.tuples, (uint32_t)s.n_ops, (uint32_t)s.n_elem};
    }
};

namespace {

int fe_fail(clair_frontend *f, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (f) f->error = buf; else g_fe_error = buf;
    return 1;
}

#define FE_TRY(f, call)                                                                                   \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess)                                                                          \
            return fe_fail((f), "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
    } while (0)

inline unsigned blocks_for(int64_t n, int64_t per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// flags[n] -> how many are set (block sums left scanned in block_sum for scan_write)
int scan_count(clair_frontend *f, const uint8_t *flags, int64_t n, uint32_t *block_sum, uint32_t *d_count, int64_t *count) {
    const unsigned nb = blocks_for(n, SCAN_BLOCK);
    if (n > 0) hipLaunchKernelGGL(fe_block_count_kernel, dim3(nb), dim3(256), 0, f->stream, flags, n, block_sum);
    hipLaunchKernelGGL(fe_scan_block_sums_kernel, dim3(1), dim3(256), 0, f->stream, block_sum, (int64_t)nb, d_count);
    FE_TRY(f, hipGetLastError());
    uint32_t total = 0;
    FE_TRY(f, hipMemcpyAsync(&total, d_count, sizeof total, hipMemcpyDeviceToHost, f->stream));
    FE_TRY(f, hipStreamSynchronize(f->stream));
    *count = total;
    return 0;
}

// after scan_count: prefix[n + 1] (optional) and the list of flagged indices (+ list_base)
int scan_write(clair_frontend *f, const uint8_t *flags, int64_t n, uint32_t *block_sum, const uint32_t *d_count, uint32_t *prefix, int64_t *list,
               int64_t list_base) {
    // entry n of the prefix is written by the thread that owns index n: cover it with one more (empty) item
    const unsigned nb = blocks_for(n, SCAN_BLOCK), nbw = blocks_for(n + 1, SCAN_BLOCK);
    if (nbw > nb) FE_TRY(f, hipMemcpyAsync(block_sum + nb, d_count, sizeof(uint32_t), hipMemcpyDeviceToDevice, f->stream));
    hipLaunchKernelGGL(fe_scan_write_kernel, dim3(nbw), dim3(256), 0, f->stream, flags, n, (const uint32_t *)block_sum, prefix, list, list_base);
    FE_TRY(f, hipGetLastError());
    return 0;
}

void free_candidates(clair_frontend *f) {
    (void)hipFree(f->d_centre); f->d_centre = nullptr;
    (void)hipFree(f->d_ins); f->d_ins = nullptr;
    (void)hipFree(f->d_keep); f->d_keep = nullptr;
    (void)hipFree(f->d_window_tuples); f->d_window_tuples = nullptr;
    (void)hipFree(f->d_kept); f->d_kept = nullptr;
    (void)hipFree(f->d_cand_block_sum); f->d_cand_block_sum = nullptr;
    (void)hipFree(f->d_counts); f->d_counts = nullptr;
    (void)hipFree(f->d_out_centre); f->d_out_centre = nullptr;
    (void)hipFree(f->d_out_refseq); f->d_out_refseq = nullptr;
    f->n_candidates = -1;
    f->n_windows = -1;
}

// after the flags are set: the candidate list, its per-position prefix, and room for the windows
int finish_candidates(clair_frontend *f, int64_t *n_candidates) {
    free_candidates(f);
    int64_t n = 0;
    if (scan_count(f, f->d_flags, f->g.n, f->d_block_sum, f->d_total, &n)) return 1;
    FE_TRY(f, hipMalloc((void **)&f->d_centre, (size_t)std::max<int64_t>(n, 1) * sizeof(int64_t)));
    if (scan_write(f, f->d_flags, f->g.n, f->d_block_sum, f->d_total, f->d_before, f->d_centre, f->g.lo + 1)) return 1;
    f->n_candidates = n;
    *n_candidates = n;
    return 0;
}

}  // namespace

extern "C" {

const char *clair_frontend_last_error(const clair_frontend_t *f) { return f ? f->error.c_str() : g_fe_error.c_str(); }

int clair_frontend_create(int device, const char *ref_seq, int64_t ref_len, int64_t reference_start_0_based, int64_t span_lo, int64_t span_hi,
                          clair_frontend_t **out) {
    if (!out) return fe_fail(nullptr, "out is NULL");
    *out = nullptr;
    if (!ref_seq || ref_len < 0) return fe_fail(nullptr, "reference sequence missing");
    if (span_hi <= span_lo) return fe_fail(nullptr, "empty span [%lld, %lld)", (long long)span_lo, (long long)span_hi);
    if (span_hi - span_lo > ((int64_t)1 << 32) - 8192) return fe_fail(nullptr, "span of %lld positions is beyond one front end (2^32)", (long long)(span_hi - span_lo));
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
        return fe_fail(nullptr, "no HIP device is visible: the front end runs on an MI355X only (the host code is clair_host_evc_* / clair_host_pileup_*)");
    if (device < 0 || device >= n_dev) return fe_fail(nullptr, "device %d out of range [0,%d)", device, n_dev);
    clair_frontend *f = new clair_frontend;
    f->device = device;
    auto bail = [&](const char *what, hipError_t err) {
        fe_fail(nullptr, "%s failed: %s", what, hipGetErrorString(err));
        clair_frontend_destroy(f);
        return 1;
    };
    hipError_t err;
    if ((err = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", err);
    if ((err = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", err);
    Region &g = f->g;
    g.lo = span_lo;
    g.n = span_hi - span_lo;
    g.ref0 = reference_start_0_based;
    g.ref_len = ref_len;
    if ((err = hipMalloc((void **)&f->d_ref, (size_t)std::max<int64_t>(ref_len, 1))) != hipSuccess) return bail("hipMalloc(reference)", err);
    if ((err = hipMemcpy(f->d_ref, ref_seq, (size_t)ref_len, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(reference)", err);
    g.ref = f->d_ref;
    const size_t table = (size_t)g.n * 8 * sizeof(uint32_t);
    if ((err = hipMalloc((void **)&g.ev, table)) != hipSuccess) return bail("hipMalloc(tallies)", err);
    if ((err = hipMalloc((void **)&g.q, table)) != hipSuccess) return bail("hipMalloc(read-base rows)", err);
    if ((err = hipMalloc((void **)&g.misc, table)) != hipSuccess) return bail("hipMalloc(counts)", err);
    if ((err = hipMalloc((void **)&g.anomalies, sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void **)&f->d_flags, (size_t)g.n + 1)) != hipSuccess) return bail("hipMalloc(flags)", err);
    if ((err = hipMalloc((void **)&f->d_before, ((size_t)g.n + 1) * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(prefix)", err);
    if ((err = hipMalloc((void **)&f->d_block_sum, ((size_t)blocks_for(g.n + 1, SCAN_BLOCK) + 1) * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void **)&f->d_total, 2 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", err);
    (void)hipMemsetAsync(g.ev, 0, table, f->stream);
    (void)hipMemsetAsync(g.q, 0, table, f->stream);
    (void)hipMemsetAsync(g.misc, 0, table, f->stream);
    (void)hipMemsetAsync(g.anomalies, 0, sizeof(uint32_t), f->stream);
    if ((err = hipStreamSynchronize(f->stream)) != hipSuccess) return bail("hipMemset", err);
    *out = f;
    return 0;
}

void clair_frontend_destroy(clair_frontend_t *f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    for (Slab &s : f->slabs) {
        (void)hipFree(s.reads); (void)hipFree(s.ops); (void)hipFree(s.op_elem); (void)hipFree(s.seq); (void)hipFree(s.tuples);
    }
    free_candidates(f);
    (void)hipFree(f->d_ref); (void)hipFree(f->g.ev); (void)hipFree(f->g.q); (void)hipFree(f->g.misc); (void)hipFree(f->g.anomalies);
    (void)hipFree(f->d_flags); (void)hipFree(f->d_before); (void)hipFree(f->d_block_sum); (void)hipFree(f->d_total); (void)hipFree(f->d_bed);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

int clair_frontend_add_reads(clair_frontend_t *f, const clair_read_t *reads, int64_t n_reads, const clair_op_t *ops, int64_t n_ops,
                             const uint32_t *op_elem, const uint8_t *seq, int64_t seq_bytes) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (n_reads < 0 || n_ops < 0 || seq_bytes < 0) return fe_fail(f, "negative slab size");
    if (n_reads == 0 || n_ops == 0) return 0;
    if (!reads || !ops || !op_elem || (!seq && seq_bytes > 0)) return fe_fail(f, "NULL slab array");
    if (n_ops > 0xfffffff0ll || seq_bytes > 0xfffffff0ll) return fe_fail(f, "slab too large for 32-bit offsets");
    if (f->n_candidates >= 0) return fe_fail(f, "reads cannot be added after the candidates were fixed");
    FE_TRY(f, hipSetDevice(f->device));
    Slab s;
    s.n_reads = n_reads; s.n_ops = n_ops; s.n_elem = op_elem[n_ops]; s.seq_bytes = seq_bytes;
    FE_TRY(f, hipMalloc((void **)&s.reads, (size_t)n_reads * sizeof(clair_read_t)));
    f->slabs.push_back(s);                    // owned from here on: destroy frees whatever was allocated
    Slab &d = f->slabs.back();
    FE_TRY(f, hipMalloc((void **)&d.ops, (size_t)n_ops * sizeof(clair_op_t)));
    FE_TRY(f, hipMalloc((void **)&d.op_elem, ((size_t)n_ops + 1) * sizeof(uint32_t)));
    FE_TRY(f, hipMalloc((void **)&d.seq, (size_t)std::max<int64_t>(seq_bytes, 1)));
    FE_TRY(f, hipMalloc((void **)&d.tuples, (size_t)n_reads * sizeof(uint64_t)));
    FE_TRY(f, hipMemcpyAsync(d.reads, reads, (size_t)n_reads * sizeof(clair_read_t), hipMemcpyHostToDevice, f->stream));
    FE_TRY(f, hipMemcpyAsync(d.ops, ops, (size_t)n_ops * sizeof(clair_op_t), hipMemcpyHostToDevice, f->stream));
    FE_TRY(f, hipMemcpyAsync(d.op_elem, op_elem, ((size_t)n_ops + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, f->stream));
    if (seq_bytes) FE_TRY(f, hipMemcpyAsync(d.seq, seq, (size_t)seq_bytes, hipMemcpyHostToDevice, f->stream));
    FE_TRY(f, hipMemsetAsync(d.tuples, 0, (size_t)n_reads * sizeof(uint64_t), f->stream));
    if (d.n_elem) hipLaunchKernelGGL(fe_tally_kernel, dim3(blocks_for(d.n_elem, 256)), dim3(256), 0, f->stream, f->g, f->view(d));
    FE_TRY(f, hipGetLastError());
    // the caller's arrays may be reused as soon as this returns (the packer's slab is reset): wait for the copies
    FE_TRY(f, hipStreamSynchronize(f->stream));
    return 0;
}

int clair_frontend_find_candidates(clair_frontend_t *f, double min_coverage, double threshold, int64_t ctg_start, int64_t ctg_end,
                                   const int64_t *bed_start, const int64_t *bed_end, int64_t n_bed, int64_t *n_candidates) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (!n_candidates) return fe_fail(f, "n_candidates is NULL");
    if (n_bed > 0 && (!bed_start || !bed_end)) return fe_fail(f, "bed intervals missing");
    FE_TRY(f, hipSetDevice(f->device));
    CandidateRule rule{min_coverage, threshold, ctg_start >= 0 && ctg_end >= 0 ? ctg_start : -1, ctg_end, nullptr, nullptr, n_bed < 0 ? -1 : 0};
    if (n_bed > 0) {   // membership only (shared/interval_tree.py:30-32, 45-57): sort, widen empty intervals by one, merge
        std::vector<std::pair<int64_t, int64_t>> iv;
        for (int64_t i = 0; i < n_bed; ++i) iv.emplace_back(bed_start[i], bed_end[i] == bed_start[i] ? bed_end[i] + 1 : bed_end[i]);
        std::sort(iv.begin(), iv.end());
        std::vector<int64_t> st, en;
        for (auto &x : iv) {
            if (x.second <= x.first) continue;
            if (!st.empty() && x.first <= en.back()) en.back() = std::max(en.back(), x.second);
            else { st.push_back(x.first); en.push_back(x.second); }
        }
        (void)hipFree(f->d_bed); f->d_bed = nullptr;
        const size_t m = st.size();
        FE_TRY(f, hipMalloc((void **)&f->d_bed, std::max<size_t>(2 * m, 1) * sizeof(int64_t)));
        if (m) {
            FE_TRY(f, hipMemcpy(f->d_bed, st.data(), m * sizeof(int64_t), hipMemcpyHostToDevice));
            FE_TRY(f, hipMemcpy(f->d_bed + m, en.data(), m * sizeof(int64_t), hipMemcpyHostToDevice));
        }
        rule.bed_start = f->d_bed;
        rule.bed_end = f->d_bed + m;
        rule.n_bed = (int64_t)m;
    }
    hipLaunchKernelGGL(fe_candidate_flags_kernel, dim3(blocks_for(f->g.n, 256)), dim3(256), 0, f->stream, f->g, rule, f->d_flags);
    FE_TRY(f, hipGetLastError());
    return finish_candidates(f, n_candidates);
}

int clair_frontend_set_candidates(clair_frontend_t *f, const int64_t *positions, int64_t n_positions, int64_t *n_candidates) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (!n_candidates || (n_positions > 0 && !positions) || n_positions < 0) return fe_fail(f, "bad candidate list");
    FE_TRY(f, hipSetDevice(f->device));
    for (int64_t i = 1; i < n_positions; ++i)
        if (positions[i] <= positions[i - 1]) {     // the sequential code keeps list order (CreateTensor.py:86-109); this path needs it ascending
            uint32_t bit = CLAIR_FE_CANDIDATES, cur = 0;
            FE_TRY(f, hipMemcpy(&cur, f->g.anomalies, sizeof cur, hipMemcpyDeviceToHost));
            cur |= bit;
            FE_TRY(f, hipMemcpy(f->g.anomalies, &cur, sizeof cur, hipMemcpyHostToDevice));
            break;
        }
    FE_TRY(f, hipMemsetAsync(f->d_flags, 0, (size_t)f->g.n + 1, f->stream));
    if (n_positions) {
        int64_t *d_pos = nullptr;
        FE_TRY(f, hipMalloc((void **)&d_pos, (size_t)n_positions * sizeof(int64_t)));
        hipError_t err = hipMemcpyAsync(d_pos, positions, (size_t)n_positions * sizeof(int64_t), hipMemcpyHostToDevice, f->stream);
        if (err == hipSuccess) {
            hipLaunchKernelGGL(fe_given_flags_kernel, dim3(blocks_for(n_positions, 256)), dim3(256), 0, f->stream, (const int64_t *)d_pos, n_positions, f->g.lo, f->g.n, f->d_flags);
            err = hipStreamSynchronize(f->stream);
        }
        (void)hipFree(d_pos);
        FE_TRY(f, err);
    }
    return finish_candidates(f, n_candidates);
}

int clair_frontend_get_candidates(clair_frontend_t *f, int64_t *positions) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (f->n_candidates < 0) return fe_fail(f, "no candidates yet: call clair_frontend_find_candidates / _set_candidates first");
    if (f->n_candidates && !positions) return fe_fail(f, "positions is NULL");
    FE_TRY(f, hipSetDevice(f->device));
    if (f->n_candidates) FE_TRY(f, hipMemcpy(positions, f->d_centre, (size_t)f->n_candidates * sizeof(int64_t), hipMemcpyDeviceToHost));
    return 0;
}

int clair_frontend_build_windows(clair_frontend_t *f, int min_coverage, int drop_non_iupac_centre, int64_t *n_windows) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (!n_windows) return fe_fail(f, "n_windows is NULL");
    if (f->n_candidates < 0) return fe_fail(f, "no candidates yet: call clair_frontend_find_candidates / _set_candidates first");
    FE_TRY(f, hipSetDevice(f->device));
    const int64_t nc = f->n_candidates, room = std::max<int64_t>(nc, 1);
    if (!f->d_ins) {
        FE_TRY(f, hipMalloc((void **)&f->d_ins, (size_t)room * N_QUAD * sizeof(uint32_t)));
        FE_TRY(f, hipMalloc((void **)&f->d_keep, (size_t)room + 1));
        FE_TRY(f, hipMalloc((void **)&f->d_window_tuples, (size_t)room * sizeof(uint64_t)));
        FE_TRY(f, hipMalloc((void **)&f->d_kept, (size_t)room * sizeof(int64_t)));
        FE_TRY(f, hipMalloc((void **)&f->d_cand_block_sum, ((size_t)blocks_for(room + 1, SCAN_BLOCK) + 1) * sizeof(uint32_t)));
    }
    FE_TRY(f, hipMemsetAsync(f->d_ins, 0, (size_t)room * N_QUAD * sizeof(uint32_t), f->stream));
    Candidates c{f->d_centre, f->d_before, f->d_ins};
    for (Slab &s : f->slabs) {
        FE_TRY(f, hipMemsetAsync(s.tuples, 0, (size_t)s.n_reads * sizeof(uint64_t), f->stream));
        if (s.n_elem && nc) hipLaunchKernelGGL(fe_windows_per_base_kernel, dim3(blocks_for(s.n_elem, 256)), dim3(256), 0, f->stream, f->g, f->view(s), c);
    }
    WindowRule rule{min_coverage, drop_non_iupac_centre};
    if (nc) hipLaunchKernelGGL(fe_window_flags_kernel, dim3(blocks_for(nc, 256)), dim3(256), 0, f->stream, f->g, (const int64_t *)f->d_centre, nc, rule, f->d_keep, f->d_window_tuples);
    FE_TRY(f, hipGetLastError());
    int64_t kept = 0;
    if (scan_count(f, f->d_keep, nc, f->d_cand_block_sum, f->d_total + 1, &kept)) return 1;
    if (scan_write(f, f->d_keep, nc, f->d_cand_block_sum, f->d_total + 1, nullptr, f->d_kept, 0)) return 1;
    (void)hipFree(f->d_counts); f->d_counts = nullptr;
    (void)hipFree(f->d_out_centre); f->d_out_centre = nullptr;
    (void)hipFree(f->d_out_refseq); f->d_out_refseq = nullptr;
    // + one engine batch of slack is the caller's business: clair_submit_ex reads exactly n windows
    FE_TRY(f, hipMalloc((void **)&f->d_counts, (size_t)std::max<int64_t>(kept, 1) * WINDOW_VALUES * sizeof(int16_t)));
    FE_TRY(f, hipMalloc((void **)&f->d_out_centre, (size_t)std::max<int64_t>(kept, 1) * sizeof(int64_t)));
    FE_TRY(f, hipMalloc((void **)&f->d_out_refseq, (size_t)std::max<int64_t>(kept, 1) * 34));
    if (kept)
        hipLaunchKernelGGL(fe_assemble_kernel, dim3(blocks_for(kept * N_QUAD, 256)), dim3(256), 0, f->stream, f->g, (const int64_t *)f->d_centre,
                           (const int64_t *)f->d_kept, kept, (const uint32_t *)f->d_ins, f->d_counts, f->d_out_centre, f->d_out_refseq);
    FE_TRY(f, hipGetLastError());
    FE_TRY(f, hipStreamSynchronize(f->stream));
    f->n_windows = kept;
    *n_windows = kept;
    return 0;
}

int clair_frontend_window_info(clair_frontend_t *f, int64_t first, int64_t n, int64_t *centres, char *refseq) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (f->n_windows < 0) return fe_fail(f, "no windows yet: call clair_frontend_build_windows first");
    if (first < 0 || n < 0 || first + n > f->n_windows) return fe_fail(f, "windows [%lld, %lld) out of range [0, %lld)", (long long)first, (long long)(first + n), (long long)f->n_windows);
    if (n == 0) return 0;
    if (!centres || !refseq) return fe_fail(f, "NULL output pointer");
    FE_TRY(f, hipSetDevice(f->device));
    FE_TRY(f, hipMemcpy(centres, f->d_out_centre + first, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    FE_TRY(f, hipMemcpy(refseq, f->d_out_refseq + first * 34, (size_t)n * 34, hipMemcpyDeviceToHost));
    return 0;
}

int clair_frontend_window_counts(clair_frontend_t *f, int64_t first, int64_t n, int16_t *counts) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (f->n_windows < 0) return fe_fail(f, "no windows yet: call clair_frontend_build_windows first");
    if (first < 0 || n < 0 || first + n > f->n_windows) return fe_fail(f, "windows [%lld, %lld) out of range [0, %lld)", (long long)first, (long long)(first + n), (long long)f->n_windows);
    if (n == 0) return 0;
    if (!counts) return fe_fail(f, "NULL output pointer");
    FE_TRY(f, hipSetDevice(f->device));
    FE_TRY(f, hipMemcpy(counts, (const int16_t *)f->d_counts + first * WINDOW_VALUES, (size_t)n * WINDOW_VALUES * sizeof(int16_t), hipMemcpyDeviceToHost));
    return 0;
}

const int16_t *clair_frontend_counts_device(clair_frontend_t *f, int64_t first) {
    if (!f || f->n_windows < 0 || first < 0 || first > f->n_windows) return nullptr;
    return (const int16_t *)f->d_counts + first * WINDOW_VALUES;
}

int clair_frontend_budget_inputs(clair_frontend_t *f, int64_t slab, uint64_t *read_tuples, int64_t *centres, uint64_t *window_tuples) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (f->n_windows < 0) return fe_fail(f, "no windows yet: call clair_frontend_build_windows first");
    FE_TRY(f, hipSetDevice(f->device));
    if (read_tuples) {
        if (slab < 0 || slab >= (int64_t)f->slabs.size()) return fe_fail(f, "slab %lld out of range [0, %lld)", (long long)slab, (long long)f->slabs.size());
        const Slab &s = f->slabs[(size_t)slab];
        FE_TRY(f, hipMemcpy(read_tuples, s.tuples, (size_t)s.n_reads * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    if (centres && f->n_candidates) FE_TRY(f, hipMemcpy(centres, f->d_centre, (size_t)f->n_candidates * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (window_tuples && f->n_candidates) FE_TRY(f, hipMemcpy(window_tuples, f->d_window_tuples, (size_t)f->n_candidates * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int clair_frontend_stats(clair_frontend_t *f, int64_t *stats) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (!stats) return fe_fail(f, "stats is NULL");
    FE_TRY(f, hipSetDevice(f->device));
    uint32_t bits = 0;
    FE_TRY(f, hipStreamSynchronize(f->stream));
    FE_TRY(f, hipMemcpy(&bits, f->g.anomalies, sizeof bits, hipMemcpyDeviceToHost));
    int64_t reads = 0, elems = 0;
    for (const Slab &s : f->slabs) { reads += s.n_reads; elems += s.n_elem; }
    stats[0] = (int64_t)bits;
    stats[1] = (int64_t)f->slabs.size();
    stats[2] = reads;
    stats[3] = elems;
    stats[4] = f->n_candidates;
    stats[5] = f->n_windows;
    return 0;
}

}  // extern "C"
