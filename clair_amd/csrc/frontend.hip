// Device front end of the C ABI (include/clair_amd.h, "front end"): `samtools view` text (or alignments already packed by the host,
// include/clair_reads.h) -> candidate sites -> pileup count windows [n][33][8][4] int16 that stay in HBM for clair_submit_ex.
// SURVEY.md 8(f) N4, the GPU half.
//
// What the reference does per read base in two interpreters (dataPrepScripts/ExtractVariantCandidates.py:296-316: a dict of
// per-position tallies; CreateTensor.py:289-365 + :29-65: a tuple per (read base, open window), summed when the window is written)
// is done here per reference POSITION: one thread per read base adds to per-position tables with atomics (pass 1), the candidate
// filter runs over the tallies, one thread per read base counts the windows it lies in and scatters the inserted bases (pass 2), and
// the windows are assembled from 33 table columns each.  Why that is the same result -- and when it is not (a tuple budget that
// binds, unsorted input, bases outside the IUPAC alphabet: reported as CLAIR_FE_* bits, the caller then runs the sequential host
// code) -- is in oracle/frontend_np.py, the NumPy restatement these kernels are tested against, and in LABNOTES.md part B section 3 / 6b.
//
// All integer work, HBM/atomic bound: nothing here is a matrix product.  Tables are sized for the whole region and the packed reads
// stay resident (a 10 Mb region at 50x: ~0.7 GB of tables, 1.3 - 2.1 GB of reads; a whole chromosome fits the 288 GB many times over), so
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
        // both scripts upper-case SEQ first (EVC :301, CT :262): a slab parsed on the device keeps the text as samtools printed it
        for (int c = 'a'; c <= 'z'; ++c) { pile[c] = pile[c - 32]; evc[c] = evc[c - 32]; }
    }
};
__constant__ BaseTables BASES = BaseTables();

struct Region {          // what every kernel needs to know about the tables and the reference
    int64_t lo, n;       // tables cover 0-based positions [lo, lo + n)
    const uint8_t *ref;  // upper-cased reference bytes for [ref0, ref0 + ref_len)
    int64_t ref0, ref_len;
    // Per position and base row b (A C G T, IUPAC codes mapped as the scripts map them) ONE 64-bit word of three 21-bit counters, so that
    // a matched base costs one atomic: bits 0-20 pileup reads on the forward strand with read base b, 21-41 the same on the reverse
    // strand, 42-62 the candidate search's tally of b.  (Matched bases per strand = the sum over the four rows.)
    unsigned long long *pq;   // [n][4]
    uint32_t *misc;      // [n][8]  0,1 D per strand (rp > POS) | 2 D at rp == POS | 3 inserted bases (rp > POS) | 4 M at rp == POS | 5,6,7 the search's I, D, N tallies
    uint32_t *anomalies;
};

constexpr int PQ_BITS = 21;
constexpr unsigned long long PQ_MASK = (1ull << PQ_BITS) - 1;
__device__ inline uint32_t pq_field(unsigned long long w, int field) { return (uint32_t)((w >> (field * PQ_BITS)) & PQ_MASK); }

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
        // the search's base index and the pileup's row agree for every code but N (tally N, row of A)
        const int row = BASES.pile[base];
        unsigned long long add = pile ? 1ull << (so * PQ_BITS) : 0;
        if (evc && ei < 4) add += 1ull << (2 * PQ_BITS);
        if (add) {      // (the reply costs nothing measurable: the atomic unit at the memory side is the bound either way)
            const unsigned long long before = atomicAdd(&g.pq[t * 4 + row], add);
            if ((pile && pq_field(before, so) == PQ_MASK) || (evc && ei < 4 && pq_field(before, 2) == PQ_MASK))
                flag(g, CLAIR_FE_OVERFLOW);                     // a counter was full: 2 097 151 reads over one position
        }
        if (evc && ei >= 4) atomicAdd(&g.misc[t * 8 + 7], 1u);
        if (pile && el.rp == el.r.pos0) atomicAdd(&g.misc[t * 8 + 4], 1u);
    } else if (el.code == CLAIR_OP_I) {
        if (el.k == 0 && evc && t - 1 >= 0 && t - 1 < g.n) atomicAdd(&g.misc[(t - 1) * 8 + 5], 1u);   // once per operation, at the base before it
        if (!pile) return;
        if (el.qp >= el.r.seq_len) { flag(g, CLAIR_FE_SEQ_OVERRUN); return; }
        if (el.rp <= el.r.pos0) return;                                                              // no window is open yet (CreateTensor.py:326-341)
        if (BASES.pile[s.seq[el.r.seq0 + el.qp]] == 255) flag(g, CLAIR_FE_BAD_BASE);
        if (inside) atomicAdd(&g.misc[t * 8 + 3], 1u);
    } else {
        if (el.k == 0 && evc && t - 1 >= 0 && t - 1 < g.n) atomicAdd(&g.misc[(t - 1) * 8 + 6], 1u);
        if (!pile || !inside) return;
        if (ref_row(g, el.rp) == 255) flag(g, CLAIR_FE_BAD_REF);
        if (el.rp > el.r.pos0) atomicAdd(&g.misc[t * 8 + so], 1u);
        else atomicAdd(&g.misc[t * 8 + 2], 1u);
    }
}

// ---- pass 1 without a device-scope atomic per base (round 4) --------------------------------------------------
// fe_tally_kernel above sends one 32-byte atomic request per aligned base to the memory side of the L2 (3.3 GB per 102 M bases, ten times
// the text in plus tables out: the atomic unit is its bound).  Alignments come sorted by position, so the ~depth bases that land on one
// position are found together if the work is cut by POSITION instead of by base: a workgroup owns a tile of TT_POS table positions,
// keeps the tile's 64 bytes of counters per position in LDS, walks the part of every alignment that crosses the tile -- the alignments
// with POS in (tile start - longest span of the slab, tile end), found by bisection; each one's first operation in the tile by bisection
// in its operation list, all alignments of the tile at once, a thread each -- and adds its tile to the tables once: one atomic per
// non-zero counter word and tile (the neighbour's I / D tally at "the position before" may land in this tile's words at the same time).
// The tile of the first / last position also owns everything left / right of the tables: those bases tally nothing but are still
// examined (sequence overrun, bases outside the alphabet), as the per-base kernel does.
constexpr int TT_POS = 512;              // table positions per tile: 16 KB + 16 KB of counters
constexpr int TT_THREADS = 1024;
constexpr int TT_WAVES = TT_THREADS / 64;
constexpr int TT_LIST = TT_THREADS;      // alignments examined per round, one thread each

__global__ __launch_bounds__(256) void fe_slab_span_kernel(SlabView s, uint32_t n_reads, uint32_t *span) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_reads) return;
    const clair_read_t r = s.reads[i];
    if (!r.n_ops) return;
    const clair_op_t last = s.ops[r.op0 + r.n_ops - 1];
    const uint32_t len = last.code_len >> 2, code = last.code_len & 3u;
    const int64_t reach = (int64_t)last.ref_off + (code == CLAIR_OP_I ? 1 : (int64_t)len);      // positions pos0 .. pos0 + reach - 1 are touched
    atomicMax(span, (uint32_t)std::min<int64_t>(std::max<int64_t>(reach, 0), 0x7fffffff));
}

struct TileOp { int32_t rel; uint32_t code_len, q_off, first, k0; };    // an operation as the tile sees it: start relative to the tile, its first flat element, first k inside

__global__ __launch_bounds__(TT_THREADS) void fe_tally_tile_kernel(Region g, SlabView s, uint32_t n_reads, const uint32_t *slab_span, int64_t tile0) {
    __shared__ unsigned long long pq_l[TT_POS * 4];
    __shared__ uint32_t misc_l[TT_POS * 8];
    __shared__ uint2 list[TT_LIST];             // (alignment, its first operation that reaches into the tile)
    __shared__ uint32_t n_list;
    __shared__ TileOp opbuf[TT_WAVES][64];
    __shared__ uint8_t evc_l[256], pile_l[256];  // the base tables and the tile's reference rows next to the counters: what a base needs besides its own byte
    __shared__ uint8_t ref_l[TT_POS];            // is one LDS read away (from constant / global memory each is a dependent trip of its own)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the launch covers table tiles tile0 .. tile0 + gridDim.x - 1: where this slab's alignments lie (a guess is enough: the first and the last
    // workgroup own everything left / right of their tile, and tally what falls inside the tables but outside their LDS window straight into them)
    const int64_t t_lo = (tile0 + (int64_t)blockIdx.x) * TT_POS;
    const bool first = blockIdx.x == 0, last = blockIdx.x == gridDim.x - 1;
    const int64_t p_lo = first ? INT64_MIN / 4 : g.lo + t_lo, p_hi = last ? INT64_MAX / 4 : g.lo + t_lo + TT_POS;     // positions this tile owns
    const int64_t p_base = g.lo + t_lo;
    for (int i = tid; i < TT_POS * 4; i += TT_THREADS) pq_l[i] = 0ull;
    for (int i = tid; i < TT_POS * 8; i += TT_THREADS) misc_l[i] = 0u;
    if (tid < 256) { evc_l[tid] = BASES.evc[tid]; pile_l[tid] = BASES.pile[tid]; }
    else if (tid < 256 + TT_POS) ref_l[tid - 256] = ref_row(g, p_base + (tid - 256));
    // alignments that may reach into the tile: POS < p_hi, and POS + (longest reach of the slab) > p_lo
    const int64_t span = (int64_t)*slab_span;
    uint32_t ra = 0, rb = n_reads;
    if (!first) { uint32_t a = 0, b = n_reads; while (a < b) { const uint32_t m = (a + b) >> 1; if (s.reads[m].pos0 + span > p_lo) b = m; else a = m + 1; } ra = a; }
    if (!last) { uint32_t a = ra, b = n_reads; while (a < b) { const uint32_t m = (a + b) >> 1; if (s.reads[m].pos0 >= p_hi) b = m; else a = m + 1; } rb = a; }
    for (uint32_t base = ra; base < rb; base += TT_LIST) {
        if (tid == 0) n_list = 0u;
        __syncthreads();
        const uint32_t i = base + (uint32_t)tid;
        if (i < rb) {
            const clair_read_t r = s.reads[i];
            if (r.n_ops && r.pos0 < p_hi) {
                // operations are ordered and their ends (start + length; an insertion counts as one position) never decrease: first one ending beyond p_lo
                uint32_t a = 0, b = r.n_ops;
                while (a < b) {
                    const uint32_t m = (a + b) >> 1;
                    const clair_op_t op = s.ops[r.op0 + m];
                    const int64_t end = r.pos0 + op.ref_off + ((op.code_len & 3u) == CLAIR_OP_I ? 1 : (int64_t)(op.code_len >> 2));
                    if (end > p_lo) b = m; else a = m + 1;
                }
                if (a < r.n_ops) list[atomicAdd(&n_list, 1u)] = make_uint2(i, r.op0 + a);
            }
        }
        __syncthreads();
        const uint32_t n = n_list;
        for (uint32_t li = (uint32_t)wave; li < n; li += TT_WAVES) {      // a wave per alignment
            const uint2 ent = list[li];
            const uint32_t ri = __builtin_amdgcn_readfirstlane(ent.x);
            const clair_read_t r = s.reads[ri];
            const bool evc = r.flags & CLAIR_READ_EVC, pile = r.flags & CLAIR_READ_PILE;
            const int so = (r.flags & CLAIR_READ_REVERSE) ? 1 : 0;
            const uint32_t jend = r.op0 + r.n_ops;
            for (uint32_t j = __builtin_amdgcn_readfirstlane(ent.y); j < jend; j += 64) {          // 64 operations at a time, one per lane
                const uint32_t jj = j + (uint32_t)lane;
                const bool valid = jj < jend;
                clair_op_t op{};
                if (valid) op = s.ops[jj];
                const uint32_t code = op.code_len & 3u, len = op.code_len >> 2;
                const int64_t start = r.pos0 + op.ref_off;
                const bool beyond = valid && start >= p_hi;
                const bool in = valid && !beyond && start + (code == CLAIR_OP_I ? 1 : (int64_t)len) > p_lo;
                uint32_t k0 = 0, k1 = 0;
                if (in) {
                    if (code == CLAIR_OP_I) { k1 = len; }
                    else { k0 = (uint32_t)std::max<int64_t>(0, p_lo - start); k1 = (uint32_t)std::min<int64_t>((int64_t)len, p_hi - start); }
                }
                const uint32_t cnt = k1 - k0;
                uint32_t incl = cnt;                       // inclusive prefix sum over the wave
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
                const uint32_t total = __shfl(incl, 63);
                opbuf[wave][lane] = TileOp{(int32_t)(start - p_base), op.code_len, op.q_off, incl - cnt, k0};
                // an element in two steps, so that two of them are in flight per lane: find its operation (LDS) and fetch its read base; tally
                struct Prep { int64_t lt; uint32_t oc, k, qp; uint8_t bs; };
                auto prep = [&](uint32_t e) {
                    int a = 0, b = 64;                     // last operation whose first flat element is <= e (empty ones share their successor's `first`: take the last)
                    while (b - a > 1) { const int m = (a + b) >> 1; if (opbuf[wave][m].first <= e) a = m; else b = m; }
                    const TileOp o = opbuf[wave][a];
                    Prep q;
                    q.oc = o.code_len & 3u;
                    q.k = o.k0 + (e - o.first);
                    q.lt = (int64_t)o.rel + (q.oc == CLAIR_OP_I ? 0 : (int64_t)q.k);   // position relative to the tile (outside [0, TT_POS) only in the first / last workgroup)
                    q.qp = o.q_off + (q.oc == CLAIR_OP_D ? 0u : q.k);
                    q.bs = (q.oc != CLAIR_OP_D && q.qp < r.seq_len) ? s.seq[r.seq0 + q.qp] : (uint8_t)0;
                    return q;
                };
                auto apply = [&](const Prep &q) {
                    const int64_t lt = q.lt, t = t_lo + lt, rp = p_base + lt;
                    const bool inside = t >= 0 && t < g.n;
                    auto row_of = [&](int64_t at, int64_t pos) -> uint8_t { return at >= 0 && at < TT_POS ? ref_l[at] : ref_row(g, pos); };
                    auto bump = [&](int64_t at, int word) {          // misc counter `word` of tile position `at`: the tile's copy, or the table itself
                        if (at >= 0 && at < TT_POS) atomicAdd(&misc_l[at * 8 + word], 1u); else atomicAdd(&g.misc[(t_lo + at) * 8 + word], 1u);
                    };
                    if (q.oc == CLAIR_OP_M) {
                        if (q.qp >= r.seq_len) { flag(g, CLAIR_FE_SEQ_OVERRUN); return; }
                        if (pile && inside && row_of(lt, rp) == 255) flag(g, CLAIR_FE_BAD_REF);
                        const uint8_t ei = evc_l[q.bs];
                        if (ei == 255) { flag(g, CLAIR_FE_BAD_BASE); return; }
                        if (!inside) return;
                        const int row = pile_l[q.bs];
                        unsigned long long add = pile ? 1ull << (so * PQ_BITS) : 0;
                        if (evc && ei < 4) add += 1ull << (2 * PQ_BITS);
                        if (add) {
                            if (lt >= 0 && lt < TT_POS) atomicAdd(&pq_l[lt * 4 + row], add);
                            else {
                                const unsigned long long before = atomicAdd(&g.pq[t * 4 + row], add);
                                if ((pile && pq_field(before, so) == PQ_MASK) || (evc && ei < 4 && pq_field(before, 2) == PQ_MASK)) flag(g, CLAIR_FE_OVERFLOW);
                            }
                        }
                        if (evc && ei >= 4) bump(lt, 7);
                        if (pile && rp == r.pos0) bump(lt, 4);
                    } else if (q.oc == CLAIR_OP_I) {
                        if (q.k == 0 && evc && t - 1 >= 0 && t - 1 < g.n) bump(lt - 1, 5);   // once per operation, at the base before it
                        if (!pile) return;
                        if (q.qp >= r.seq_len) { flag(g, CLAIR_FE_SEQ_OVERRUN); return; }
                        if (rp <= r.pos0) return;                                   // no window is open yet (CreateTensor.py:326-341)
                        if (pile_l[q.bs] == 255) flag(g, CLAIR_FE_BAD_BASE);
                        if (inside) bump(lt, 3);
                    } else {
                        if (q.k == 0 && evc && t - 1 >= 0 && t - 1 < g.n) bump(lt - 1, 6);
                        if (!pile || !inside) return;
                        if (row_of(lt, rp) == 255) flag(g, CLAIR_FE_BAD_REF);
                        bump(lt, rp > r.pos0 ? so : 2);
                    }
                };
                for (uint32_t e = (uint32_t)lane; e < total; e += 128) {
                    const Prep q0 = prep(e);
                    const bool two = e + 64 < total;
                    Prep q1 = q0;
                    if (two) q1 = prep(e + 64);
                    apply(q0);
                    if (two) apply(q1);
                }
                if (__ballot(beyond || !valid)) break;      // the rest of the alignment lies beyond the tile (or there is no rest)
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // the tile into the tables: one atomic per counter word that moved
    for (int i = tid; i < TT_POS * 4; i += TT_THREADS) {
        const unsigned long long add = pq_l[i];
        const int64_t t = t_lo + (i >> 2);
        if (add && t >= 0 && t < g.n) {
            const unsigned long long before = atomicAdd(&g.pq[t * 4 + (i & 3)], add);
            for (int fld = 0; fld < 3; ++fld)
                if (pq_field(before, fld) + pq_field(add, fld) > PQ_MASK) flag(g, CLAIR_FE_OVERFLOW);       // a 21-bit counter ran over: 2 097 151 reads over one position
        }
    }
    for (int i = tid; i < TT_POS * 8; i += TT_THREADS) {
        const uint32_t add = misc_l[i];
        const int64_t t = t_lo + (i >> 3);
        if (add && t >= 0 && t < g.n) atomicAdd(&g.misc[t * 8 + (i & 7)], add);
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
    const ulonglong2 w01 = *(const ulonglong2 *)(g.pq + t * 4), w23 = *(const ulonglong2 *)(g.pq + t * 4 + 2);
    const uint4 hi4 = *(const uint4 *)(g.misc + t * 8 + 4);
    const uint32_t n[7] = {pq_field(w01.x, 2), pq_field(w01.y, 2), pq_field(w23.x, 2), pq_field(w23.y, 2), hi4.y, hi4.z, hi4.w};
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

// NEWLINES: the "flags" are text and a line feed is a set flag
template <bool NEWLINES> __device__ inline uint8_t flag_of(const uint8_t *flags, int64_t i) { return NEWLINES ? (uint8_t)(flags[i] == '\n') : flags[i]; }

template <bool NEWLINES>
__global__ __launch_bounds__(256) void fe_block_count_kernel(const uint8_t *flags, int64_t n, uint32_t *block_sum) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t c = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) if (at + i < n) c += flag_of<NEWLINES>(flags, at + i);
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

template <bool NEWLINES>
__global__ __launch_bounds__(256) void fe_scan_write_kernel(const uint8_t *flags, int64_t n, const uint32_t *block_sum, uint32_t *prefix,
                                                            int64_t *list, int64_t list_base) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint8_t f[SCAN_ITEMS];
    uint32_t c = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) { f[i] = at + i < n ? flag_of<NEWLINES>(flags, at + i) : 0; c += f[i]; }
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
    const unsigned long long *before_sum;   // before_sum[t] = before[0] + ... + before[t]: windows over a RUN of positions in four look-ups
    // --stop_consider_left_edge only (NULL otherwise): what reads that START inside a window added to the tables under it -- per
    // window and column 8 read-base rows, M per strand, D per strand -- and the tuples per window as range additions over the index
    uint32_t *late;              // [n_candidates][33][12]
    int *tuple_diff;             // [n_candidates + 1]  (per-base pass 2, CLAIR_AMD_FE_PASS2=base: range additions over the candidate index)
    // per-operation pass 2: the same totals as SECOND differences over the centre value u (index u - lo, n + 2 entries; 64-bit two's complement):
    // tuples(u) = sum over p <= u of (u - p + 1) d2[p]; what falls left of the tables goes into fold[0] = sum v, fold[1] = sum v p
    unsigned long long *d2, *fold;
};
constexpr int LATE_ROW = 12;

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
            const bool left_edge = c.late == nullptr;
            const int so = (el.r.flags & CLAIR_READ_REVERSE) ? 4 : 0;
            int64_t a = -1, b = -2;
            if (el.code == CLAIR_OP_M) {
                a = el.rp > el.r.pos0 ? el.rp - 17 : el.r.pos0 - 16;
                b = el.rp + 17;
            } else if (el.rp > el.r.pos0) {
                a = el.rp - 17;
                b = el.rp + 16;
            }
            // without left-edge windows a read opens only the windows whose first column it walks: centres from POS + 17 on
            if (!left_edge && a < el.r.pos0 + 17) a = el.r.pos0 + 17;
            if (b >= a) {
                const uint32_t i0 = before_at(g, c, a - 1 - g.lo), i1 = before_at(g, c, b - g.lo);
                nc = i1 - i0;
                if (!left_edge && nc) { atomicAdd(&c.tuple_diff[i0], 1); atomicSub(&c.tuple_diff[i1], 1); }
            }
            read = el.read;
            if (!left_edge && el.rp - el.r.pos0 <= 31 && (el.code == CLAIR_OP_M || (el.code == CLAIR_OP_D && el.rp > el.r.pos0))) {
                // ... and what this base put into the tables belongs to none of the windows it starts inside: centres rp - 15 .. POS + 16
                const uint8_t row = el.code == CLAIR_OP_M && el.qp < el.r.seq_len ? BASES.pile[s.seq[el.r.seq0 + el.qp]] : (uint8_t)255;
                if (el.code == CLAIR_OP_D || row != 255) {
                    const uint32_t j0 = before_at(g, c, el.rp - 15 - 1 - g.lo), j1 = before_at(g, c, el.r.pos0 + 16 - g.lo);
                    for (uint32_t j = j0; j < j1; ++j) {
                        uint32_t *cell = c.late + ((size_t)j * N_POS + (el.rp - c.centre[j] + 17)) * LATE_ROW;
                        if (el.code == CLAIR_OP_M) { atomicAdd(cell + row + so, 1u); atomicAdd(cell + 8 + (so >> 2), 1u); }
                        else atomicAdd(cell + 10 + (so >> 2), 1u);
                    }
                }
            }
            if (el.code == CLAIR_OP_I && el.rp > el.r.pos0 && el.qp < el.r.seq_len) {
                const uint8_t row = BASES.pile[s.seq[el.r.seq0 + el.qp]];
                if (row != 255) {
                    int64_t first = el.rp - 15;
                    if (!left_edge && first < el.r.pos0 + 17) first = el.r.pos0 + 17;
                    const uint32_t i0 = before_at(g, c, first - 1 - g.lo), i1 = before_at(g, c, el.rp + 16 - g.lo);
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

// ---- pass 2 with left-edge windows, per OPERATION: the windows open over a run of positions is a difference of sums of the prefix ----
// F(x) = candidates with centre <= x (the prefix, clamped to the span); G(x) = F(lo) + ... + F(x).  A base at rp lies in F(rp + 17) -
// F(rp - 18) open windows (rp + 16 for a deleted or inserted base), so an operation over rp = A..B lies in
// [G(B + 17) - G(A + 16)] - [G(B - 18) - G(A - 19)]: no walk over its bases.  Only a read's very first matched base differs (its
// windows start one centre later, CreateTensor.py:296-305), and the inserted bases still go to their windows one by one.
__device__ inline unsigned long long prefix_sum_at(const Region &g, const Candidates &c, int64_t x) {
    const int64_t i = x - g.lo;
    if (i < 0) return 0;
    if (i <= g.n) return c.before_sum[i];
    return c.before_sum[g.n] + (unsigned long long)(i - g.n) * c.before[g.n];
}

__global__ __launch_bounds__(256) void fe_windows_per_op_kernel(Region g, SlabView s, Candidates c) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    uint32_t read = 0xffffffffu;
    unsigned long long nc = 0;
    if (j < s.n_ops) {
        const clair_op_t op = s.ops[j];
        const clair_read_t r = s.reads[op.read];
        if (r.flags & CLAIR_READ_PILE) {
            read = op.read;
            const uint32_t code = op.code_len & 3u;
            const int64_t len = op.code_len >> 2, pos0 = r.pos0;
            int64_t a = pos0 + op.ref_off, b = a + len - 1;          // first and last reference position of the run (I: both the same)
            if (code == CLAIR_OP_M) {
                if (a == pos0) {                                       // the read's first base: windows of centres POS - 16 .. POS + 17
                    nc += before_at(g, c, pos0 + 17 - g.lo) - before_at(g, c, pos0 - 17 - g.lo);
                    ++a;
                }
                if (b >= a) nc += (prefix_sum_at(g, c, b + 17) - prefix_sum_at(g, c, a + 16)) - (prefix_sum_at(g, c, b - 18) - prefix_sum_at(g, c, a - 19));
            } else if (code == CLAIR_OP_D) {
                if (a == pos0) ++a;                                    // offered before any window is open
                if (b >= a) nc += (prefix_sum_at(g, c, b + 16) - prefix_sum_at(g, c, a + 15)) - (prefix_sum_at(g, c, b - 18) - prefix_sum_at(g, c, a - 19));
            } else if (a > pos0) {
                nc += (unsigned long long)len * (before_at(g, c, a + 16 - g.lo) - before_at(g, c, a - 18 - g.lo));
                const int so = (r.flags & CLAIR_READ_REVERSE) ? 4 : 0;
                const uint32_t i0 = before_at(g, c, a - 15 - 1 - g.lo), i1 = before_at(g, c, a + 16 - g.lo);
                for (int64_t k = 0; k < len && i1 > i0; ++k) {         // generate_tensor :51-53: column min(idx + k, 32), channel 1
                    if (op.q_off + k >= r.seq_len) break;
                    const uint8_t row = BASES.pile[s.seq[r.seq0 + op.q_off + k]];
                    if (row == 255) continue;
                    for (uint32_t i = i0; i < i1; ++i) {
                        const int64_t col = a - c.centre[i] + 17 + k;
                        atomicAdd(&c.ins[((size_t)i * N_POS + (col < N_POS - 1 ? col : N_POS - 1)) * N_ROW + row + so], 1u);
                    }
                }
            }
        }
    }
    unsigned long long todo = __ballot(nc != 0);                       // one atomic per run of equal reads in the wave
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

// ---- pass 2 WITHOUT left-edge windows, per operation (round 4; per base: fe_windows_per_base_kernel above, kept as the checker) ----
// A base at rp of an alignment that starts at POS lies in the windows of centres u in [max(rp - 17, POS + 17), rp + 17] (rp + 16 for a deleted
// or inserted base): the lower clamp is "a read opens only the windows whose first column it walks".  Per operation over rp = A..B:
//   * the alignment's tuple count is again a difference of running sums of the candidate prefix, the clamped part (rp < POS + 34) a
//     multiple of one prefix value;
//   * the tuples PER WINDOW are a trapezoid over u (a ramp of min(length, window width) centres up, a plateau, the same ramp down), cut off
//     below POS + 17: six second-difference entries over the centre VALUE instead of two range additions per base over the candidate index;
//     two running sums over the positions and a gather at the centres give the totals (fe_tuple_scan_*, fe_window_totals_at_kernel).
__device__ inline void d2_add(const Region &g, const Candidates &c, int64_t u, long long v) {
    const int64_t i = u - g.lo;
    if (i < 0) { atomicAdd(&c.fold[0], (unsigned long long)v); atomicAdd(&c.fold[1], (unsigned long long)(v * u)); }
    else if (i <= g.n + 1) atomicAdd(&c.d2[i], (unsigned long long)v);
}
// tuples a run of bases rp = A..B adds to the window of centre u, for u >= c0: #{rp: rp - 17 <= u <= rp + reach}
__device__ inline void d2_add_run(const Region &g, const Candidates &c, int64_t A, int64_t B, int reach, int64_t c0) {
    if (B < A || B + reach < c0) return;
    const int64_t L = B - A + 1, W = 17 + reach + 1, m = L < W ? L : W;
    int64_t from = INT64_MIN / 4;
    if (A - 17 < c0) {            // the window of c0 and its right neighbours only: a step up to T(c0), then the rest of the shape
        const int64_t hi = B < c0 + 17 ? B : c0 + 17, lo = A > c0 - reach ? A : c0 - reach;
        const long long t0 = hi >= lo ? hi - lo + 1 : 0;
        if (t0) { d2_add(g, c, c0, t0); d2_add(g, c, c0 + 1, -t0); }
        from = c0 + 1;
    }
    const int64_t u0 = A - 17 > from ? A - 17 : from, u1 = A - 17 + m - 1;            // first differences +1
    if (u0 <= u1) { d2_add(g, c, u0, 1); d2_add(g, c, u1 + 1, -1); }
    const int64_t w0 = B + reach + 2 - m > from ? B + reach + 2 - m : from, w1 = B + reach + 1;   // first differences -1
    if (w0 <= w1) { d2_add(g, c, w0, -1); d2_add(g, c, w1 + 1, 1); }
}

__global__ __launch_bounds__(256) void fe_windows_per_op_noleft_kernel(Region g, SlabView s, Candidates c) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    uint32_t read = 0xffffffffu;
    unsigned long long nc = 0;
    if (j < s.n_ops) {
        const clair_op_t op = s.ops[j];
        const clair_read_t r = s.reads[op.read];
        if (r.flags & CLAIR_READ_PILE) {
            read = op.read;
            const uint32_t code = op.code_len & 3u;
            const int64_t len = op.code_len >> 2, pos0 = r.pos0, c0 = pos0 + 17;
            int64_t a = pos0 + op.ref_off, b = a + len - 1;          // first and last reference position of the run (I: both the same)
            auto below = [&](int64_t A, int64_t B) -> unsigned long long {      // sum over rp = A..B of the prefix at max(rp - 18, POS + 16)
                unsigned long long v = 0;
                const int64_t r1 = B < pos0 + 33 ? B : pos0 + 33;
                if (r1 >= A) v += (unsigned long long)(r1 - A + 1) * before_at(g, c, pos0 + 16 - g.lo);
                const int64_t a2 = A > pos0 + 34 ? A : pos0 + 34;
                if (B >= a2) v += prefix_sum_at(g, c, B - 18) - prefix_sum_at(g, c, a2 - 19);
                return v;
            };
            if (code == CLAIR_OP_M) {
                nc += (prefix_sum_at(g, c, b + 17) - prefix_sum_at(g, c, a + 16)) - below(a, b);
                d2_add_run(g, c, a, b, 17, c0);
            } else if (code == CLAIR_OP_D) {
                if (a == pos0) ++a;                                    // offered before any window is open
                if (b >= a) {
                    nc += (prefix_sum_at(g, c, b + 16) - prefix_sum_at(g, c, a + 15)) - below(a, b);
                    d2_add_run(g, c, a, b, 16, c0);
                }
            } else if (a > pos0) {
                const int64_t lo_u = a - 17 > c0 ? a - 17 : c0;
                if (lo_u <= a + 16) {
                    nc += (unsigned long long)len * (before_at(g, c, a + 16 - g.lo) - before_at(g, c, lo_u - 1 - g.lo));
                    d2_add(g, c, lo_u, len); d2_add(g, c, lo_u + 1, -len);
                    d2_add(g, c, a + 17, -len); d2_add(g, c, a + 18, len);
                }
                const int so = (r.flags & CLAIR_READ_REVERSE) ? 4 : 0;
                const int64_t first = a - 15 > c0 ? a - 15 : c0;
                const uint32_t i0 = before_at(g, c, first - 1 - g.lo), i1 = before_at(g, c, a + 16 - g.lo);
                for (int64_t k = 0; k < len && i1 > i0; ++k) {         // generate_tensor :51-53: column min(idx + k, 32), channel 1
                    if (op.q_off + k >= r.seq_len) break;
                    const uint8_t row = BASES.pile[s.seq[r.seq0 + op.q_off + k]];
                    if (row == 255) continue;
                    for (uint32_t i = i0; i < i1; ++i) {
                        const int64_t col = a - c.centre[i] + 17 + k;
                        atomicAdd(&c.ins[((size_t)i * N_POS + (col < N_POS - 1 ? col : N_POS - 1)) * N_ROW + row + so], 1u);
                    }
                }
            }
        }
    }
    unsigned long long todo = __ballot(nc != 0);                       // one atomic per run of equal reads in the wave
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

// what the first 32 walked positions of an alignment put into the tables belongs to none of the windows it starts inside (centres rp - 15 ..
// POS + 16): collected per window and column, the assembly takes it out again.  One thread per alignment: at most 32 positions of it.
__global__ __launch_bounds__(256) void fe_late_starters_kernel(Region g, SlabView s, Candidates c, uint32_t n_reads) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_reads) return;
    const clair_read_t r = s.reads[i];
    if (!(r.flags & CLAIR_READ_PILE)) return;
    const int so = (r.flags & CLAIR_READ_REVERSE) ? 4 : 0;
    const uint32_t j1 = before_at(g, c, r.pos0 + 16 - g.lo);
    for (uint32_t o = 0; o < r.n_ops; ++o) {
        const clair_op_t op = s.ops[r.op0 + o];
        if (op.ref_off > 31) break;
        const uint32_t code = op.code_len & 3u;
        if (code == CLAIR_OP_I) continue;
        const int64_t len = op.code_len >> 2;
        for (int64_t k = 0; k < len && op.ref_off + k <= 31; ++k) {
            const int64_t rp = r.pos0 + op.ref_off + k;
            if (code == CLAIR_OP_D && rp <= r.pos0) continue;
            uint8_t row = 255;
            if (code == CLAIR_OP_M) {
                const uint32_t qp = op.q_off + (uint32_t)k;
                if (qp < r.seq_len) row = BASES.pile[s.seq[r.seq0 + qp]];
                if (row == 255) continue;
            }
            for (uint32_t j = before_at(g, c, rp - 15 - 1 - g.lo); j < j1; ++j) {
                uint32_t *cell = c.late + ((size_t)j * N_POS + (rp - c.centre[j] + 17)) * LATE_ROW;
                if (code == CLAIR_OP_M) { atomicAdd(cell + row + so, 1u); atomicAdd(cell + 8 + (so >> 2), 1u); }
                else atomicAdd(cell + 10 + (so >> 2), 1u);
            }
        }
    }
}

__device__ inline unsigned long long block_exclusive_scan64(unsigned long long v, unsigned long long *total);   // below, with the candidate prefix's running sum

// running sums of a 64-bit array in place, twice in a row: d2 -> first differences (seeded with fold[0]) -> tuples per centre value (seeded with
// the tuples of u = lo - 1: lo fold[0] - fold[1]).  Two's complement throughout: signed sums come out right modulo 2^64.
__global__ __launch_bounds__(256) void fe_tuple_scan_sums_kernel(const unsigned long long *v, int64_t n, unsigned long long *block_sum) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long x = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) if (at + i < n) x += v[at + i];
    unsigned long long total;
    (void)block_exclusive_scan64(x, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void fe_tuple_scan_write_kernel(unsigned long long *v, int64_t n, const unsigned long long *block_sum, const unsigned long long *fold,
                                                                  int second, int64_t lo) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long f[SCAN_ITEMS], x = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) { f[i] = at + i < n ? v[at + i] : 0; x += f[i]; }
    unsigned long long total;
    const unsigned long long seed = second ? (unsigned long long)lo * fold[0] - fold[1] : fold[0];
    unsigned long long run = seed + block_sum[blockIdx.x] + block_exclusive_scan64(x, &total);
    for (int i = 0; i < SCAN_ITEMS && at + i < n; ++i) { run += f[i]; v[at + i] = run; }
}
__global__ __launch_bounds__(256) void fe_window_totals_at_kernel(const unsigned long long *tuples_at, const int64_t *centre, int64_t n_candidates, int64_t lo, int64_t n,
                                                                  uint64_t *window_tuples) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_candidates) return;
    const int64_t at = centre[i] - lo;
    window_tuples[i] = at >= 0 && at <= n + 1 ? tuples_at[at] : 0;
}

// before[0..n] -> before_sum (inclusive running sum, 64 bit): block sums, one workgroup over them, write
__device__ inline unsigned long long block_exclusive_scan64(unsigned long long v, unsigned long long *total) {   // 256 threads
    __shared__ unsigned long long wave_sum64[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long x = v;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wave_sum64[w] = x;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (int i = 0; i < 4; ++i) { if (i < w) before += wave_sum64[i]; all += wave_sum64[i]; }
    __syncthreads();
    *total = all;
    return before + x - v;
}

__global__ __launch_bounds__(256) void fe_prefix_block_sums_kernel(const uint32_t *before, int64_t n, unsigned long long *block_sum) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long v = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) if (at + i < n) v += before[at + i];
    unsigned long long total;
    (void)block_exclusive_scan64(v, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void fe_prefix_scan_sums_kernel(unsigned long long *block_sum, int64_t n_blocks) {
    unsigned long long carry = 0;
    for (int64_t at = 0; at < n_blocks; at += 256) {
        const int64_t i = at + threadIdx.x;
        unsigned long long total;
        const unsigned long long ex = block_exclusive_scan64(i < n_blocks ? block_sum[i] : 0, &total);
        if (i < n_blocks) block_sum[i] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(256) void fe_prefix_write_kernel(const uint32_t *before, int64_t n, const unsigned long long *block_sum, unsigned long long *before_sum) {
    const int64_t at = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t f[SCAN_ITEMS];
    unsigned long long v = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) { f[i] = at + i < n ? before[at + i] : 0; v += f[i]; }
    unsigned long long total;
    unsigned long long run = block_sum[blockIdx.x] + block_exclusive_scan64(v, &total);
    for (int i = 0; i < SCAN_ITEMS && at + i < n; ++i) { run += f[i]; before_sum[at + i] = run; }
}

// ---- per candidate: was its window ever opened, how many tuples did it hold, does it survive -----------------------------------
struct WindowRule {
    int min_coverage;
    int drop_non_iupac_centre;   // clair/utils.py:90-91
    const uint32_t *late;        // --stop_consider_left_edge: see Candidates; window_tuples then comes in filled (fe_window_totals_kernel)
};

// --stop_consider_left_edge: tuples per window = running sum of the range additions of pass 2
__global__ __launch_bounds__(256) void fe_window_totals_kernel(const int *diff, int64_t n_candidates, uint64_t *window_tuples) {
    long long carry = 0;
    for (int64_t at = 0; at < n_candidates; at += 256) {
        const int64_t i = at + threadIdx.x;
        const int v = i < n_candidates ? diff[i] : 0;
        // signed values: scan the positive and the negative parts apart (the exclusive scan is unsigned)
        uint32_t tp, tn;
        const uint32_t ep = block_exclusive_scan(v > 0 ? (uint32_t)v : 0u, &tp);
        const uint32_t en = block_exclusive_scan(v < 0 ? (uint32_t)(-v) : 0u, &tn);
        if (i < n_candidates) window_tuples[i] = (uint64_t)(carry + (long long)ep - (long long)en + v);
        carry += (long long)tp - (long long)tn;
    }
}

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
        const ulonglong2 w01 = *(const ulonglong2 *)(g.pq + t * 4), w23 = *(const ulonglong2 *)(g.pq + t * 4 + 2);
        const uint4 a = *(const uint4 *)(g.misc + t * 8);
        const uint32_t start_m = g.misc[t * 8 + 4];
        uint64_t m = 0;
        for (int field = 0; field < 2; ++field) m += (uint64_t)pq_field(w01.x, field) + pq_field(w01.y, field) + pq_field(w23.x, field) + pq_field(w23.y, field);
        const uint64_t d = (uint64_t)a.x + a.y;
        if (rule.late ? rp == c - 17 : rp <= c + 16) walked += m + d + a.z;   // M or D (incl. a read's first D) opens the window
        tuples += m;
        if (rp == c + 17) tuples -= start_m;                        // a read that STARTS there never opened this window
        if (rp >= c - 16) tuples += d + a.w;
        if (rp == c - 1) depth_centre = (uint32_t)m;
    }
    const bool opened = walked > 0;
    if (rule.late) {
        const uint32_t *cell = rule.late + ((size_t)i * N_POS + 16) * LATE_ROW;
        depth_centre -= cell[8] + cell[9];
        tuples = window_tuples[i];
    }
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
                                                          const uint32_t *late, short4 *counts, int64_t *out_centre, uint8_t *out_refseq) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t o = id / N_QUAD;
    if (o >= n_kept) return;
    const int quad = (int)(id - o * N_QUAD), idx = quad >> 3, row = quad & 7, so = row >> 2, b = row & 3;
    const int64_t ci = kept[o], c = centre[ci];
    const int64_t rp = c - 17 + idx, t = rp - g.lo;
    uint32_t qv = 0, mw = 0, dw = 0;
    if (t >= 0 && t < g.n) {
        const ulonglong2 w01 = *(const ulonglong2 *)(g.pq + t * 4), w23 = *(const ulonglong2 *)(g.pq + t * 4 + 2);
        const unsigned long long w[4] = {w01.x, w01.y, w23.x, w23.y};
        qv = pq_field(w[b], so);
        mw = pq_field(w[0], so) + pq_field(w[1], so) + pq_field(w[2], so) + pq_field(w[3], so);
        dw = g.misc[t * 8 + so];
    }
    if (late) {
        const uint32_t *cell = late + ((size_t)ci * N_POS + idx) * LATE_ROW;
        qv -= cell[row];
        mw -= cell[8 + so];
        dw -= cell[10 + so];
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

// ---- `samtools view` text parsed on the device: the line handling of clair_host_sampack_* (hostsrc/host_sampack.cpp), one thread per line ---
// What the host packer does per line -- split on whitespace, FLAG / RNAME / POS / MAPQ / CIGAR / SEQ, the two stages' filters, the CIGAR
// reduced to M / I / D operations -- on the text where the copy from the pipe left it; the host only moves bytes.  Of the text only the
// SEQ columns of the kept alignments stay resident (fe_text_seq_kernel packs them, 4-byte aligned per alignment, into the slab's `seq`):
// QNAME, QUAL and the tags are more than half of a line, and a region's text is what fills the HBM first (64 B per position of tables
// against ~75 B of text per position at 30x).  The per-read state the scripts carry from line
// to line (--dcov, sortedness) is recovered from the sorted start positions (a read's rank among the pileup reads of its start
// position is its index minus the lower bound of that position) and carried from chunk to chunk by the host.
struct TextOptions {
    const uint8_t *ctg;
    int ctg_len;
    int dcov, evc_min_mq, pile_min_mq;
    int64_t pile_start, pile_end;        // 1-based inclusive, -1 -1: none
};

struct TextLine {                        // what pass A learns about a line
    int64_t pos0;
    uint32_t cigar_off, cigar_len, seq_off, seq_len;
    uint32_t n_ops, n_elem;
    uint32_t flags;                      // CLAIR_READ_REVERSE | CLAIR_READ_EVC | TL_CANDIDATE | TL_ZERO_INDEL | TL_LONG_SPAN | TL_LEAD_INDEL
    uint32_t pad;
};
enum { TL_CANDIDATE = 16, TL_ZERO_INDEL = 32, TL_LONG_SPAN = 64, TL_LEAD_INDEL = 128 };   // TL_LEAD_INDEL: an I / D while the reference cursor is still at POS

struct TextState {                       // carried from chunk to chunk (host copy in clair_frontend)
    int64_t prev_pos, depth_cap;         // CreateTensor.py:249-250, 277-287
    int64_t last_pos, have_last;         // sortedness of the kept alignments
    int64_t evc_last_pos, have_evc_last; // POS of the last alignment the candidate search accepted (CLAIR_FE_LEAD_INDEL)
    int64_t lines, evc_reads, pile_reads;
    uint32_t anomalies, malformed;       // malformed: 1 + index of the first line the host packer would reject
};

__device__ inline bool text_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

// end of the token that starts at q: 16 bytes at a time while none of them is below 0x21 (every white-space character is; SEQ and QUAL
// bytes are not), byte by byte from the first word that holds one
__device__ inline int64_t token_end(const uint8_t *t, int64_t q, int64_t end) {
    while (q < end && (q & 15)) { if (text_space(t[q])) return q; ++q; }
    while (q + 16 <= end) {
        const uint4 w = *(const uint4 *)(t + q);
        const uint32_t low = ((w.x - 0x21212121u) & ~w.x) | ((w.y - 0x21212121u) & ~w.y) | ((w.z - 0x21212121u) & ~w.z) | ((w.w - 0x21212121u) & ~w.w);
        if (low & 0x80808080u) break;
        q += 16;
    }
    while (q < end && !text_space(t[q])) ++q;
    return q;
}

// walks a CIGAR; EMIT writes the kept operations
template <bool EMIT>
__device__ inline void walk_cigar(const uint8_t *cg, uint32_t cl, uint32_t read, clair_op_t *ops, uint32_t *op_elem, uint32_t elem0,
                                  int64_t *o_rp, int64_t *o_qp, int64_t *o_soft, int64_t *o_total, int64_t *o_rlen, uint32_t *o_ops, uint64_t *o_elems, bool *o_zero,
                                  bool *o_lead = nullptr) {
    int64_t adv = 0, rp = 0, qp = 0, soft = 0, total = 0, rlen = 0;
    uint32_t n_ops = 0;
    uint64_t elems = 0;
    bool zero = false, lead = false;
    for (uint32_t i = 0; i < cl; ++i) {
        const uint8_t ch = cg[i];
        if (ch >= '0' && ch <= '9') { adv = adv * 10 + (ch - '0'); if (adv > ((int64_t)1 << 40)) adv = (int64_t)1 << 40; continue; }
        int code = -1;
        switch (ch) {
        case 'S': soft += adv; qp += adv; break;
        case 'M': case '=': case 'X': code = CLAIR_OP_M; break;
        case 'I': code = CLAIR_OP_I; break;
        case 'D': code = CLAIR_OP_D; break;
        case 'N': rlen += adv; break;
        default: break;
        }
        if (code >= 0) {
            if (adv > 0) {
                const int64_t len = adv > 0x3fffffff ? 0x3fffffff : adv;
                if (EMIT) {
                    ops[n_ops] = clair_op_t{read, (uint32_t)len << 2 | (uint32_t)code, (int32_t)rp, (uint32_t)qp};
                    op_elem[n_ops] = elem0 + (uint32_t)elems;
                }
                ++n_ops;
                elems += (uint64_t)len;
                if (code != CLAIR_OP_M && rp == 0) lead = true;       // tallied at POS - 1 by the candidate search (EVC :326-336)
            } else if (code != CLAIR_OP_M) {
                zero = true;
            }
            if (code != CLAIR_OP_I) { rp += adv; rlen += adv; }
            if (code != CLAIR_OP_D) qp += adv;
        }
        total += adv;
        adv = 0;
    }
    if (!EMIT) { *o_rp = rp; *o_qp = qp; *o_soft = soft; *o_total = total; *o_rlen = rlen; *o_ops = n_ops; *o_elems = elems; *o_zero = zero; if (o_lead) *o_lead = lead; }
}

__device__ inline bool text_int(const uint8_t *s, uint32_t len, int64_t *out) {     // [+-]digits, at most 18 of them (host_sampack.cpp)
    uint32_t i = 0;
    bool neg = false;
    if (i < len && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; ++i; }
    if (i == len || len - i > 18) return false;
    int64_t x = 0;
    for (; i < len; ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        x = x * 10 + (s[i] - '0');
    }
    *out = neg ? -x : x;
    return true;
}

__global__ __launch_bounds__(256) void fe_text_lines_kernel(const uint8_t *text, const int64_t *newline, int64_t n_lines, TextOptions opt, TextLine *lines,
                                                            uint8_t *is_candidate, TextState *state) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_lines) return;
    TextLine out{};
    is_candidate[k] = 0;
    const int64_t begin = k == 0 ? 0 : newline[k - 1] + 1, end = newline[k];
    uint32_t col[10], len[10];
    int n = 0;
    int64_t p = begin;
    while (p < end && n < 10) {
        while (p < end && text_space(text[p])) ++p;
        if (p >= end) break;
        const int64_t q = token_end(text, p, end);
        col[n] = (uint32_t)p;
        len[n] = (uint32_t)(q - p);
        ++n;
        p = q;
    }
    bool bad = n == 0;
    if (!bad && text[col[0]] == '@') { lines[k] = out; return; }            // header line
    bad = bad || n < 10;
    int64_t flag = 0, pos1 = 0, mq = 0;
    bad = bad || !text_int(text + col[1], len[1], &flag) || !text_int(text + col[3], len[3], &pos1) || !text_int(text + col[4], len[4], &mq);
    if (bad) {                                                                // the host packer names the line and the column
        atomicMin(&state->malformed, (uint32_t)(k + 1));
        lines[k] = out;
        return;
    }
    bool same_ctg = (int)len[2] == opt.ctg_len;
    for (int i = 0; same_ctg && i < opt.ctg_len; ++i) same_ctg = text[col[2] + i] == opt.ctg[i];
    int64_t rp, qp, soft, total, rlen;
    uint32_t n_ops;
    uint64_t elems;
    bool zero, lead;
    walk_cigar<false>(text + col[5], len[5], 0, nullptr, nullptr, 0, &rp, &qp, &soft, &total, &rlen, &n_ops, &elems, &zero, &lead);
    const bool evc_ok = same_ctg && mq >= opt.evc_min_mq && !(len[5] == 1 && text[col[5]] == '*') && !(1.0 - (double)soft / (double)(total + 1) < 0.55);
    bool in_region = true;
    if (opt.pile_start >= 0 && opt.pile_end >= 0) {
        const int64_t end1 = pos1 + (rlen > 0 ? rlen : 1) - 1;                // bam_endpos
        in_region = same_ctg && pos1 <= opt.pile_end && end1 >= opt.pile_start;
    }
    const bool candidate = in_region && mq >= opt.pile_min_mq;
    out.pos0 = pos1 - 1;
    out.cigar_off = col[5]; out.cigar_len = len[5];
    out.seq_off = col[9]; out.seq_len = len[9];
    out.n_ops = n_ops;
    out.n_elem = elems > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)elems;
    out.flags = ((flag & 16) ? CLAIR_READ_REVERSE : 0) | (evc_ok ? CLAIR_READ_EVC : 0) | (candidate ? TL_CANDIDATE : 0) | (zero ? TL_ZERO_INDEL : 0) | (lead ? TL_LEAD_INDEL : 0)
                | ((rp > (int64_t)len[9] + 100000 - 64 || rp > 0x7fffff00 || qp > 0x7fffff00) ? TL_LONG_SPAN : 0);
    lines[k] = out;
    is_candidate[k] = candidate ? 1 : 0;
}

// --dcov: the rank of a pileup read among those of its start position = its index among the candidates minus the lower bound of its
// position (start positions ascend; where they do not, CLAIR_FE_UNSORTED sends the run to the host anyway)
__global__ __launch_bounds__(256) void fe_text_dcov_kernel(TextLine *lines, const int64_t *cand, int64_t n_cand, int dcov, TextState carry) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_cand) return;
    TextLine &ln = lines[cand[j]];
    const int64_t pos = ln.pos0;
    int64_t a = 0, b = j;                                                     // first candidate with this start position
    while (a < b) { const int64_t mid = (a + b) >> 1; if (lines[cand[mid]].pos0 < pos) a = mid + 1; else b = mid; }
    int64_t depth_cap = j - a;
    if (a == 0 && pos == carry.prev_pos) depth_cap += carry.depth_cap + 1;   // the run began in an earlier chunk (or is the scripts' initial previous_position = 0)
    uint32_t fl = ln.flags;
    if (depth_cap < dcov) fl |= CLAIR_READ_PILE | (depth_cap == 0 ? CLAIR_READ_FLUSH : 0);
    ln.flags = fl;
}

__global__ __launch_bounds__(256) void fe_text_keep_kernel(const TextLine *lines, int64_t n_lines, uint8_t *keep) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_lines) return;
    keep[k] = (lines[k].flags & (CLAIR_READ_EVC | CLAIR_READ_PILE)) ? 1 : 0;
}

// one workgroup: exclusive sums of the kept lines' operation and element counts, sortedness, counters, the state for the next chunk
__global__ __launch_bounds__(256) void fe_text_offsets_kernel(const TextLine *lines, int64_t n_lines, const int64_t *kept, int64_t n_kept, const int64_t *cand, int64_t n_cand,
                                                              uint32_t *op0, uint32_t *elem0, uint32_t *seq0, uint64_t *totals, TextState carry, TextState *state) {
    __shared__ uint64_t s_ops, s_elems;
    __shared__ uint32_t s_anom, s_reach;
    __shared__ unsigned long long s_evc, s_pile;
    __shared__ unsigned long long s_last_evc;                                 // 1 + index (among the kept lines) of the last one the candidate search accepts
    if (threadIdx.x == 0) { s_ops = 0; s_elems = 0; s_anom = 0; s_reach = 0; s_evc = 0; s_pile = 0; s_last_evc = 0; }
    __syncthreads();
    uint64_t carry_ops = 0, carry_elems = 0, carry_seq = 0;
    uint32_t anom = 0, reach = 0;                 // reach: the longest alignment in elements (no reference span is longer)
    unsigned long long evc = 0, pile = 0;
    for (int64_t at = 0; at < n_kept; at += 256) {
        const int64_t i = at + threadIdx.x;
        uint32_t a = 0, b = 0, c = 0;
        if (i < n_kept) {
            const TextLine ln = lines[kept[i]];
            a = ln.n_ops;
            b = ln.n_elem;
            c = (ln.seq_len + 3u) & ~3u;          // the alignment's bases in the packed buffer: a whole number of dwords
            reach = max(reach, b);
            const int64_t before = i > 0 ? lines[kept[i - 1]].pos0 : (carry.have_last ? carry.last_pos : ln.pos0);
            if (ln.pos0 < before) anom |= CLAIR_FE_UNSORTED;
            if ((ln.flags & TL_ZERO_INDEL) && (ln.flags & CLAIR_READ_EVC)) anom |= CLAIR_FE_ZERO_INDEL;
            if (ln.flags & TL_LONG_SPAN) anom |= CLAIR_FE_LONG_SPAN;
            if (ln.flags & CLAIR_READ_EVC) {
                atomicMax(&s_last_evc, (unsigned long long)(i + 1));
                if (ln.flags & TL_LEAD_INDEL) {                                // rare: look back over the run of equal start positions for an accepted alignment
                    bool earlier = false;
                    int64_t j = i - 1;
                    for (; j >= 0; --j) {
                        const TextLine &b4 = lines[kept[j]];
                        if (b4.pos0 != ln.pos0) break;
                        if (b4.flags & CLAIR_READ_EVC) { earlier = true; break; }
                    }
                    if (!earlier && j < 0 && carry.have_evc_last && carry.evc_last_pos == ln.pos0) earlier = true;   // the run began in an earlier chunk
                    if (earlier) anom |= CLAIR_FE_LEAD_INDEL;
                }
            }
            evc += (ln.flags & CLAIR_READ_EVC) ? 1 : 0;
            pile += (ln.flags & CLAIR_READ_PILE) ? 1 : 0;
        }
        uint32_t ta, tb, tc;
        const uint32_t ea = block_exclusive_scan(a, &ta);
        const uint32_t eb = block_exclusive_scan(b, &tb);
        const uint32_t ec = block_exclusive_scan(c, &tc);
        if (i < n_kept) { op0[i] = (uint32_t)(carry_ops + ea); elem0[i] = (uint32_t)(carry_elems + eb); seq0[i] = (uint32_t)(carry_seq + ec); }
        carry_ops += ta;
        carry_elems += tb;
        carry_seq += tc;
    }
    atomicOr(&s_anom, anom);
    atomicMax(&s_reach, reach);
    atomicAdd(&s_evc, evc);
    atomicAdd(&s_pile, pile);
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = carry_ops;
        totals[1] = carry_elems;
        totals[2] = s_reach;
        totals[3] = n_kept ? (uint64_t)lines[kept[0]].pos0 : 0;       // where the slab begins (the kept lines are sorted, or CLAIR_FE_UNSORTED is raised)
        totals[4] = carry_seq;                                        // bytes of the packed bases
        TextState st = carry;
        st.anomalies = carry.anomalies | s_anom;
        st.malformed = state->malformed;
        st.lines = carry.lines + n_lines;
        st.evc_reads = carry.evc_reads + (int64_t)s_evc;
        st.pile_reads = carry.pile_reads + (int64_t)s_pile;
        if (n_kept) { st.last_pos = lines[kept[n_kept - 1]].pos0; st.have_last = 1; }
        if (s_last_evc) { st.evc_last_pos = lines[kept[s_last_evc - 1]].pos0; st.have_evc_last = 1; }
        if (n_cand) {                                                         // previous_position / depthCap after the last pileup candidate
            const int64_t pos = lines[cand[n_cand - 1]].pos0;
            int64_t a = 0, b = n_cand - 1;
            while (a < b) { const int64_t mid = (a + b) >> 1; if (lines[cand[mid]].pos0 < pos) a = mid + 1; else b = mid; }
            int64_t depth_cap = n_cand - 1 - a;
            if (a == 0 && pos == carry.prev_pos) depth_cap += carry.depth_cap + 1;
            st.prev_pos = pos;
            st.depth_cap = depth_cap;
        }
        *state = st;
    }
}

__global__ __launch_bounds__(256) void fe_text_emit_kernel(const uint8_t *text, const TextLine *lines, const int64_t *kept, int64_t n_kept, const uint32_t *op0,
                                                           const uint32_t *elem0, const uint32_t *seq0, clair_read_t *reads, clair_op_t *ops, uint32_t *op_elem,
                                                           uint64_t total_ops, uint64_t total_elems) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) op_elem[total_ops] = (uint32_t)total_elems;
    if (i >= n_kept) return;
    const TextLine ln = lines[kept[i]];
    walk_cigar<true>(text + ln.cigar_off, ln.cigar_len, (uint32_t)i, ops + op0[i], op_elem + op0[i], elem0[i], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr);
    clair_read_t r;
    r.pos0 = ln.pos0;
    r.seq0 = seq0[i];
    r.seq_len = ln.seq_len;
    r.op0 = op0[i];
    r.n_ops = ln.n_ops;
    r.flags = ln.flags & (CLAIR_READ_REVERSE | CLAIR_READ_EVC | CLAIR_READ_PILE | CLAIR_READ_FLUSH);
    r.reserved = 0;
    reads[i] = r;
}

// The SEQ column of every kept line into the slab's packed buffer: a wave per alignment, a dword per lane and step.  The source starts at any
// byte of the text: two aligned dwords and v_alignbyte give the four bytes from there (the text buffer is allocated 16 bytes longer than the text).
__global__ __launch_bounds__(256) void fe_text_seq_kernel(const uint8_t *text, const TextLine *lines, const int64_t *kept, int64_t n_kept, const uint32_t *seq0, uint8_t *seq) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_kept) return;
    const int lane = threadIdx.x & 63;
    const TextLine ln = lines[kept[i]];
    const uint32_t n_dw = (ln.seq_len + 3u) >> 2;
    const uint32_t *src = (const uint32_t *)(text + (ln.seq_off & ~3u));
    const uint32_t shift = ln.seq_off & 3u;
    uint32_t *dst = (uint32_t *)(seq + seq0[i]);
    for (uint32_t k = (uint32_t)lane; k < n_dw; k += 64) {
        const uint32_t lo = src[k], hi = src[k + 1];
        dst[k] = __builtin_amdgcn_alignbyte(hi, lo, shift);
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
    unsigned long long *d_before_sum = nullptr, *d_prefix_block_sum = nullptr;   // running sum of d_before (pass 2 per operation)
    uint32_t *d_late = nullptr;          // --stop_consider_left_edge only
    int *d_tuple_diff = nullptr;
    uint8_t *d_keep = nullptr;
    uint64_t *d_window_tuples = nullptr;
    int64_t *d_kept = nullptr;
    uint32_t *d_cand_block_sum = nullptr;
    short4 *d_counts = nullptr;
    int64_t *d_out_centre = nullptr;
    uint8_t *d_out_refseq = nullptr;
    int64_t n_windows = -1;
    // text parsed on the device (clair_frontend_text_options / _add_text)
    bool text_ready = false;
    TextOptions text_opt{};
    uint8_t *d_ctg = nullptr;
    TextState text_state{};
    TextState *d_text_state = nullptr;
    uint32_t *d_span = nullptr;          // the longest reference reach of the slab being tallied (fe_slab_span_kernel)
    unsigned long long *d_d2 = nullptr, *d_fold = nullptr;   // --stop_consider_left_edge, pass 2 per operation: tuples per window as second differences over positions
    bool pass2_per_base = false;         // CLAIR_AMD_FE_PASS2=base when the handle was created: --stop_consider_left_edge's pass 2 by fe_windows_per_base_kernel
    bool tally_per_base = false;         // CLAIR_AMD_FE_TALLY=atomic when the handle was created: pass 1 by fe_tally_kernel (one device atomic per base)
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
int scan_count(clair_frontend *f, const uint8_t *flags, int64_t n, uint32_t *block_sum, uint32_t *d_count, int64_t *count, bool newlines = false) {
    const unsigned nb = blocks_for(n, SCAN_BLOCK);
    if (n > 0 && newlines) hipLaunchKernelGGL(fe_block_count_kernel<true>, dim3(nb), dim3(256), 0, f->stream, flags, n, block_sum);
    else if (n > 0) hipLaunchKernelGGL(fe_block_count_kernel<false>, dim3(nb), dim3(256), 0, f->stream, flags, n, block_sum);
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
               int64_t list_base, bool newlines = false) {
    // entry n of the prefix is written by the thread that owns index n: cover it with one more (empty) item
    const unsigned nb = blocks_for(n, SCAN_BLOCK), nbw = blocks_for(n + 1, SCAN_BLOCK);
    if (nbw > nb) FE_TRY(f, hipMemcpyAsync(block_sum + nb, d_count, sizeof(uint32_t), hipMemcpyDeviceToDevice, f->stream));
    if (newlines) hipLaunchKernelGGL(fe_scan_write_kernel<true>, dim3(nbw), dim3(256), 0, f->stream, flags, n, (const uint32_t *)block_sum, prefix, list, list_base);
    else hipLaunchKernelGGL(fe_scan_write_kernel<false>, dim3(nbw), dim3(256), 0, f->stream, flags, n, (const uint32_t *)block_sum, prefix, list, list_base);
    FE_TRY(f, hipGetLastError());
    return 0;
}

void free_candidates(clair_frontend *f) {
    (void)hipFree(f->d_centre); f->d_centre = nullptr;
    (void)hipFree(f->d_ins); f->d_ins = nullptr;
    (void)hipFree(f->d_before_sum); f->d_before_sum = nullptr;
    (void)hipFree(f->d_prefix_block_sum); f->d_prefix_block_sum = nullptr;
    (void)hipFree(f->d_late); f->d_late = nullptr;
    (void)hipFree(f->d_tuple_diff); f->d_tuple_diff = nullptr;
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
    { const char *t = getenv("CLAIR_AMD_FE_TALLY"); f->tally_per_base = t && !strcmp(t, "atomic"); }
    { const char *t = getenv("CLAIR_AMD_FE_PASS2"); f->pass2_per_base = t && !strcmp(t, "base"); }
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
    const size_t table = (size_t)g.n * 32;
    if ((err = hipMalloc((void **)&g.pq, table)) != hipSuccess) return bail("hipMalloc(read-base counters)", err);
    if ((err = hipMalloc((void **)&g.misc, table)) != hipSuccess) return bail("hipMalloc(counts)", err);
    if ((err = hipMalloc((void **)&g.anomalies, sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void **)&f->d_flags, (size_t)g.n + 1)) != hipSuccess) return bail("hipMalloc(flags)", err);
    if ((err = hipMalloc((void **)&f->d_before, ((size_t)g.n + 1) * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(prefix)", err);
    if ((err = hipMalloc((void **)&f->d_block_sum, ((size_t)blocks_for(g.n + 1, SCAN_BLOCK) + 1) * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void **)&f->d_total, 2 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", err);
    (void)hipMemsetAsync(g.pq, 0, table, f->stream);
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
    (void)hipFree(f->d_ref); (void)hipFree(f->g.pq); (void)hipFree(f->g.misc); (void)hipFree(f->g.anomalies);
    (void)hipFree(f->d_flags); (void)hipFree(f->d_before); (void)hipFree(f->d_block_sum); (void)hipFree(f->d_total); (void)hipFree(f->d_bed);
    (void)hipFree(f->d_ctg); (void)hipFree(f->d_text_state); (void)hipFree(f->d_span); (void)hipFree(f->d_d2); (void)hipFree(f->d_fold);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

__global__ void fe_flag_kernel(Region g, uint32_t bits) { atomicOr(g.anomalies, bits); }

// Pass 1 over one slab.  The tiled kernel wherever its assumptions hold (fewer alignments in the slab than a 21-bit counter holds, so that a
// tile's counters cannot carry into their neighbours); CLAIR_AMD_FE_TALLY=atomic keeps the per-base kernel (tests compare the two, bit for bit).
// [p_first, p_reach): reference positions the slab may touch (a generous guess is fine, see fe_tally_tile_kernel).
static int launch_tally(clair_frontend *f, const Slab &d, int64_t p_first, int64_t p_reach, bool sorted = true) {
    if (!d.n_elem) return 0;
    if (f->tally_per_base || d.n_reads >= (int64_t)PQ_MASK || !sorted) {   // the tiled kernel finds a tile's alignments by bisection over their starts
        hipLaunchKernelGGL(fe_tally_kernel, dim3(blocks_for(d.n_elem, 256)), dim3(256), 0, f->stream, f->g, f->view(d));
        return 0;
    }
    if (!f->d_span) FE_TRY(f, hipMalloc((void **)&f->d_span, sizeof(uint32_t)));
    FE_TRY(f, hipMemsetAsync(f->d_span, 0, sizeof(uint32_t), f->stream));
    hipLaunchKernelGGL(fe_slab_span_kernel, dim3(blocks_for(d.n_reads, 256)), dim3(256), 0, f->stream, f->view(d), (uint32_t)d.n_reads, f->d_span);
    const int64_t n_tiles = (f->g.n + TT_POS - 1) / TT_POS;
    const int64_t tile0 = std::min(std::max<int64_t>((p_first - f->g.lo) / TT_POS - (p_first < f->g.lo ? 1 : 0), 0), n_tiles - 1);
    const int64_t tile1 = std::min(std::max<int64_t>((p_reach - f->g.lo) / TT_POS, tile0), n_tiles - 1);
    hipLaunchKernelGGL(fe_tally_tile_kernel, dim3((unsigned)(tile1 - tile0 + 1)), dim3(TT_THREADS), 0, f->stream, f->g, f->view(d), (uint32_t)d.n_reads, (const uint32_t *)f->d_span, tile0);
    return 0;
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
    {   // where the slab lies, for the grid of the tiled tally: first and last start, the longest reach (host arrays: one look at every alignment's last operation)
        int64_t p_first = reads[0].pos0, p_last = reads[0].pos0, reach = 0;
        bool sorted = true;
        for (int64_t i = 0; i < n_reads; ++i) {
            sorted &= reads[i].pos0 >= p_last;      // p_last is still the largest start among 0 .. i-1
            p_first = std::min(p_first, reads[i].pos0);
            p_last = std::max(p_last, reads[i].pos0);
            if (reads[i].n_ops && (int64_t)reads[i].op0 + reads[i].n_ops <= n_ops) {
                const clair_op_t &o = ops[reads[i].op0 + reads[i].n_ops - 1];
                reach = std::max(reach, (int64_t)o.ref_off + (int64_t)(o.code_len >> 2));
            }
        }
        // Starts that decrease inside a slab (a direct caller of this entry point; the packers raise the same bit themselves): the tallies stay
        // right -- the per-base kernel does not care about order -- and the caller hears about it like about every other thing the reference's
        // scripts would have treated differently (they stop at the first such line): CLAIR_FE_UNSORTED in the anomaly word.
        if (!sorted) hipLaunchKernelGGL(fe_flag_kernel, dim3(1), dim3(1), 0, f->stream, f->g, (uint32_t)CLAIR_FE_UNSORTED);
        if (launch_tally(f, d, p_first, p_last + reach, sorted)) return 1;
    }
    FE_TRY(f, hipGetLastError());
    // the caller's arrays may be reused as soon as this returns (the packer's slab is reset): wait for the copies
    FE_TRY(f, hipStreamSynchronize(f->stream));
    return 0;
}

int clair_frontend_text_options(clair_frontend_t *f, const char *ctg_name, int dcov, int evc_min_mq, int pile_min_mq, int64_t pile_start, int64_t pile_end) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (!ctg_name) return fe_fail(f, "contig name missing");
    if (!f->slabs.empty() && f->text_ready) return fe_fail(f, "text options cannot change once text was added");
    FE_TRY(f, hipSetDevice(f->device));
    const size_t n = strlen(ctg_name);
    (void)hipFree(f->d_ctg); f->d_ctg = nullptr;
    FE_TRY(f, hipMalloc((void **)&f->d_ctg, std::max<size_t>(n, 1)));
    if (n) FE_TRY(f, hipMemcpy(f->d_ctg, ctg_name, n, hipMemcpyHostToDevice));
    if (!f->d_text_state) FE_TRY(f, hipMalloc((void **)&f->d_text_state, sizeof(TextState)));
    const bool have_region = pile_start >= 0 && pile_end >= 0;
    f->text_opt = TextOptions{f->d_ctg, (int)n, dcov, evc_min_mq, pile_min_mq, have_region ? pile_start : -1, have_region ? pile_end : -1};
    f->text_state = TextState{};
    f->text_ready = true;
    return 0;
}

// returns 2 (not 1) when a line of the text is malformed: clair_host_sampack_feed on the same text names the line and the column
int clair_frontend_add_text(clair_frontend_t *f, const char *sam, int64_t len) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (!f->text_ready) return fe_fail(f, "call clair_frontend_text_options first");
    if (len < 0 || (len > 0 && !sam)) return fe_fail(f, "bad text");
    if (len == 0) return 0;
    if (len > 0x7ffffff0ll) return fe_fail(f, "text chunk of %lld bytes: at most 2 GB at a time", (long long)len);
    if (sam[len - 1] != '\n') return fe_fail(f, "the text must end at a line end");
    if (f->n_candidates >= 0) return fe_fail(f, "reads cannot be added after the candidates were fixed");
    FE_TRY(f, hipSetDevice(f->device));
    struct Temp {           // freed on every way out
        std::vector<void *> p;
        ~Temp() { for (void *x : p) (void)hipFree(x); }
        hipError_t get(void **out, size_t bytes) { hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16)); if (e == hipSuccess) p.push_back(*out); return e; }
    } tmp;
    uint8_t *d_text = nullptr;
    uint32_t *block_sum = nullptr, *d_count = nullptr;
    FE_TRY(f, tmp.get((void **)&d_text, (size_t)len + 16));            // fe_text_seq_kernel reads whole dwords
    FE_TRY(f, tmp.get((void **)&block_sum, ((size_t)blocks_for(len + 1, SCAN_BLOCK) + 2) * sizeof(uint32_t)));
    FE_TRY(f, tmp.get((void **)&d_count, 4 * sizeof(uint32_t)));
    FE_TRY(f, hipMemcpyAsync(d_text, sam, (size_t)len, hipMemcpyHostToDevice, f->stream));
    int64_t n_lines = 0;
    if (scan_count(f, d_text, len, block_sum, d_count, &n_lines, true)) return 1;
    int64_t *newline = nullptr, *cand = nullptr, *kept = nullptr;
    TextLine *lines = nullptr;
    uint8_t *is_cand = nullptr, *keep = nullptr;
    uint32_t *line_sum = nullptr;
    FE_TRY(f, tmp.get((void **)&newline, (size_t)n_lines * sizeof(int64_t)));
    if (scan_write(f, d_text, len, block_sum, d_count, nullptr, newline, 0, true)) return 1;
    FE_TRY(f, tmp.get((void **)&lines, (size_t)n_lines * sizeof(TextLine)));
    FE_TRY(f, tmp.get((void **)&is_cand, (size_t)n_lines + 1));
    FE_TRY(f, tmp.get((void **)&keep, (size_t)n_lines + 1));
    FE_TRY(f, tmp.get((void **)&line_sum, ((size_t)blocks_for(n_lines + 1, SCAN_BLOCK) + 2) * sizeof(uint32_t)));
    TextState carry = f->text_state;
    carry.malformed = 0xffffffffu;
    FE_TRY(f, hipMemcpyAsync(f->d_text_state, &carry, sizeof carry, hipMemcpyHostToDevice, f->stream));
    hipLaunchKernelGGL(fe_text_lines_kernel, dim3(blocks_for(n_lines, 256)), dim3(256), 0, f->stream, (const uint8_t *)d_text, (const int64_t *)newline, n_lines,
                       f->text_opt, lines, is_cand, f->d_text_state);
    FE_TRY(f, hipGetLastError());
    int64_t n_cand = 0, n_kept = 0;
    if (scan_count(f, is_cand, n_lines, line_sum, d_count + 1, &n_cand)) return 1;
    FE_TRY(f, tmp.get((void **)&cand, (size_t)n_cand * sizeof(int64_t)));
    if (scan_write(f, is_cand, n_lines, line_sum, d_count + 1, nullptr, cand, 0)) return 1;
    if (n_cand) hipLaunchKernelGGL(fe_text_dcov_kernel, dim3(blocks_for(n_cand, 256)), dim3(256), 0, f->stream, lines, (const int64_t *)cand, n_cand, f->text_opt.dcov, carry);
    hipLaunchKernelGGL(fe_text_keep_kernel, dim3(blocks_for(n_lines, 256)), dim3(256), 0, f->stream, (const TextLine *)lines, n_lines, keep);
    FE_TRY(f, hipGetLastError());
    if (scan_count(f, keep, n_lines, line_sum, d_count + 2, &n_kept)) return 1;
    FE_TRY(f, tmp.get((void **)&kept, (size_t)n_kept * sizeof(int64_t)));
    if (scan_write(f, keep, n_lines, line_sum, d_count + 2, nullptr, kept, 0)) return 1;
    uint32_t *op0 = nullptr, *elem0 = nullptr, *seq0 = nullptr;
    uint64_t *d_totals = nullptr;
    FE_TRY(f, tmp.get((void **)&op0, (size_t)n_kept * sizeof(uint32_t)));
    FE_TRY(f, tmp.get((void **)&elem0, (size_t)n_kept * sizeof(uint32_t)));
    FE_TRY(f, tmp.get((void **)&seq0, (size_t)n_kept * sizeof(uint32_t)));
    FE_TRY(f, tmp.get((void **)&d_totals, 5 * sizeof(uint64_t)));
    hipLaunchKernelGGL(fe_text_offsets_kernel, dim3(1), dim3(256), 0, f->stream, (const TextLine *)lines, n_lines, (const int64_t *)kept, n_kept, (const int64_t *)cand, n_cand,
                       op0, elem0, seq0, d_totals, carry, f->d_text_state);
    FE_TRY(f, hipGetLastError());
    uint64_t totals[5] = {0, 0, 0, 0, 0};
    TextState after{};
    FE_TRY(f, hipMemcpyAsync(totals, d_totals, sizeof totals, hipMemcpyDeviceToHost, f->stream));
    FE_TRY(f, hipMemcpyAsync(&after, f->d_text_state, sizeof after, hipMemcpyDeviceToHost, f->stream));
    FE_TRY(f, hipStreamSynchronize(f->stream));
    if (after.malformed != 0xffffffffu) {
        fe_fail(f, "line %u of this chunk (%lld lines before it) is not an alignment line the scripts accept: clair_host_sampack_feed on the same text names "
                   "the column", after.malformed - 1, (long long)f->text_state.lines);
        return 2;
    }
    if (totals[0] > 0xfffffff0ull || totals[1] > 0xfffffff0ull || totals[4] > 0xfffffff0ull) return fe_fail(f, "text chunk too dense for 32-bit offsets: feed smaller chunks");
    after.malformed = 0;
    f->text_state = after;
    if (n_kept == 0 || totals[0] == 0) return 0;
    Slab s;
    s.n_reads = n_kept; s.n_ops = (int64_t)totals[0]; s.n_elem = (int64_t)totals[1]; s.seq_bytes = (int64_t)totals[4];
    FE_TRY(f, hipMalloc((void **)&s.reads, (size_t)n_kept * sizeof(clair_read_t)));
    f->slabs.push_back(s);
    Slab &d = f->slabs.back();
    FE_TRY(f, hipMalloc((void **)&d.ops, (size_t)d.n_ops * sizeof(clair_op_t)));
    FE_TRY(f, hipMalloc((void **)&d.op_elem, ((size_t)d.n_ops + 1) * sizeof(uint32_t)));
    FE_TRY(f, hipMalloc((void **)&d.tuples, (size_t)n_kept * sizeof(uint64_t)));
    FE_TRY(f, hipMalloc((void **)&d.seq, (size_t)std::max<uint64_t>(totals[4], 16)));      // the bases only: the text itself is released when this call returns
    FE_TRY(f, hipMemsetAsync(d.tuples, 0, (size_t)n_kept * sizeof(uint64_t), f->stream));
    hipLaunchKernelGGL(fe_text_emit_kernel, dim3(blocks_for(n_kept, 256)), dim3(256), 0, f->stream, (const uint8_t *)d_text, (const TextLine *)lines, (const int64_t *)kept, n_kept,
                       (const uint32_t *)op0, (const uint32_t *)elem0, (const uint32_t *)seq0, d.reads, d.ops, d.op_elem, totals[0], totals[1]);
    hipLaunchKernelGGL(fe_text_seq_kernel, dim3(blocks_for(n_kept, 4)), dim3(256), 0, f->stream, (const uint8_t *)d_text, (const TextLine *)lines, (const int64_t *)kept, n_kept,
                       (const uint32_t *)seq0, d.seq);
    // the tiled tally bisects over the slab's start positions: once the packer has seen a start go backwards (this slab or an earlier
    // one: the word is cumulative, and a start below the previous slab's last is exactly such a case) the per-base kernel is used, as
    // clair_frontend_add_reads does -- an unsorted slab is flagged AND tallied correctly on both entry points
    if (launch_tally(f, d, (int64_t)totals[3], after.last_pos + (int64_t)totals[2], !(after.anomalies & CLAIR_FE_UNSORTED))) return 1;
    FE_TRY(f, hipGetLastError());
    FE_TRY(f, hipStreamSynchronize(f->stream));
    return 0;
}

int clair_frontend_text_stats(clair_frontend_t *f, int64_t *stats) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (!stats) return fe_fail(f, "stats is NULL");
    stats[0] = f->text_state.lines;
    stats[1] = f->text_state.evc_reads;
    stats[2] = f->text_state.pile_reads;
    stats[3] = (int64_t)f->text_state.anomalies;
    return 0;
}

int clair_frontend_slab_reads(clair_frontend_t *f, int64_t slab, struct clair_read *reads, int64_t capacity, int64_t *n_reads) {
    if (!f) return fe_fail(nullptr, "front end is NULL");
    if (slab < 0 || slab >= (int64_t)f->slabs.size()) return fe_fail(f, "slab %lld out of range [0, %lld)", (long long)slab, (long long)f->slabs.size());
    if (!n_reads) return fe_fail(f, "n_reads is NULL");
    const Slab &s = f->slabs[(size_t)slab];
    *n_reads = s.n_reads;
    if (!reads) return 0;
    if (capacity < s.n_reads) return fe_fail(f, "room for %lld alignments, the slab holds %lld", (long long)capacity, (long long)s.n_reads);
    FE_TRY(f, hipSetDevice(f->device));
    FE_TRY(f, hipMemcpy(reads, s.reads, (size_t)s.n_reads * sizeof(clair_read_t), hipMemcpyDeviceToHost));
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
    return clair_frontend_build_windows_ex(f, min_coverage, drop_non_iupac_centre, 1, n_windows);
}

int clair_frontend_build_windows_ex(clair_frontend_t *f, int min_coverage, int drop_non_iupac_centre, int consider_left_edge, int64_t *n_windows) {
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
    if (!consider_left_edge) {
        if (!f->d_late) {
            FE_TRY(f, hipMalloc((void **)&f->d_late, (size_t)room * N_POS * LATE_ROW * sizeof(uint32_t)));
            FE_TRY(f, hipMalloc((void **)&f->d_tuple_diff, ((size_t)room + 1) * sizeof(int)));
        }
        FE_TRY(f, hipMemsetAsync(f->d_late, 0, (size_t)room * N_POS * LATE_ROW * sizeof(uint32_t), f->stream));
        FE_TRY(f, hipMemsetAsync(f->d_tuple_diff, 0, ((size_t)room + 1) * sizeof(int), f->stream));
    }
    uint32_t *late = consider_left_edge ? nullptr : f->d_late;
    const bool per_op = consider_left_edge || !f->pass2_per_base;
    const int64_t m2 = f->g.n + 2;
    if (per_op && nc) {      // the running sum of the candidate prefix: what lets pass 2 go operation by operation
        const int64_t m = f->g.n + 1;
        const unsigned nb = blocks_for(m2, SCAN_BLOCK);
        if (!f->d_before_sum) {
            FE_TRY(f, hipMalloc((void **)&f->d_before_sum, (size_t)m * sizeof(unsigned long long)));
            FE_TRY(f, hipMalloc((void **)&f->d_prefix_block_sum, ((size_t)nb + 1) * sizeof(unsigned long long)));
        }
        hipLaunchKernelGGL(fe_prefix_block_sums_kernel, dim3(blocks_for(m, SCAN_BLOCK)), dim3(256), 0, f->stream, (const uint32_t *)f->d_before, m, f->d_prefix_block_sum);
        hipLaunchKernelGGL(fe_prefix_scan_sums_kernel, dim3(1), dim3(256), 0, f->stream, f->d_prefix_block_sum, (int64_t)blocks_for(m, SCAN_BLOCK));
        hipLaunchKernelGGL(fe_prefix_write_kernel, dim3(blocks_for(m, SCAN_BLOCK)), dim3(256), 0, f->stream, (const uint32_t *)f->d_before, m, (const unsigned long long *)f->d_prefix_block_sum, f->d_before_sum);
        if (!consider_left_edge) {
            if (!f->d_d2) {
                FE_TRY(f, hipMalloc((void **)&f->d_d2, (size_t)m2 * sizeof(unsigned long long)));
                FE_TRY(f, hipMalloc((void **)&f->d_fold, 2 * sizeof(unsigned long long)));
            }
            FE_TRY(f, hipMemsetAsync(f->d_d2, 0, (size_t)m2 * sizeof(unsigned long long), f->stream));
            FE_TRY(f, hipMemsetAsync(f->d_fold, 0, 2 * sizeof(unsigned long long), f->stream));
        }
    }
    Candidates c{f->d_centre, f->d_before, f->d_ins, f->d_before_sum, late, consider_left_edge ? nullptr : f->d_tuple_diff, f->d_d2, f->d_fold};
    for (Slab &s : f->slabs) {
        FE_TRY(f, hipMemsetAsync(s.tuples, 0, (size_t)s.n_reads * sizeof(uint64_t), f->stream));
        if (!s.n_elem || !nc) continue;
        if (consider_left_edge) hipLaunchKernelGGL(fe_windows_per_op_kernel, dim3(blocks_for(s.n_ops, 256)), dim3(256), 0, f->stream, f->g, f->view(s), c);
        else if (per_op) {
            hipLaunchKernelGGL(fe_windows_per_op_noleft_kernel, dim3(blocks_for(s.n_ops, 256)), dim3(256), 0, f->stream, f->g, f->view(s), c);
            hipLaunchKernelGGL(fe_late_starters_kernel, dim3(blocks_for(s.n_reads, 256)), dim3(256), 0, f->stream, f->g, f->view(s), c, (uint32_t)s.n_reads);
        } else hipLaunchKernelGGL(fe_windows_per_base_kernel, dim3(blocks_for(s.n_elem, 256)), dim3(256), 0, f->stream, f->g, f->view(s), c);
    }
    if (nc && late && per_op) {          // second differences -> first differences -> tuples per centre value -> per window
        const unsigned nb = blocks_for(m2, SCAN_BLOCK);
        for (int second = 0; second < 2; ++second) {
            hipLaunchKernelGGL(fe_tuple_scan_sums_kernel, dim3(nb), dim3(256), 0, f->stream, (const unsigned long long *)f->d_d2, m2, f->d_prefix_block_sum);
            hipLaunchKernelGGL(fe_prefix_scan_sums_kernel, dim3(1), dim3(256), 0, f->stream, f->d_prefix_block_sum, (int64_t)nb);
            hipLaunchKernelGGL(fe_tuple_scan_write_kernel, dim3(nb), dim3(256), 0, f->stream, f->d_d2, m2, (const unsigned long long *)f->d_prefix_block_sum,
                               (const unsigned long long *)f->d_fold, second, f->g.lo);
        }
        hipLaunchKernelGGL(fe_window_totals_at_kernel, dim3(blocks_for(nc, 256)), dim3(256), 0, f->stream, (const unsigned long long *)f->d_d2, (const int64_t *)f->d_centre, nc,
                           f->g.lo, f->g.n, f->d_window_tuples);
    }
    WindowRule rule{min_coverage, drop_non_iupac_centre, late};
    if (nc && late && !per_op) hipLaunchKernelGGL(fe_window_totals_kernel, dim3(1), dim3(256), 0, f->stream, (const int *)f->d_tuple_diff, nc, f->d_window_tuples);
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
                           (const int64_t *)f->d_kept, kept, (const uint32_t *)f->d_ins, (const uint32_t *)late, f->d_counts, f->d_out_centre, f->d_out_refseq);
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
    stats[0] = (int64_t)(bits | f->text_state.anomalies);
    stats[1] = (int64_t)f->slabs.size();
    stats[2] = reads;
    stats[3] = elems;
    stats[4] = f->n_candidates;
    stats[5] = f->n_windows;
    return 0;
}

}  // extern "C"
