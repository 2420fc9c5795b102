// Host-side helpers of the Clair hot path (include/clair_host.h): `samtools view` text -> packed alignments (include/clair_reads.h)
// for the device front end.  Plain C++17, no HIP.
//
// One pass over the text does what the reference's two stages each do per line before their base loops
// (dataPrepScripts/ExtractVariantCandidates.py:266-295, CreateTensor.py:251-287): split on whitespace, take FLAG, RNAME, POS, MAPQ,
// CIGAR, SEQ, apply the per-alignment filters of both stages -- kept as two flag bits, because the stages filter differently -- and
// reduce the CIGAR to the operations their loops act on.  The per-base work is the device's (clair_amd/csrc/frontend.hip).
// oracle/frontend_np.py: pack_sam is the Python twin this file is tested against (tests/test_frontend.py).
#include "../../include/clair_host.h"
#include "../../include/clair_reads.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int clair_host_fail(const char *fmt, ...);   // host_io.cpp

namespace {

constexpr int64_t LOOKAHEAD = 100000;        // CreateTensor.py:275

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

}  // namespace

struct clair_sampack {
    std::string ctg;
    int dcov = 250, evc_min_mq = 0, pile_min_mq = 0;
    bool have_region = false;
    int64_t region_start = 0, region_end = 0;     // 1-based inclusive: what `samtools view ctg:start-end` selects for the pileup
    // the slab being filled
    std::vector<clair_read_t> reads;
    std::vector<clair_op_t> ops;
    std::vector<uint32_t> op_elem;                // exclusive prefix of the operation lengths, ops.size() + 1 entries
    std::vector<uint8_t> seq;
    // across slabs
    int64_t prev_pos = 0, depth_cap = 0;          // CreateTensor.py:249-250, 277-287
    bool have_last = false;
    int64_t last_pos = 0;
    bool have_evc_last = false;                   // POS of the last alignment the candidate search accepted (CLAIR_FE_LEAD_INDEL)
    int64_t evc_last_pos = 0;
    int64_t lines_seen = 0, evc_reads = 0, pile_reads = 0, reads_total = 0;
    uint32_t anomalies = 0;

    clair_sampack() { op_elem.push_back(0); }

    void reset_slab() {
        reads.clear();
        ops.clear();
        op_elem.assign(1, 0);
        seq.clear();
    }

    int add_line(const char *p, const char *end, int64_t line_no) {
        // str.split(): columns 1..5 and 9 of a whitespace-separated line
        const char *col[10];
        size_t len[10];
        int n = 0;
        while (p < end && n < 10) {
            while (p < end && is_space((unsigned char)*p)) ++p;
            if (p >= end) break;
            const char *q = p;
            while (q < end && !is_space((unsigned char)*q)) ++q;
            col[n] = p;
            len[n] = (size_t)(q - p);
            ++n;
            p = q;
        }
        if (n == 0) return clair_host_fail("alignment line %lld is empty", (long long)line_no);
        if (col[0][0] == '@') return 0;
        if (n < 10) return clair_host_fail("alignment line %lld has %d columns (11 expected)", (long long)line_no, n);
        int64_t v[3];
        const int which[3] = {1, 3, 4};
        for (int i = 0; i < 3; ++i) {
            const char *s = col[which[i]], *e = s + len[which[i]];
            bool neg = false;
            if (s < e && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
            if (s == e || e - s > 18) return clair_host_fail("alignment line %lld: column %d is not an integer", (long long)line_no, which[i] + 1);
            int64_t x = 0;
            for (; s < e; ++s) {
                if (*s < '0' || *s > '9') return clair_host_fail("alignment line %lld: column %d is not an integer", (long long)line_no, which[i] + 1);
                x = x * 10 + (*s - '0');
            }
            v[i] = neg ? -x : x;
        }
        const int64_t flag = v[0], pos1 = v[1], mq = v[2], pos = pos1 - 1;
        const char *cigar = col[5];
        const size_t cl = len[5], sl = len[9];
        const bool same_ctg = len[2] == ctg.size() && memcmp(col[2], ctg.data(), ctg.size()) == 0;

        // the CIGAR, once: the operations, the aligned fraction of the candidate search (EVC :143-157), samtools' reference length
        const size_t op_first = ops.size();
        int64_t adv = 0, rp = 0, qp = 0, soft = 0, total = 0, rlen = 0;
        bool zero_indel = false, lead_indel = false;   // lead_indel: an I / D while the reference cursor is still at POS (tallied at POS - 1, EVC :326-336)
        uint64_t elems = op_elem.back();
        const uint32_t read_index = (uint32_t)reads.size();
        auto push = [&](uint32_t code) {
            if (adv > 0x3fffffff) adv = 0x3fffffff;   // absurd; the span check below sends the run to the host path
            ops.push_back(clair_op_t{read_index, (uint32_t)adv << 2 | code, (int32_t)rp, (uint32_t)qp});
            elems += (uint64_t)adv;
            op_elem.push_back((uint32_t)elems);
        };
        for (size_t i = 0; i < cl; ++i) {
            const char ch = cigar[i];
            if (ch >= '0' && ch <= '9') { adv = adv * 10 + (ch - '0'); if (adv > (int64_t)1 << 40) adv = (int64_t)1 << 40; continue; }
            switch (ch) {
            case 'S': soft += adv; qp += adv; break;
            case 'M': case '=': case 'X':
                if (adv) push(CLAIR_OP_M);
                rp += adv; qp += adv; rlen += adv;
                break;
            case 'I':
                if (adv) { push(CLAIR_OP_I); lead_indel |= rp == 0; } else zero_indel = true;
                qp += adv;
                break;
            case 'D':
                if (adv) { push(CLAIR_OP_D); lead_indel |= rp == 0; } else zero_indel = true;
                rp += adv; rlen += adv;
                break;
            case 'N': rlen += adv; break;
            default: break;
            }
            total += adv;
            adv = 0;
        }
        const bool evc_ok = same_ctg && mq >= evc_min_mq && !(cl == 1 && cigar[0] == '*') && !(1.0 - (double)soft / (double)(total + 1) < 0.55);
        bool in_region = true;
        if (have_region) {
            const int64_t end1 = pos1 + (rlen > 0 ? rlen : 1) - 1;    // bam_endpos
            in_region = same_ctg && pos1 <= region_end && end1 >= region_start;
        }
        bool pile_ok = false, flush = false;
        if (in_region && mq >= pile_min_mq) {
            if (prev_pos != pos) {
                prev_pos = pos;
                depth_cap = 0;
                pile_ok = true;
            } else {
                depth_cap += 1;
                pile_ok = depth_cap < dcov;
            }
            flush = pile_ok && depth_cap == 0;
        }
        ++reads_total;
        if (!evc_ok && !pile_ok) {   // neither stage looks at its bases
            ops.resize(op_first);
            op_elem.resize(op_first + 1);
            return 0;
        }
        if (have_last && pos < last_pos) anomalies |= CLAIR_FE_UNSORTED;
        have_last = true;
        last_pos = pos;
        if (zero_indel && evc_ok) anomalies |= CLAIR_FE_ZERO_INDEL;
        if (evc_ok) {   // the search flushes every position < POS after each alignment it accepts (EVC :345): a tally at POS - 1 by a later one stands alone
            if (lead_indel && have_evc_last && evc_last_pos == pos) anomalies |= CLAIR_FE_LEAD_INDEL;
            have_evc_last = true;
            evc_last_pos = pos;
        }
        if (rp > (int64_t)sl + LOOKAHEAD - 64 || rp > 0x7fffff00 || qp > 0x7fffff00) anomalies |= CLAIR_FE_LONG_SPAN;   // the last two: offsets beyond 32 bits
        if (elems > 0xfffffff0ull || seq.size() + sl > 0xfffffff0ull)
            return clair_host_fail("alignment line %lld: the slab is full (take it before feeding more)", (long long)line_no);
        clair_read_t r;
        r.pos0 = pos;
        r.seq0 = (uint32_t)seq.size();
        r.seq_len = (uint32_t)sl;
        r.op0 = (uint32_t)op_first;
        r.n_ops = (uint32_t)(ops.size() - op_first);
        r.flags = (flag & 16 ? CLAIR_READ_REVERSE : 0) | (evc_ok ? CLAIR_READ_EVC : 0) | (pile_ok ? CLAIR_READ_PILE : 0) | (flush ? CLAIR_READ_FLUSH : 0);
        r.reserved = 0;
        reads.push_back(r);
        const size_t at = seq.size();
        seq.resize(at + sl);
        const unsigned char *s = (const unsigned char *)col[9];
        uint8_t *d = seq.data() + at;
        for (size_t i = 0; i < sl; ++i) {
            const unsigned char c = s[i];
            d[i] = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
        }
        evc_reads += evc_ok;
        pile_reads += pile_ok;
        return 0;
    }
};

extern "C" {

int clair_host_sampack_create(const char *ctg_name, int dcov, int evc_min_mq, int pile_min_mq, int64_t pile_start, int64_t pile_end,
                              clair_sampack_t **out) {
    if (!out) return clair_host_fail("out is NULL");
    *out = nullptr;
    if (!ctg_name) return clair_host_fail("contig name missing");
    clair_sampack *p = new clair_sampack;
    p->ctg = ctg_name;
    p->dcov = dcov;
    p->evc_min_mq = evc_min_mq;
    p->pile_min_mq = pile_min_mq;
    p->have_region = pile_start >= 0 && pile_end >= 0;
    p->region_start = pile_start;
    p->region_end = pile_end;
    *out = p;
    return 0;
}

void clair_host_sampack_destroy(clair_sampack_t *p) { delete p; }

int clair_host_sampack_feed(clair_sampack_t *p, const char *sam, int64_t len, int final, int64_t *bytes_consumed) {
    if (!p || (!sam && len > 0) || !bytes_consumed) return clair_host_fail("bad argument");
    int64_t at = 0;
    while (at < len) {
        const char *nl = (const char *)memchr(sam + at, '\n', (size_t)(len - at));
        if (!nl && !final) break;
        const char *end = nl ? nl : sam + len;
        if (p->add_line(sam + at, end, p->lines_seen)) { *bytes_consumed = at; return 1; }
        ++p->lines_seen;
        at = (nl ? nl + 1 : end) - sam;
    }
    *bytes_consumed = at;
    return 0;
}

int clair_host_sampack_stats(const clair_sampack_t *p, int64_t *stats) {
    if (!p || !stats) return clair_host_fail("bad argument");
    stats[0] = (int64_t)p->reads.size();
    stats[1] = (int64_t)p->ops.size();
    stats[2] = (int64_t)p->op_elem.back();
    stats[3] = (int64_t)p->seq.size();
    stats[4] = (int64_t)p->anomalies;
    stats[5] = p->lines_seen;
    stats[6] = p->evc_reads;
    stats[7] = p->pile_reads;
    return 0;
}

int clair_host_sampack_slab(const clair_sampack_t *p, const clair_read_t **reads, const clair_op_t **ops, const uint32_t **op_elem,
                            const uint8_t **seq) {
    if (!p || !reads || !ops || !op_elem || !seq) return clair_host_fail("bad argument");
    *reads = p->reads.data();
    *ops = p->ops.data();
    *op_elem = p->op_elem.data();
    *seq = p->seq.data();
    return 0;
}

int clair_host_sampack_reset(clair_sampack_t *p) {
    if (!p) return clair_host_fail("bad argument");
    p->reset_slab();
    return 0;
}

/* The replay of CreateTensor.py's count of free tuple slots (:181, 283-289, 369-373) from what the device counted: reads in
 * stream order with their flags and tuple counts, candidate centres ascending with the tuples their windows held. */
int clair_host_tuple_budget_binds(const clair_read_t *reads, const uint64_t *tuples, int64_t n_reads, const int64_t *centres,
                                  const uint64_t *window_tuples, int64_t n_centres, int64_t *state, int *binds) {
    if ((n_reads > 0 && (!reads || !tuples)) || (n_centres > 0 && (!centres || !window_tuples)) || !state || !binds)
        return clair_host_fail("bad argument");
    // state[0] = free slots, state[1] = first centre not released yet; carried from slab to slab (initialise to {slots, 0})
    int64_t free_slots = state[0], ci = state[1];
    *binds = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        if (!(reads[r].flags & CLAIR_READ_PILE)) continue;
        free_slots -= (int64_t)tuples[r];
        if (free_slots < 1) { *binds = 1; break; }
        if (reads[r].flags & CLAIR_READ_FLUSH)
            while (ci < n_centres && centres[ci] + 17 < reads[r].pos0) free_slots += (int64_t)window_tuples[ci++];
    }
    state[0] = free_slots;
    state[1] = ci;
    return 0;
}

}  // extern "C"
