// Host-side helpers of the Clair hot path (include/clair_host.h): native pileup tensor generation.  Plain C++17, no HIP.
//
// Restates the main loop of the reference's dataPrepScripts/CreateTensor.py (OutputAlnTensor :251-388 and generate_tensor
// :29-65) as a streaming builder: SAM text in, finished [33][8][4] count windows out.  See clair_amd/create_tensor.py for
// the semantics in prose and for PileupBuilderPy, the Python twin this file is pinned against (tests/test_pileup.py); both
// are pinned against records minted from the real script (tests/golden/pileup_ct_*.json.gz).
//
// Differences in mechanism, not in result:
//   * the reference stores a (position, advance, ref base, read base, strand) tuple per window and read base and sums
//     them when the window is written; here the four counters are bumped while the read is walked (the sum does not
//     depend on order) and only the NUMBER of tuples is kept, for the reference's budget of outstanding tuples;
//   * begin_to_end (reference position -> windows that may open there) is a Python dict holding 34 entries per candidate
//     forever; for a sorted candidate list -- the normal case -- the same answer is the index range of loaded candidates
//     within [position-16, position+17], tracked with two cursors per read.  Unsorted lists take a hash-map path that
//     mirrors the dict.
#include "../../include/clair_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

int clair_host_fail(const char *fmt, ...);   // host_io.cpp

namespace {

constexpr int FLANK = 16;                 // shared/param.py:9
constexpr int N_POS = 2 * FLANK + 1;      // 33
constexpr int N_VAL = N_POS * 8 * 4;      // 1056
constexpr int64_t LOOKAHEAD = 100000;     // CreateTensor.py:275

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

// IUPAC_base_to_num_dict (shared/utils.py:24-27); -1: not a key
struct BaseTable {
    signed char num[256];
    BaseTable() {
        memset(num, -1, sizeof num);
        const char *keys = "ACGTURYSWKMBDHVN";
        const int vals[16] = {0, 1, 2, 3, 3, 0, 1, 1, 0, 2, 0, 1, 0, 0, 0, 0};
        for (int i = 0; i < 16; ++i) num[(unsigned char)keys[i]] = (signed char)vals[i];
    }
};
const BaseTable BASE;

struct Window {
    int64_t centre = 0;
    int64_t used = 0;        // tuples the reference would be holding for this window
    uint64_t order = 0;      // first-touch sequence number: the reference writes windows in dict insertion order
    int64_t depth_centre = 0;
    int seq_len = 0;         // set when the window is finished: the reference bases under it
    char seq[N_POS + 1];
    int32_t counts[N_VAL];
};

struct Active {
    int64_t centre;
    Window *win;
};

// CPython's set of ints, as far as its ITERATION ORDER goes (Objects/setobject.c of CPython 3.7 - 3.12: open addressing, 9 linear probes, then
// i = 5 i + 1 + perturb with perturb >>= 5; a removed key leaves a dummy that the next insertion along the same probe path takes over; the table
// is rebuilt at four times the live keys once active + dummy entries reach 3/5 of it; hash(i) = i for 0 <= i < 2^61 - 1).  The reference keeps
// the windows open over a read base in such a set and offers the base to them in the set's order (CreateTensor.py:296-310): the order shows
// in its output only where the tuple budget runs out in the middle of one base.  Pinned against the interpreter itself (tests/test_pileup.py:
// random add / remove histories) and against records minted from the real script with the budget binding (tests/golden/pileup_ct_budget_binds).
struct PySetOrder {
    static constexpr int64_t EMPTY = INT64_MIN, DUMMY = INT64_MIN + 1;
    std::vector<int64_t> table;
    size_t mask = 7, fill = 0, used = 0;
    PySetOrder() : table(8, EMPTY) {}
    void clear() { if (fill || mask != 7) { table.assign(8, EMPTY); mask = 7; fill = used = 0; } }
    static void insert_clean(std::vector<int64_t> &t, size_t mask, int64_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            const size_t probes = i + 9 <= mask ? 9 : 0;
            for (size_t j = i; j <= i + probes; ++j)
                if (t[j] == EMPTY) { t[j] = key; return; }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void resize(size_t minused) {
        size_t n = 8;
        while (n <= minused) n <<= 1;
        std::vector<int64_t> old;
        old.swap(table);
        table.assign(n, EMPTY);
        mask = n - 1;
        fill = used;
        for (int64_t k : old)
            if (k != EMPTY && k != DUMMY) insert_clean(table, mask, k);
    }
    void add(int64_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        int64_t *freeslot = nullptr;
        for (;;) {
            const size_t probes = i + 9 <= mask ? 9 : 0;
            for (size_t j = i; j <= i + probes; ++j) {
                int64_t &e = table[j];
                if (e == EMPTY) {
                    if (freeslot) { *freeslot = key; ++used; return; }
                    e = key; ++fill; ++used;
                    if (fill * 5 >= mask * 3) resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (e == key) return;
                if (e == DUMMY) freeslot = &e;
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void remove(int64_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            const size_t probes = i + 9 <= mask ? 9 : 0;
            for (size_t j = i; j <= i + probes; ++j) {
                if (table[j] == EMPTY) return;
                if (table[j] == key) { table[j] = DUMMY; --used; return; }
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
};

}  // namespace

struct clair_pileup {
    std::string ref;
    int64_t ref0 = 0;
    std::vector<int64_t> cands;
    bool sorted = true;
    bool left_edge = true;
    int dcov = 250, min_cov = 0, min_mq = 0;
    int64_t slots = 5000000;
    // candidate generator state (CreateTensor.py:68-109, 219, 274-275)
    size_t next_cand = 0;
    int64_t cand_pos = 0;
    // general path only: begin_to_end as the reference builds it
    std::unordered_map<int64_t, std::vector<int64_t>> begin;   // position -> centres, in load order
    // live windows by centre (ordered: a flush takes a prefix), pool behind them
    std::map<int64_t, Window *> live;
    std::vector<Window *> pool;
    uint64_t next_order = 0;
    int64_t prev_pos = 0;
    int64_t depth_cap = 0;
    std::vector<Window *> out;             // finished windows; [out_head, size) not taken yet (then back to the pool)
    size_t out_head = 0;
    std::vector<Active> active;            // per read
    int set_order = 0;                     // 0: a base is offered to the open windows in the order they were opened in; 1: in the order of CPython's set
    PySetOrder hashed;                     // set_order 1: the reference's active_set of this read
    std::vector<int64_t> offer;            // scratch
    std::vector<Window *> flushed;         // scratch
    int64_t reads_seen = 0;

    ~clair_pileup() {
        for (auto &kv : live) delete kv.second;
        for (Window *w : pool) delete w;
        for (size_t i = out_head; i < out.size(); ++i) delete out[i];
    }

    Window *new_window(int64_t centre) {
        Window *w;
        if (!pool.empty()) { w = pool.back(); pool.pop_back(); }
        else w = new Window;
        w->centre = centre;
        w->used = 0;
        w->depth_centre = 0;
        w->order = next_order++;
        memset(w->counts, 0, sizeof w->counts);
        return w;
    }

    void load_candidates(int64_t limit) {
        while (cand_pos != -1 && cand_pos < limit) {
            if (next_cand >= cands.size()) { cand_pos = -1; break; }
            const int64_t p = cands[next_cand++];
            if (!sorted) {
                if (left_edge) {
                    for (int64_t i = p - (FLANK + 1); i < p + (FLANK + 1); ++i) begin[i].push_back(p);
                } else {
                    auto &v = begin[p - (FLANK + 1)];
                    v.assign(1, p);
                }
            }
            cand_pos = p;
        }
    }

    // reference_sequence[reference_position - reference_start_0_based] with Python's negative-index wrap
    bool ref_base(int64_t rp, char *out) {
        int64_t i = rp - ref0;
        if (i < 0) i += (int64_t)ref.size();
        if (i < 0 || i >= (int64_t)ref.size()) {
            clair_host_fail("reference position %lld is outside the loaded reference sequence", (long long)(rp + 1));
            return false;
        }
        *out = ref[(size_t)i];
        return true;
    }

    void open_window(int64_t centre) {
        for (const Active &a : active)
            if (a.centre == centre) return;
        auto it = live.find(centre);
        Window *w;
        if (it == live.end()) {
            w = new_window(centre);
            live.emplace(centre, w);
        } else {
            w = it->second;
        }
        // keep `active` in ascending centre order (the sorted path appends in that order anyway)
        auto pos = active.end();
        if (!active.empty() && active.back().centre > centre)
            pos = std::lower_bound(active.begin(), active.end(), centre, [](const Active &a, int64_t c) { return a.centre < c; });
        active.insert(pos, Active{centre, w});
        if (set_order) hashed.add(centre);
    }

    // One tuple per open window, while the budget lasts (CreateTensor.py:306-310, 326-330, 343-347).  The order matters only to the base
    // the budget runs out on.
    template <class F> inline void offer_base(F &&one) {
        if (set_order && slots < (int64_t)active.size()) {
            offer.clear();
            for (int64_t k : hashed.table)
                if (k != PySetOrder::EMPTY && k != PySetOrder::DUMMY) offer.push_back(k);
            for (int64_t centre : offer) {
                if (slots <= 0) break;
                for (const Active &a : active)
                    if (a.centre == centre) { one(a.win); break; }
            }
            return;
        }
        for (const Active &a : active) {
            if (slots <= 0) break;
            one(a.win);
        }
    }

    void close_window_ending_at(int64_t rp) {
        const int64_t centre = rp - (FLANK + 1);
        for (size_t i = 0; i < active.size(); ++i)
            if (active[i].centre == centre) { active.erase(active.begin() + (long)i); if (set_order) hashed.remove(centre); return; }
    }

    // one tuple of the reference's alignment lists applied to one window (generate_tensor :34-56); rn / qn: row of the
    // reference / read base, -2 for the gap character, -1 for a character outside BASES (tuple kept, never counted)
    static inline int base_row(char b) { return b == '-' ? -2 : BASE.num[(unsigned char)b]; }
    inline void count(Window *w, int64_t rp, int64_t query_adv, int rn, int qn, int so) {
        w->used += 1;
        slots -= 1;
        if (rn == -1 || qn == -1) return;
        int64_t idx = rp - w->centre + (FLANK + 1);
        if (idx < 0 || idx >= N_POS) return;
        if (qn >= 0 && rn >= 0) {
            int32_t *r = w->counts + (idx * 8 + rn + so) * 4, *q = w->counts + (idx * 8 + qn + so) * 4;
            r[0] += 1;
            q[1] += 1;
            r[2] += 1;
            q[3] += 1;
            if (idx == FLANK) w->depth_centre += 1;
        } else if (qn >= 0) {
            idx = std::min<int64_t>(idx + query_adv, N_POS - 1);
            w->counts[(idx * 8 + qn + so) * 4 + 1] += 1;
        } else {
            w->counts[(idx * 8 + rn + so) * 4 + 2] += 1;
        }
    }

    // -> true: the window went to the output queue (which owns it now); false: dropped, the caller recycles it
    bool finish_window(Window *w) {
        const int64_t nrp = w->centre - ref0;
        if (nrp - (FLANK + 1) < 0 || w->depth_centre < min_cov) return false;
        // reference_sequence[nrp-17 : nrp+16]: a Python slice, clamped to the string
        const int64_t a = std::min<int64_t>(nrp - (FLANK + 1), (int64_t)ref.size());
        const int64_t b = std::min<int64_t>(nrp + FLANK, (int64_t)ref.size());
        w->seq_len = (int)std::max<int64_t>(0, b - a);
        memcpy(w->seq, ref.data() + a, (size_t)w->seq_len);
        w->seq[w->seq_len] = 0;
        out.push_back(w);
        return true;
    }

    void taken_up_to_head() {
        if (out_head == out.size()) { out.clear(); out_head = 0; }
    }

    int add_read(int flag, int64_t pos1, int64_t mapq, const char *cigar, size_t cigar_len, char *seq, size_t seq_len) {
        const int64_t pos = pos1 - 1;
        for (size_t i = 0; i < seq_len; ++i)
            if (seq[i] >= 'a' && seq[i] <= 'z') seq[i] = (char)(seq[i] - 32);
        const int so = (flag & 16) ? 4 : 0;
        if (mapq < min_mq) return 0;
        load_candidates(pos + (int64_t)seq_len + LOOKAHEAD);
        if (prev_pos != pos) {
            prev_pos = pos;
            depth_cap = 0;
        } else {
            depth_cap += 1;
            if (depth_cap >= dcov) return 0;
        }
        active.clear();
        if (set_order) hashed.clear();
        int64_t rp = pos, qp = 0, adv = 0;
        // sorted path: candidates [0, next_cand) are loaded; those within [rp-16, rp+17] may open at rp (left-edge mode), or
        // the one at rp+17 (otherwise).  `seen` = loaded candidates already offered to this read.
        size_t seen = 0;
        if (sorted) {
            const int64_t first = left_edge ? rp - FLANK : rp + (FLANK + 1);
            seen = (size_t)(std::lower_bound(cands.begin(), cands.begin() + (long)next_cand, first) - cands.begin());
        }
        auto open_at = [&](int64_t p) {
            if (sorted) {
                const int64_t last = p + (FLANK + 1);
                if (!left_edge)
                    while (seen < next_cand && cands[seen] < last) ++seen;   // passed over without being walked at p-17
                while (seen < next_cand && cands[seen] <= last) open_window(cands[seen++]);
            } else {
                auto it = begin.find(p);
                if (it != begin.end())
                    for (int64_t centre : it->second) open_window(centre);
            }
        };
        for (size_t ci = 0; ci < cigar_len; ++ci) {
            if (slots <= 0) break;
            const char ch = cigar[ci];
            if (ch >= '0' && ch <= '9') { adv = adv * 10 + (ch - '0'); continue; }
            if (ch == 'S') qp += adv;
            if (ch == 'M' || ch == '=' || ch == 'X') {
                for (int64_t k = 0; k < adv; ++k) {
                    open_at(rp);
                    if (!active.empty() && slots > 0) {
                        if (qp >= (int64_t)seq_len) return clair_host_fail("read at %lld: CIGAR walks past the end of SEQ (%zu bases)", (long long)pos1, seq_len);
                        char rb;
                        if (!ref_base(rp, &rb)) return 1;
                        const int rn = base_row(rb), qn = base_row(seq[qp]);
                        offer_base([&](Window *w) { count(w, rp, 0, rn, qn, so); });
                    }
                    if (!active.empty() && active.front().centre <= rp - (FLANK + 1)) close_window_ending_at(rp);
                    ++rp;
                    ++qp;
                }
            }
            if (ch == 'I') {
                for (int64_t k = 0; k < adv; ++k) {
                    if (!active.empty() && slots > 0) {
                        if (qp >= (int64_t)seq_len) return clair_host_fail("read at %lld: CIGAR walks past the end of SEQ (%zu bases)", (long long)pos1, seq_len);
                        const int qn = base_row(seq[qp]);
                        offer_base([&](Window *w) { count(w, rp, k, -2, qn, so); });
                    }
                    ++qp;
                }
            }
            if (ch == 'D') {
                for (int64_t k = 0; k < adv; ++k) {
                    if (!active.empty() && slots > 0) {
                        char rb;
                        if (!ref_base(rp, &rb)) return 1;
                        const int rn = base_row(rb);
                        offer_base([&](Window *w) { count(w, rp, 0, rn, -2, so); });
                    }
                    open_at(rp);
                    if (!active.empty() && active.front().centre <= rp - (FLANK + 1)) close_window_ending_at(rp);
                    ++rp;
                }
            }
            adv = 0;
        }
        if (depth_cap == 0) {   // a new start position: every window with centre + 17 < POS is complete
            flushed.clear();
            auto it = live.begin();
            while (it != live.end() && it->first + (FLANK + 1) < pos) {
                flushed.push_back(it->second);
                it = live.erase(it);
            }
            std::sort(flushed.begin(), flushed.end(), [](const Window *a, const Window *b) { return a->order < b->order; });
            for (Window *w : flushed) {
                slots += w->used;
                if (!finish_window(w)) pool.push_back(w);
            }
        }
        ++reads_seen;
        return 0;
    }

    int add_line(const char *p, const char *end, int64_t line_no) {
        // str.split(): columns 1, 3, 4, 5, 9 of a whitespace-separated line (CreateTensor.py:252-263)
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
        scratch_seq.assign(col[9], len[9]);
        return add_read((int)v[0], v[1], v[2], col[5], len[5], scratch_seq.data(), scratch_seq.size());
    }
    std::string scratch_seq;
    int64_t lines_seen = 0;
};

extern "C" {

int clair_host_pileup_create(const char *ref_seq, int64_t ref_len, int64_t reference_start_0_based, const int64_t *candidates,
                             int64_t n_candidates, int consider_left_edge, int dcov, int min_coverage, int min_mq,
                             int64_t available_slots, int force_general_path, clair_pileup_t **out) {
    if (!out) return clair_host_fail("out is NULL");
    *out = nullptr;
    if (!ref_seq || ref_len < 0) return clair_host_fail("reference sequence missing");
    if (n_candidates < 0 || (n_candidates > 0 && !candidates)) return clair_host_fail("candidate list missing");
    clair_pileup *p = new clair_pileup;
    p->ref.assign(ref_seq, (size_t)ref_len);
    p->ref0 = reference_start_0_based;
    p->cands.assign(candidates, candidates + n_candidates);
    p->sorted = !force_general_path && std::is_sorted(p->cands.begin(), p->cands.end());
    p->left_edge = consider_left_edge != 0;
    p->dcov = dcov;
    p->min_cov = min_coverage;
    p->min_mq = min_mq;
    p->slots = available_slots;
    *out = p;
    return 0;
}

void clair_host_pileup_destroy(clair_pileup_t *p) { delete p; }

int clair_host_pileup_set_order(clair_pileup_t *p, int cpython_set) {
    if (!p) return clair_host_fail("pileup builder is NULL");
    if (p->reads_seen) return clair_host_fail("the offer order is chosen before the first alignment");
    p->set_order = cpython_set ? 1 : 0;
    return 0;
}

// test hook: replay a history of set operations (key >= 0: add; -(key + 1): remove) and hand back the keys in iteration order
int clair_host_pyset_order(const int64_t *ops, int64_t n_ops, int64_t *keys, int64_t capacity, int64_t *n_keys) {
    if (!ops || !keys || !n_keys || n_ops < 0) return clair_host_fail("bad arguments to clair_host_pyset_order");
    PySetOrder s;
    for (int64_t i = 0; i < n_ops; ++i) {
        if (ops[i] >= 0) s.add(ops[i]); else s.remove(-(ops[i] + 1));
    }
    int64_t n = 0;
    for (int64_t k : s.table)
        if (k != PySetOrder::EMPTY && k != PySetOrder::DUMMY) {
            if (n >= capacity) return clair_host_fail("room for %lld keys", (long long)capacity);
            keys[n++] = k;
        }
    *n_keys = n;
    return 0;
}

int clair_host_pileup_feed(clair_pileup_t *p, const char *sam, int64_t len, int final, int64_t *bytes_consumed) {
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

int clair_host_pileup_finish(clair_pileup_t *p) {
    if (!p) return clair_host_fail("bad argument");
    // the reference's closing loop (CreateTensor.py:375-382): what is left, in first-touch order
    p->flushed.clear();
    for (auto &kv : p->live) p->flushed.push_back(kv.second);
    p->live.clear();
    std::sort(p->flushed.begin(), p->flushed.end(), [](const Window *a, const Window *b) { return a->order < b->order; });
    for (Window *w : p->flushed)
        if (!p->finish_window(w)) p->pool.push_back(w);
    return 0;
}

int64_t clair_host_pileup_pending(const clair_pileup_t *p) { return p ? (int64_t)(p->out.size() - p->out_head) : 0; }

int clair_host_pileup_take(clair_pileup_t *p, int64_t max_rows, int64_t *centres, char *refseq, int32_t *counts, int64_t *n_taken) {
    if (!p || !centres || !refseq || !counts || !n_taken || max_rows < 0) return clair_host_fail("bad argument");
    int64_t n = 0;
    while (n < max_rows && p->out_head < p->out.size()) {
        Window *w = p->out[p->out_head++];
        const Window &r = *w;
        centres[n] = r.centre;
        memset(refseq + n * (N_POS + 1), 0, N_POS + 1);
        memcpy(refseq + n * (N_POS + 1), r.seq, (size_t)r.seq_len);
        memcpy(counts + n * N_VAL, r.counts, sizeof r.counts);
        p->pool.push_back(w);
        ++n;
    }
    p->taken_up_to_head();
    *n_taken = n;
    return 0;
}

int clair_host_pileup_take_text(clair_pileup_t *p, const char *ctg_name, char *out, int64_t cap, int64_t *out_len, int64_t *n_taken) {
    if (!p || !ctg_name || !out || !out_len || !n_taken) return clair_host_fail("bad argument");
    const size_t ctg_len = strlen(ctg_name);
    // worst case per record: ctg + ' ' + 20 digits + ' ' + 33 + 1056 * (1 + 11) + '\n'
    const int64_t worst = (int64_t)ctg_len + 1 + 20 + 1 + N_POS + (int64_t)N_VAL * 12 + 1;
    int64_t at = 0, n = 0;
    while (p->out_head < p->out.size() && cap - at >= worst) {
        Window *w = p->out[p->out_head];
        const Window &r = *w;
        memcpy(out + at, ctg_name, ctg_len);
        at += (int64_t)ctg_len;
        at += snprintf(out + at, 24, " %lld ", (long long)r.centre);
        memcpy(out + at, r.seq, (size_t)r.seq_len);
        at += r.seq_len;
        for (int i = 0; i < N_VAL; ++i) {
            out[at++] = ' ';
            int32_t v = r.counts[i];
            if (v == 0) { out[at++] = '0'; continue; }
            char tmp[12];
            int k = 0;
            uint32_t u = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
            while (u) { tmp[k++] = (char)('0' + u % 10); u /= 10; }
            if (v < 0) out[at++] = '-';
            while (k) out[at++] = tmp[--k];
        }
        out[at++] = '\n';
        ++p->out_head;
        p->pool.push_back(w);
        ++n;
    }
    p->taken_up_to_head();
    *out_len = at;
    *n_taken = n;
    return 0;
}

int clair_host_pileup_stats(const clair_pileup_t *p, int64_t *stats) {
    if (!p || !stats) return clair_host_fail("bad argument");
    stats[0] = p->reads_seen;
    stats[1] = (int64_t)p->live.size();
    stats[2] = p->slots;
    stats[3] = p->sorted ? 1 : 0;
    return 0;
}

}  // extern "C"

// =====================================================================================================================
// Candidate extraction: the per-position base / indel tallies of dataPrepScripts/ExtractVariantCandidates.py
// (make_candidates :160-393) and its depth / allele-frequency filter, streaming.  Inference mode only: the training-set
// sampling switches (--gen4Training, --var_fn, --outputProb) draw from Python's random module and are not restated.
namespace {

struct Tally { int32_t n[7]; };   // A C G T I D N, the reference's dict order (:265) -- ties in the sort below keep it

const char TALLY_NAME[7] = {'A', 'C', 'G', 'T', 'I', 'D', 'N'};

// IUPAC_base_to_ACGT_base_dict (shared/utils.py:19-22) through evc_base_from (:27-28): index into TALLY_NAME, -1 = KeyError
struct AcgtTable {
    signed char idx[256];
    AcgtTable() {
        memset(idx, -1, sizeof idx);
        const char *keys = "ACGTURYSWKMBDHVN";
        const char *vals = "ACGTTACCAGACAAAA";
        for (int i = 0; i < 16; ++i) {
            const char v = vals[i];
            idx[(unsigned char)keys[i]] = (signed char)(v == 'A' ? 0 : v == 'C' ? 1 : v == 'G' ? 2 : 3);
        }
        idx[(unsigned char)'N'] = 6;   // evc_base_from keeps N
    }
};
const AcgtTable ACGT;

}  // namespace

struct clair_evc {
    std::string ctg, ref;
    int64_t ref0 = 0;
    bool have_range = false, have_bed = false;
    int64_t ctg_start = 0, ctg_end = 0;
    std::vector<int64_t> bed_start, bed_end;   // sorted, merged, 0-based half-open
    double min_depth = 4, min_af = 0.125;
    int min_mq = 0;
    // pileup[position] of the reference (:265) as a dense window over the positions still open: base .. base + size
    std::deque<Tally> window;
    int64_t base = 0;
    int64_t reads = 0, lines_seen = 0;

    inline Tally &at(int64_t p) {
        if (window.empty()) base = p;
        if (p < base) {   // unsorted input only
            window.insert(window.begin(), (size_t)(base - p), Tally{{0, 0, 0, 0, 0, 0, 0}});
            base = p;
        }
        const int64_t i = p - base;
        if (i >= (int64_t)window.size()) window.resize((size_t)i + 1, Tally{{0, 0, 0, 0, 0, 0, 0}});
        return window[(size_t)i];
    }
    struct Cand { int64_t pos1; char ref_base; int64_t depth; signed char order[7]; int32_t n[7]; };
    std::deque<Cand> out;

    bool in_bed(int64_t p0) const {
        auto it = std::upper_bound(bed_start.begin(), bed_start.end(), p0);
        if (it == bed_start.begin()) return false;
        return p0 < bed_end[(size_t)(it - bed_start.begin()) - 1];
    }

    void flush(int64_t before, bool all) {
        while (!window.empty() && (all || base < before)) {
            const Tally &t = window.front();
            if (t.n[0] | t.n[1] | t.n[2] | t.n[3] | t.n[4] | t.n[5] | t.n[6]) emit(base, t);   // a key of the reference's dict
            window.pop_front();
            ++base;
        }
    }

    void emit(int64_t p0, const Tally &t) {
        if (have_range && !(ctg_start <= p0 + 1 && p0 + 1 <= ctg_end)) return;
        if (have_bed && !in_bed(p0)) return;
        int64_t i = p0 - ref0;
        if (i < 0) i += (int64_t)ref.size();           // Python str indexing
        if (i < 0 || i >= (int64_t)ref.size()) return;   // IndexError -> except: continue (:349-355)
        const int rb = ACGT.idx[(unsigned char)ref[(size_t)i]];
        if (rb < 0) return;                              // KeyError -> continue
        int64_t depth = 0;
        for (int k = 0; k < 7; ++k) depth += t.n[k];
        depth -= t.n[4] + t.n[5];
        if ((double)depth < min_depth) return;
        const int64_t denom = depth > 0 ? depth : 1;
        Cand c;
        for (int k = 0; k < 7; ++k) { c.order[k] = (signed char)k; c.n[k] = t.n[k]; }
        std::stable_sort(c.order, c.order + 7, [&](signed char a, signed char b) { return t.n[a] > t.n[b]; });
        const bool pass = c.order[0] != rb || ((double)t.n[c.order[1]] / (double)denom) >= min_af;
        if (!pass) return;
        c.pos1 = p0 + 1;
        c.ref_base = TALLY_NAME[rb];
        c.depth = depth;
        out.push_back(c);
    }

    int add_line(const char *p, const char *end, int64_t line_no) {
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
        if (len[2] != ctg.size() || memcmp(col[2], ctg.data(), ctg.size()) != 0) return 0;   // RNAME != ctgName (:279-281)
        int64_t v[2];
        const int which[2] = {3, 4};
        for (int i = 0; i < 2; ++i) {
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
        const int64_t pos = v[0] - 1;
        if (v[1] < min_mq) return 0;
        const char *cigar = col[5];
        const size_t cl = len[5];
        if (cl == 1 && cigar[0] == '*') return 0;
        {   // a read less than 55 % aligned is skipped (:143-157)
            int64_t soft = 0, total = 0, adv = 0;
            for (size_t i = 0; i < cl; ++i) {
                const char ch = cigar[i];
                if (ch >= '0' && ch <= '9') { adv = adv * 10 + (ch - '0'); continue; }
                if (ch == 'S') soft += adv;
                total += adv;
                adv = 0;
            }
            if (1.0 - (double)soft / (double)(total + 1) < 0.55) return 0;
        }
        ++reads;
        // positions before POS - 1 cannot change any more: write them now rather than zero-filling across a coverage gap
        // (POS - 1 itself can still receive this read's leading insertion / deletion, :307-313)
        if (!window.empty() && pos - 1 > base + (int64_t)window.size()) flush(pos - 1, false);
        const char *seq = col[9];
        const int64_t sl = (int64_t)len[9];
        int64_t rp = pos, qp = 0, adv = 0;
        for (size_t i = 0; i < cl; ++i) {
            const char ch = cigar[i];
            if (ch >= '0' && ch <= '9') { adv = adv * 10 + (ch - '0'); continue; }
            if (ch == 'S') {
                qp += adv;
            } else if (ch == 'M' || ch == '=' || ch == 'X') {
                for (int64_t k = 0; k < adv; ++k) {
                    if (qp >= sl) return clair_host_fail("read at %lld: CIGAR walks past the end of SEQ (%lld bases)", (long long)v[0], (long long)sl);
                    unsigned char b = (unsigned char)seq[qp];
                    if (b >= 'a' && b <= 'z') b = (unsigned char)(b - 32);
                    const int bi = ACGT.idx[b];
                    if (bi < 0) return clair_host_fail("read at %lld: SEQ holds '%c', not an IUPAC base code", (long long)v[0], (char)b);
                    at(rp).n[bi] += 1;
                    ++rp;
                    ++qp;
                }
            } else if (ch == 'I') {
                at(rp - 1).n[4] += 1;
                qp += adv;
            } else if (ch == 'D') {
                at(rp - 1).n[5] += 1;
                rp += adv;
            }
            adv = 0;
        }
        flush(pos, false);   // positions before this read's start are complete (:319)
        return 0;
    }
};

extern "C" {

int clair_host_evc_create(const char *ctg_name, const char *ref_seq, int64_t ref_len, int64_t reference_start_0_based,
                          int64_t ctg_start, int64_t ctg_end, const int64_t *bed_start, const int64_t *bed_end, int64_t n_bed,
                          double min_coverage, double threshold, int min_mq, clair_evc_t **out) {
    if (!out) return clair_host_fail("out is NULL");
    *out = nullptr;
    if (!ctg_name || !ref_seq || ref_len < 0) return clair_host_fail("contig name / reference sequence missing");
    if (n_bed > 0 && (!bed_start || !bed_end)) return clair_host_fail("bed intervals missing");
    clair_evc *e = new clair_evc;
    e->ctg = ctg_name;
    e->ref.assign(ref_seq, (size_t)ref_len);
    e->ref0 = reference_start_0_based;
    e->have_range = ctg_start >= 0 && ctg_end >= 0;
    e->ctg_start = ctg_start;
    e->ctg_end = ctg_end;
    e->have_bed = n_bed >= 0;
    if (n_bed > 0) {
        // membership only (interval_tree.at(p), shared/interval_tree.py:45-57): sort and merge
        std::vector<std::pair<int64_t, int64_t>> iv;
        for (int64_t i = 0; i < n_bed; ++i) iv.emplace_back(bed_start[i], bed_end[i] == bed_start[i] ? bed_end[i] + 1 : bed_end[i]);
        std::sort(iv.begin(), iv.end());
        for (auto &x : iv) {
            if (x.second <= x.first) continue;
            if (!e->bed_start.empty() && x.first <= e->bed_end.back()) e->bed_end.back() = std::max(e->bed_end.back(), x.second);
            else { e->bed_start.push_back(x.first); e->bed_end.push_back(x.second); }
        }
    }
    e->min_depth = min_coverage;
    e->min_af = threshold;
    e->min_mq = min_mq;
    *out = e;
    return 0;
}

void clair_host_evc_destroy(clair_evc_t *e) { delete e; }

int clair_host_evc_feed(clair_evc_t *e, const char *sam, int64_t len, int final, int64_t *bytes_consumed) {
    if (!e || (!sam && len > 0) || !bytes_consumed) return clair_host_fail("bad argument");
    int64_t at = 0;
    while (at < len) {
        const char *nl = (const char *)memchr(sam + at, '\n', (size_t)(len - at));
        if (!nl && !final) break;
        const char *end = nl ? nl : sam + len;
        if (e->add_line(sam + at, end, e->lines_seen)) { *bytes_consumed = at; return 1; }
        ++e->lines_seen;
        at = (nl ? nl + 1 : end) - sam;
    }
    *bytes_consumed = at;
    return 0;
}

int clair_host_evc_finish(clair_evc_t *e) {
    if (!e) return clair_host_fail("bad argument");
    e->flush(0, true);
    return 0;
}

int64_t clair_host_evc_pending(const clair_evc_t *e) { return e ? (int64_t)e->out.size() : 0; }
int64_t clair_host_evc_reads(const clair_evc_t *e) { return e ? e->reads : 0; }

int clair_host_evc_take(clair_evc_t *e, int64_t max_rows, int64_t *positions, int64_t *n_taken) {
    if (!e || !positions || !n_taken || max_rows < 0) return clair_host_fail("bad argument");
    int64_t n = 0;
    while (n < max_rows && !e->out.empty()) {
        positions[n++] = e->out.front().pos1;
        e->out.pop_front();
    }
    *n_taken = n;
    return 0;
}

int clair_host_evc_take_text(clair_evc_t *e, char *out, int64_t cap, int64_t *out_len, int64_t *n_taken) {
    if (!e || !out || !out_len || !n_taken) return clair_host_fail("bad argument");
    const int64_t worst = (int64_t)e->ctg.size() + 64 + 7 * 16;
    int64_t at = 0, n = 0;
    while (!e->out.empty() && cap - at >= worst) {
        const clair_evc::Cand &c = e->out.front();
        memcpy(out + at, e->ctg.data(), e->ctg.size());
        at += (int64_t)e->ctg.size();
        at += snprintf(out + at, 64, " %lld %c %lld", (long long)c.pos1, c.ref_base, (long long)c.depth);
        for (int k = 0; k < 7; ++k) at += snprintf(out + at, 16, " %c %d", TALLY_NAME[(int)c.order[k]], c.n[(int)c.order[k]]);
        out[at++] = '\n';
        e->out.pop_front();
        ++n;
    }
    *out_len = at;
    *n_taken = n;
    return 0;
}

}  // extern "C"
