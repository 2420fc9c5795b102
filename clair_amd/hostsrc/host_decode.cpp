// Native VCF decode (include/clair_host.h: clair_host_decode_rows): probabilities -> VCF rows, the work of
// clair/call_var.py:589-1236 (possible_outcome_probabilites_from, output_from, output_with, batch_output), restated from
// the build's Python decoder clair_amd/call_var.py (VariantDecoder / OutcomeFamilies / _IndelResolver), which is pinned byte
// for byte against fixtures minted from the real reference.  Scope: the configuration the GPU pipeline runs in -- no BAM
// look-ups available (every look-up answers "", the reference's own fall-back, :520-524, :562-564), no --debug, no
// --output_for_ensemble.  Everything else stays on the Python path.
//
// Bit-exactness rules: all float32 products are formed in the reference's operand order with plain float multiplies (the
// file is compiled with -ffp-contract=off; x86-64 SSE arithmetic is IEEE float32); 8-element float32 sums follow NumPy's
// pairwise scheme ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)); equality tests are exact; QUAL / AF follow either arithmetic mode
// of the Python decoder ("legacy" = float64 as under the NumPy 1.18 the reference pins, "numpy2" = float32 scalars).
#include "../../include/clair_host.h"
#include "../../include/clair_call.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

int clair_host_fail(const char *fmt, ...);   // host_io.cpp
extern "C" int clair_host_threads(int work_items);
void clair_host_parallel(int nthreads, const std::function<void(int)> &work);   // host_io.cpp: persistent pool

namespace {

constexpr int CENTER = 16, NEXT = 17, LONG_INDEL = 16;
constexpr int CH_REF = 0, CH_INS = 1, CH_DEL = 2, CH_SNP = 3;
enum { F_REF = CLAIR_F_REF, F_HOMO_SNP, F_HET_SNP, F_HOMO_INS, F_ACGT_INS, F_INSINS, F_HOMO_DEL, F_ACGT_DEL, F_DELDEL, F_INSDEL, N_FAM };
static_assert((int)N_FAM == (int)CLAIR_F_COUNT && (int)F_INSDEL == (int)CLAIR_F_INSDEL && sizeof(clair_call_t) == 32, "families and record as declared in clair_call.h");
const int FAM_SIZE[N_FAM] = {1, 4, 6, 16, 64, 256, 16, 64, 240, 512};
// gt21 labels (task/gt21.py:3-50): AA AC AG AT CC CG CT GG GT TT DelDel ADel CDel GDel TDel InsIns AIns CIns GIns TIns InsDel
const char *const GT21[21] = {"AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT", "DelDel", "ADel", "CDel", "GDel", "TDel",
                              "InsIns", "AIns", "CIns", "GIns", "TIns", "InsDel"};
const int HOMO_SNP_IDX[4] = {0, 4, 7, 9};
const int HET_SNP_IDX[6] = {1, 2, 3, 5, 6, 8};
const int INS_BASE_IDX[4] = {16, 17, 18, 19}, DEL_BASE_IDX[4] = {11, 12, 13, 14};
constexpr int IDX_DELDEL = 10, IDX_INSINS = 15, IDX_INSDEL = 20;
const char *const GT_STR[4] = {"0/0", "1/1", "0/1", "1/2"};

inline float xat(const float *x, int pos, int row, int ch) { return x[(pos * 8 + row) * 4 + ch]; }
inline float sum8(const float a[8]) { return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])); }
inline float sum_rows(const float *x, int pos, int ch) {
    float a[8];
    for (int r = 0; r < 8; ++r) a[r] = xat(x, pos, r, ch);
    return sum8(a);
}
inline int iupac_num(char c) {   // shared/utils.py:19-23
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'U': return 3; case 'R': return 0;
        case 'Y': return 1; case 'S': return 1; case 'W': return 0; case 'K': return 2; case 'M': return 0; case 'B': return 1;
        case 'D': return 0; case 'H': return 0; case 'V': return 0; case 'N': return 0; default: return -1;
    }
}
inline char iupac_acgt(char c) { return "ACGT"[iupac_num(c)]; }   // shared/utils.py:24-27 (same table, as letters)
inline int gt21_index(const std::string &label) {
    for (int i = 0; i < 21; ++i)
        if (label == GT21[i]) return i;
    return -1;
}

struct Families {
    float v[N_FAM][512];
    bool alive[N_FAM][512];
};

// call_var.py:430-437, 466-472: per-base insertion evidence, entries 4..7 zero; argmax is the first maximum
inline void insertion_votes(const float *x, int pos, float votes[8]) {
    for (int b = 0; b < 4; ++b)
        votes[b] = (xat(x, pos, b, CH_INS) + xat(x, pos, b + 4, CH_INS)) - (xat(x, pos, b, CH_SNP) + xat(x, pos, b + 4, CH_SNP));
    votes[4] = votes[5] = votes[6] = votes[7] = 0.0f;
}
// deletion_bases_from (no BAM): the reference sequence after the centre (call_var.py:527-565)
inline std::string deletion_bases(const char *seq, int seq_len, int length) {
    const int a = NEXT < seq_len ? NEXT : seq_len, b = NEXT + length < seq_len ? NEXT + length : seq_len;
    return std::string(seq + a, (size_t)(b > a ? b - a : 0));
}

// task/gt21.py:60-110
inline std::string allele_kind(const std::string &ref, const std::string &alt) {
    if (ref.size() > alt.size()) return "Del";
    if (ref.size() < alt.size()) return "Ins";
    return std::string(1, alt[0]);
}
int gt21_index_of_call(const std::string &ref, const std::string &alt, int g1, int g2) {
    std::string a0, a1;
    const size_t comma = alt.find(',');
    if (comma == std::string::npos) { a0 = (g1 == 0 || g2 == 0) ? ref : alt; a1 = alt; }
    else { a0 = alt.substr(0, comma); a1 = alt.substr(comma + 1); const size_t c2 = a1.find(','); if (c2 != std::string::npos) a1 = a1.substr(0, c2); }
    const std::string a = allele_kind(ref, a0), b = allele_kind(ref, a1);
    std::string label;
    if (a.size() == 1 && b.size() == 1) label = a <= b ? a + b : b + a;
    else if (a.size() == 1 || b.size() == 1) label = a.size() == 1 ? a + b : b + a;
    else if (a == b) label = a + b;
    else label = "InsDel";
    return gt21_index(label);
}

struct Config { int show_ref, haploid_precision, haploid_sensitive, has_qual, qual; int numpy2; };

// ---- allele kinds and the gt21 class of a call (task/gt21.py:60-110), on codes instead of strings: 0..3 = the base, 4 = Ins, 5 = Del,
//      254 = a character outside ACGT (the reference's label look-up fails there)
constexpr int K_INS = 4, K_DEL = 5, K_BAD = 254;
inline int base_kind(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : K_BAD; }
inline int class_of_kinds(int a, int b) {
    static const int PAIR[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};   // AA AC AG AT / CC CG CT / GG GT / TT
    if (a == K_BAD || b == K_BAD) return 255;
    if (a < 4 && b < 4) return PAIR[a][b];
    if (a < 4 || b < 4) { const int base = a < 4 ? a : b, other = a < 4 ? b : a; return (other == K_INS ? 16 : 11) + base; }
    if (a == b) return a == K_INS ? IDX_INSINS : IDX_DELDEL;
    return IDX_INSDEL;
}

// bases the window supports for an insertion of LONG_INDEL or more (call_var.py:487-497): positions 17..31 always, 32 under the
// read-support condition
inline int long_insertion_bases(const float *x) {
    float votes[8];
    int n = 0;
    for (int p = NEXT; p <= 2 * CENTER; ++p) {
        insertion_votes(x, p, votes);
        if (p < CENTER + LONG_INDEL || (double)sum8(votes) >= 0.125 * (double)sum_rows(x, p, CH_REF)) ++n;
        else break;
    }
    return n;
}
inline int deletion_length(int seq_len, int length) {   // characters deletion_bases() would return
    const int a = NEXT < seq_len ? NEXT : seq_len, b = NEXT + length < seq_len ? NEXT + length : seq_len;
    return b > a ? b - a : 0;
}

// ---- resolve: probabilities + window -> the call record (include/clair_call.h).  possible_outcome_probabilites_from (:589-690),
//      output_from (:693-947) and the numeric half of output_with (:1002-1166).  Integer logic throughout: whether an outcome can be
//      written down depends only on LENGTHS (an insertion always finds bases in the window; a deletion needs reference characters
//      behind the centre; an Ins/Ins pair needs its two alleles to differ), never on the strings themselves.
void resolve_one(const float *x, const float *g, const float *z, const float *l1, const float *l2, char ref0, int seq_len, Families &fam,
                 clair_call_t &c) {
    memset(&c, 0, sizeof c);
    c.alt_b0 = c.alt_b1 = 255;
    c.gt = c.gi = 255;
    if (!(ref0 == 'A' || ref0 == 'C' || ref0 == 'G' || ref0 == 'T' || ref0 == 'U')) return;   // call_var.py:1018
    float dsum[8];
    for (int r = 0; r < 8; ++r) dsum[r] = xat(x, CENTER, r, CH_DEL) + xat(x, CENTER, r, CH_REF);
    const float depth = sum8(dsum);                                                              // :1022-1024
    if (depth == 0.0f) return;
    c.depth = depth;
    c.status = CLAIR_CALL_RESOLVED;
    {   // the tensor's vote on inserted bases (:428-447), all sixteen positions
        float votes[8];
        for (int k = 0; k < 16; ++k) {
            insertion_votes(x, NEXT + k, votes);
            int best = 0;
            for (int i = 1; i < 8; ++i)
                if (votes[i] > votes[best]) best = i;
            c.ins_code |= (uint32_t)(best % 4) << (2 * k);
        }
        c.ins_avail = (uint8_t)long_insertion_bases(x);
    }
    // ---- the ten outcome families, products left to right as written in possible_outcome_probabilites_from (:589-690) ----
    const float p_ref = z[0], p_hom = z[1], p_het = z[2];
    const float z1 = l1[16], z2 = l2[16], zero = z1 * z2;
    const int ref_num = iupac_num(ref0);
    const int ref_class = HOMO_SNP_IDX[ref_num];
    fam.v[F_REF][0] = (zero * p_ref) * g[ref_class];
    for (int k = 0; k < 4; ++k) fam.v[F_HOMO_SNP][k] = (zero * p_hom) * g[HOMO_SNP_IDX[k]];
    for (int k = 0; k < 6; ++k) fam.v[F_HET_SNP][k] = (zero * p_het) * g[HET_SNP_IDX[k]];
    float ins1[16], ins2[16], del1[16], del2[16];
    for (int i = 0; i < 16; ++i) { ins1[i] = l1[17 + i]; ins2[i] = l2[17 + i]; del1[i] = l1[15 - i]; del2[i] = l2[15 - i]; }
    const float e_homins = p_hom * g[IDX_INSINS], e_insins = p_het * g[IDX_INSINS];
    const float e_homdel = p_hom * g[IDX_DELDEL], e_deldel = p_het * g[IDX_DELDEL], e_insdel = p_het * g[IDX_INSDEL];
    for (int i = 0; i < 16; ++i) {
        fam.v[F_HOMO_INS][i] = (ins1[i] * ins2[i]) * e_homins;
        fam.v[F_HOMO_DEL][i] = (del1[i] * del2[i]) * e_homdel;
        const float a = z1 * ins2[i], b = ins1[i] * z2, one_ins = a > b ? a : b;   // np.maximum
        const float cc = z1 * del2[i], d = del1[i] * z2, one_del = cc > d ? cc : d;
        for (int k = 0; k < 4; ++k) {
            fam.v[F_ACGT_INS][i * 4 + k] = (one_ins * g[INS_BASE_IDX[k]]) * p_het;
            fam.v[F_ACGT_DEL][i * 4 + k] = (one_del * g[DEL_BASE_IDX[k]]) * p_het;
        }
        int dd = 0;
        for (int j = 0; j < 16; ++j) {
            fam.v[F_INSINS][i * 16 + j] = (ins1[i] * ins2[j]) * e_insins;
            fam.v[F_INSDEL][(i * 16 + j) * 2 + 0] = (ins1[i] * del2[j]) * e_insdel;
            fam.v[F_INSDEL][(i * 16 + j) * 2 + 1] = (del1[i] * ins2[j]) * e_insdel;
            if (j != i) { fam.v[F_DELDEL][i * 15 + dd] = (del1[i] * del2[j]) * e_deldel; ++dd; }
        }
    }
    for (int k = 0; k < N_FAM; ++k)
        for (int i = 0; i < FAM_SIZE[k]; ++i) fam.alive[k][i] = true;

    // ---- iterative arg-max with exact-equality membership (output_from, :693-947) ----
    unsigned flags = 0;
    int family = F_REF, index = 0;
    bool consulted = false, multi = false, same = false;
    for (;;) {
        ++c.rounds;
        float tops[N_FAM];
        for (int k = 0; k < N_FAM; ++k) {
            bool any = false;
            float m = 0.0f;                              // `max(...) if len(...) else 0`
            for (int i = 0; i < FAM_SIZE[k]; ++i)
                if (fam.alive[k][i] && (!any || fam.v[k][i] > m)) { m = fam.v[k][i]; any = true; }
            tops[k] = m;
        }
        float best = tops[0];
        for (int k = 1; k < N_FAM; ++k)
            if (tops[k] > best) best = tops[k];
        if (best == tops[F_REF]) { flags = 1u << F_REF; family = F_REF; index = 0; break; }
        flags = 0;
        int first[N_FAM];
        family = -1;
        for (int k = 1; k < N_FAM; ++k) {
            first[k] = -1;
            for (int i = 0; i < FAM_SIZE[k]; ++i)
                if (fam.alive[k][i] && fam.v[k][i] == best) { first[k] = i; break; }
            if (first[k] >= 0) { flags |= 1u << k; if (family < 0) family = k; }
        }
        if (family < 0) {   // nothing equals the best: a NaN among the probabilities (the reference's max() is undefined there): no call
            memset(&c, 0, sizeof c);
            c.alt_b0 = c.alt_b1 = c.gt = c.gi = 255;
            return;
        }
        bool have = false;
        index = first[family];
        c.alt_b0 = c.alt_b1 = 255;
        multi = false;
        switch (family) {
            case F_HOMO_SNP: {   // call_var.py:60-62
                int bi = 0;
                for (int k = 1; k < 4; ++k)
                    if (g[HOMO_SNP_IDX[k]] > g[HOMO_SNP_IDX[bi]]) bi = k;
                c.alt_b0 = (uint8_t)bi;
                same = "ACGT"[bi] == ref0;
                have = true;
                break;
            }
            case F_HET_SNP: {    // call_var.py:65-67
                int bi = 0;
                for (int k = 1; k < 6; ++k)
                    if (g[HET_SNP_IDX[k]] > g[HET_SNP_IDX[bi]]) bi = k;
                const char b1 = GT21[HET_SNP_IDX[bi]][0], b2 = GT21[HET_SNP_IDX[bi]][1];
                if (b1 != ref0 && b2 != ref0) { c.alt_b0 = (uint8_t)base_kind(b1); c.alt_b1 = (uint8_t)base_kind(b2); multi = true; }
                else c.alt_b0 = (uint8_t)base_kind(b1 != ref0 ? b1 : b2);
                have = true;
                break;
            }
            case F_HOMO_INS:
                fam.alive[F_HOMO_INS][index] = false;
                consulted |= index + 1 >= LONG_INDEL;
                have = true;
                break;
            case F_ACGT_INS: {
                fam.alive[F_ACGT_INS][index] = false;
                consulted |= index / 4 + 1 >= LONG_INDEL;
                c.alt_b0 = (uint8_t)(index % 4);
                multi = "ACGT"[index % 4] != ref0;
                have = true;
                break;
            }
            case F_INSINS: {
                fam.alive[F_INSINS][index] = false;
                const int i = index / 16 + 1, j = index % 16 + 1;
                const int short_ = i <= j ? i : j, long_ = i <= j ? j : i;
                consulted = true;                                        // long_ >= 16, or the second allele's look-up (:805-823)
                const int eff = long_ < LONG_INDEL ? long_ : c.ins_avail;   // bases of the longer allele
                have = (short_ < eff ? short_ : eff) < eff;               // the shorter allele is a proper prefix
                multi = true;
                break;
            }
            case F_HOMO_DEL:
                fam.alive[F_HOMO_DEL][index] = false;
                consulted |= index + 1 >= LONG_INDEL;
                have = deletion_length(seq_len, index + 1) > 0;
                break;
            case F_ACGT_DEL: {
                fam.alive[F_ACGT_DEL][index] = false;
                const int length = index / 4 + 1;
                consulted |= length >= LONG_INDEL;
                have = deletion_length(seq_len, length) > 0;
                c.alt_b0 = (uint8_t)(index % 4);
                multi = "ACGT"[index % 4] != ref0;
                break;
            }
            case F_DELDEL: {
                fam.alive[F_DELDEL][index] = false;
                const int i = index / 15 + 1, jj = index % 15, j = (jj < i - 1 ? jj : jj + 1) + 1;   // pairs (i, j), j != i, in list order
                const int short_ = i < j ? i : j, long_ = i < j ? j : i;
                consulted |= long_ >= LONG_INDEL;
                have = deletion_length(seq_len, long_) > short_;         // the shorter deletion leaves a proper suffix
                multi = true;
                break;
            }
            default: {   // F_INSDEL
                fam.alive[F_INSDEL][index] = false;
                const int i = (index / 2) / 16 + 1, j = (index / 2) % 16 + 1;
                const int del_len = index % 2 == 0 ? j : i, ins_len = index % 2 == 0 ? i : j;
                consulted |= ins_len >= LONG_INDEL || del_len >= LONG_INDEL;
                have = deletion_length(seq_len, del_len) > 0;
                multi = true;
                break;
            }
        }
        if (have) break;
    }
    c.family = (uint8_t)family;
    c.index = (uint16_t)index;
    c.flags = (uint16_t)flags;
    auto flag = [&](int k) { return (flags >> k) & 1u; };
    // ---- the numeric half of output_with (:1002-1166) ----
    const bool is_ref = flag(F_REF);
    const bool hetero_call = flag(F_HET_SNP) || flag(F_ACGT_INS) || flag(F_INSINS) || flag(F_ACGT_DEL) || flag(F_DELDEL);
    int gt = 255;
    if (is_ref) gt = 0;
    else if (flag(F_HOMO_SNP) || flag(F_HOMO_INS) || flag(F_HOMO_DEL)) gt = 1;
    else if (hetero_call) gt = 2;
    if (multi) gt = 3;
    c.gt = (uint8_t)gt;
    c.status |= (consulted ? CLAIR_CALL_CONSULTED : 0) | (multi ? CLAIR_CALL_MULTI : 0) | (same && !is_ref ? CLAIR_CALL_SAME : 0);
    // supporting reads (:1096-1150), float32 like the NumPy scalars
    auto snp_reads = [&](int b) {   // call_var.py:1100-1107
        return ((xat(x, CENTER, b, CH_SNP) + xat(x, CENTER, b + 4, CH_SNP)) + xat(x, CENTER, b, CH_REF)) + xat(x, CENTER, b + 4, CH_REF);
    };
    float support = 0.0f;
    if (is_ref) support = xat(x, CENTER, ref_num, CH_REF) + xat(x, CENTER, ref_num + 4, CH_REF);
    else if (flag(F_HOMO_SNP) || flag(F_HET_SNP)) {
        support = support + snp_reads(c.alt_b0);
        if (c.alt_b1 != 255) support = support + snp_reads(c.alt_b1);
    } else {
        const float ins_reads = sum_rows(x, NEXT, CH_INS) - sum_rows(x, NEXT, CH_SNP);
        const float del_reads = sum_rows(x, NEXT, CH_DEL);
        if (flag(F_HOMO_INS) || flag(F_INSINS)) support = ins_reads;
        else if (flag(F_ACGT_INS)) support = multi ? ins_reads + snp_reads(c.alt_b0) : ins_reads;
        else if (flag(F_HOMO_DEL) || flag(F_DELDEL)) support = del_reads;
        else if (flag(F_ACGT_DEL)) support = multi ? del_reads + snp_reads(c.alt_b0) : del_reads;
        else if (flag(F_INSDEL)) support = (sum_rows(x, NEXT, CH_INS) + sum_rows(x, NEXT, CH_DEL)) - sum_rows(x, NEXT, CH_SNP);
    }
    c.support = support;
    // the probability the quality starts from (:568-586): gt21 class of the call (task/gt21.py:60-110) x zygosity
    if (gt != 255) {
        const bool has0 = gt == 0 || gt == 2;                       // an allele index 0 in the genotype string
        const int kref = is_ref ? ref_num : base_kind(ref0);        // kind of the REF string's first character
        int k0, k1;
        switch (family) {
            case F_REF: k0 = k1 = ref_num; break;
            case F_HOMO_SNP: case F_HET_SNP:
                if (multi) { k0 = c.alt_b0; k1 = c.alt_b1; } else { k1 = c.alt_b0; k0 = has0 ? kref : k1; }
                break;
            case F_HOMO_INS: k1 = K_INS; k0 = has0 ? kref : K_INS; break;
            case F_ACGT_INS: k1 = K_INS; k0 = multi ? (int)c.alt_b0 : (has0 ? kref : K_INS); break;
            case F_INSINS: k0 = k1 = K_INS; break;
            case F_HOMO_DEL: k1 = K_DEL; k0 = has0 ? kref : K_DEL; break;
            case F_ACGT_DEL: if (multi) { k0 = K_DEL; k1 = c.alt_b0; } else { k1 = K_DEL; k0 = has0 ? kref : K_DEL; } break;
            case F_DELDEL: k0 = k1 = K_DEL; break;
            default: k0 = K_DEL; k1 = K_INS; break;
        }
        const int gi = class_of_kinds(k0, k1);
        c.gi = (uint8_t)gi;
        if (gi != 255) c.p_call = g[gi] * z[gt == 0 ? 0 : (gt == 1 ? 1 : 2)];
    }
}

// ---- format: the record + the candidate's text -> a VCF row.  Returns 0 = no row, 1 = row appended (without '\n'), -1 = error.
int format_one(const clair_call_t &c, const char *ctg, int ctg_len, long long position, const char *seq, int seq_len, const Config &cfg,
               std::string &out) {
    if (!(c.status & CLAIR_CALL_RESOLVED)) return 0;
    auto flag = [&](int k) { return (c.flags >> k) & 1u; };
    const bool is_ref = flag(F_REF), is_multi = c.status & CLAIR_CALL_MULTI;
    if ((!cfg.show_ref && is_ref) || (c.status & CLAIR_CALL_SAME)) return 0;
    const bool hetero_call = flag(F_HET_SNP) || flag(F_ACGT_INS) || flag(F_INSINS) || flag(F_ACGT_DEL) || flag(F_DELDEL);
    if (cfg.haploid_precision && (hetero_call || flag(F_INSDEL))) return 0;
    if (cfg.haploid_sensitive && is_multi) return 0;
    if (c.gt > 3) return clair_host_fail("candidate %.*s:%lld: no genotype string (InsDel call that is not multi-allelic)", ctg_len, ctg, position), -1;
    const char *gt = GT_STR[c.gt];
    // REF / ALT (:764-947): inserted bases from the tensor's votes, deleted bases from the reference window
    const char ref0 = seq[CENTER];
    const std::string r0(1, ref0);
    auto ins = [&](int length) {
        const int n = length < LONG_INDEL ? length : c.ins_avail;
        std::string s((size_t)n, 'A');
        for (int k = 0; k < n; ++k) s[(size_t)k] = "ACGT"[(c.ins_code >> (2 * k)) & 3u];
        return s;
    };
    auto base = [](int b) { return std::string(1, "ACGT"[b & 3]); };
    std::string ref, alt;
    const int idx = c.index;
    switch (c.family) {
        case F_REF: ref = alt = std::string(1, iupac_acgt(ref0)); break;
        case F_HOMO_SNP: ref = r0; alt = base(c.alt_b0); break;
        case F_HET_SNP: ref = r0; alt = is_multi ? base(c.alt_b0) + "," + base(c.alt_b1) : base(c.alt_b0); break;
        case F_HOMO_INS: ref = r0; alt = r0 + ins(idx + 1); break;
        case F_ACGT_INS: ref = r0; alt = r0 + ins(idx / 4 + 1); if (is_multi) alt = base(c.alt_b0) + "," + alt; break;
        case F_INSINS: {
            const int i = idx / 16 + 1, j = idx % 16 + 1, short_ = i <= j ? i : j, long_ = i <= j ? j : i;
            const std::string all = ins(long_);
            ref = r0; alt = r0 + all.substr(0, (size_t)short_ < all.size() ? (size_t)short_ : all.size()) + "," + r0 + all;
            break;
        }
        case F_HOMO_DEL: ref = r0 + deletion_bases(seq, seq_len, idx + 1); alt = r0; break;
        case F_ACGT_DEL:
            ref = r0 + deletion_bases(seq, seq_len, idx / 4 + 1); alt = r0;
            if (is_multi) alt = r0 + "," + base(c.alt_b0) + ref.substr(1);
            break;
        case F_DELDEL: {
            const int i = idx / 15 + 1, jj = idx % 15, j = (jj < i - 1 ? jj : jj + 1) + 1, short_ = i < j ? i : j, long_ = i < j ? j : i;
            ref = r0 + deletion_bases(seq, seq_len, long_);
            alt = r0 + "," + r0 + ref.substr((size_t)short_ + 1);
            break;
        }
        case F_INSDEL: {
            const int i = (idx / 2) / 16 + 1, j = (idx / 2) % 16 + 1, del_len = idx % 2 == 0 ? j : i, ins_len = idx % 2 == 0 ? i : j;
            ref = r0 + deletion_bases(seq, seq_len, del_len);
            alt = r0 + "," + r0 + ins(ins_len) + ref.substr(1);
            break;
        }
        default: return clair_host_fail("candidate %.*s:%lld: call record names family %d", ctg_len, ctg, position, (int)c.family), -1;
    }
    const float depth = c.depth, support = c.support;
    double af;
    if (cfg.numpy2) { const float af32 = support / depth; af = af32 > 1.0f ? 1.0 : (double)af32; }
    else { af = (double)support / (double)depth; if (af > 1.0) af = 1.0; }
    // quality_score_from (:568-586).  The record's class is derived from allele KINDS; the strings must agree (task/gt21.py:60-110)
    const int g1 = gt[0] - '0', g2 = gt[2] - '0';
    const int gi = gt21_index_of_call(ref, alt, g1, g2);
    if (gi < 0) return clair_host_fail("candidate %.*s:%lld: call %s>%s has no gt21 class", ctg_len, ctg, position, ref.c_str(), alt.c_str()), -1;
    if (gi != c.gi) return clair_host_fail("candidate %.*s:%lld: call %s>%s is gt21 class %d by its strings, %d in its record", ctg_len, ctg, position,
                                           ref.c_str(), alt.c_str(), gi, (int)c.gi), -1;
    const float p32 = c.p_call;
    double ratio;
    if (cfg.numpy2) {
        ratio = (double)((1.0f - p32) / p32);
        if (!(ratio > 0.0))   // math.log(0) / log of a negative: the reference raises ValueError under NumPy 2
            return clair_host_fail("candidate %.*s:%lld: math domain error in the quality score (probability %g)", ctg_len, ctg, position, (double)p32), -1;
    } else {
        const double p = (double)p32;
        ratio = ((1.0 - p) + 1e-300) / (p + 1e-300);
    }
    static const double QUAL_SLOPE = -10.0 * (std::log(M_E) / std::log(10.0));
    double score = QUAL_SLOPE * std::log(ratio) + 16.0;
    if (!(score > 0.0)) score = 0.0;                                 // max(score, 0)
    const double q = std::nearbyint(score * score);                  // round(): half to even
    const long long qual = (long long)q;
    if (cfg.haploid_precision || cfg.haploid_sensitive) gt = strchr(gt, '1') ? "1" : "0";
    const char *filt = !cfg.has_qual ? "." : (qual >= cfg.qual ? "PASS" : "LowQual");
    char tail[160];
    snprintf(tail, sizeof tail, "\t%lld\t%s\t.\tGT:GQ:DP:AF\t%s:%lld:%lld:%.4f", qual, filt, gt, qual, (long long)depth, af);
    char posbuf[32];
    snprintf(posbuf, sizeof posbuf, "\t%lld\t.\t", position);
    out.append(ctg, (size_t)ctg_len);
    out.append(posbuf);
    out.append(ref);
    out.push_back('\t');
    out.append(alt);
    out.append(tail);
    return 1;
}

// the candidate's text fields out of the batch's meta buffer; returns false (message set) when they are malformed
bool candidate_text(const char *meta, const int32_t *tk, int i, const char *&ctg, int &ctg_len, long long &position, const char *&seq, int &seq_len) {
    ctg = meta + tk[0]; ctg_len = tk[1]; seq = meta + tk[4]; seq_len = tk[5];
    if (seq_len <= CENTER) { clair_host_fail("candidate %d: reference sequence has %d characters, the centre base is index 16", i, seq_len); return false; }
    char *endp = nullptr;
    const std::string ptxt(meta + tk[2], (size_t)tk[3]);
    position = strtoll(ptxt.c_str(), &endp, 10);
    if (endp == ptxt.c_str() || *endp) { clair_host_fail("candidate %d: position %s is not an integer", i, ptxt.c_str()); return false; }
    return true;
}

// candidates are independent: contiguous ranges per thread, rows concatenated in input order.  per_candidate(i, buf) appends a row
// (without '\n') and returns 1, or returns 0 (no row) / -1 (error, message set).
template <class F>
int rows_in_parallel(int n, char *out, int64_t out_cap, int64_t *out_len, int *n_rows, uint8_t *status, const clair_call_t *calls, F per_candidate) {
    const int nthreads = clair_host_threads(n);
    std::vector<std::string> parts((size_t)nthreads), errs((size_t)nthreads);
    std::vector<int> part_rows((size_t)nthreads, 0), err_at((size_t)nthreads, -1);
    auto work = [&](int t) {
        const int lo = (int)((int64_t)n * t / nthreads), hi = (int)((int64_t)n * (t + 1) / nthreads);
        std::string &buf = parts[(size_t)t];
        buf.reserve((size_t)(hi - lo) * 64);
        for (int i = lo; i < hi; ++i) {
            const size_t before = buf.size();
            bool consulted = false;
            const int rc = per_candidate(i, buf, consulted);
            if (rc == 1) { buf.push_back('\n'); ++part_rows[(size_t)t]; }
            else buf.resize(before);
            if (status && rc >= 0) status[i] = (uint8_t)((rc == 1 ? 1 : 0) | (consulted ? 2 : 0));
            if (rc < 0) { errs[(size_t)t] = clair_host_last_error(); err_at[(size_t)t] = i; return; }   // thread-local message -> caller
        }
    };
    (void)calls;
    clair_host_parallel(nthreads, work);
    for (int t = 0; t < nthreads; ++t)                       // the first failing candidate in input order
        if (err_at[(size_t)t] >= 0) return clair_host_fail("%s", errs[(size_t)t].c_str());
    size_t total = 0;
    int rows = 0;
    for (int t = 0; t < nthreads; ++t) { total += parts[(size_t)t].size(); rows += part_rows[(size_t)t]; }
    if ((int64_t)total > out_cap) return clair_host_fail("clair_host_decode_rows: output needs %lld bytes, buffer has %lld", (long long)total, (long long)out_cap);
    size_t at = 0;
    for (int t = 0; t < nthreads; ++t) { memcpy(out + at, parts[(size_t)t].data(), parts[(size_t)t].size()); at += parts[(size_t)t].size(); }
    *out_len = (int64_t)total;
    *n_rows = rows;
    return 0;
}

}  // namespace

extern "C" int clair_host_decode_rows_ex(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                                         const char *meta, const int32_t *meta_tok, int n, int show_reference, int haploid_precision,
                                         int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out, int64_t out_cap,
                                         int64_t *out_len, int *n_rows, uint8_t *status) {
    if (!x || !gt21 || !genotype || !len1 || !len2 || !meta || !meta_tok || !out || !out_len || !n_rows || n < 0)
        return clair_host_fail("clair_host_decode_rows: bad arguments");
    if (status) memset(status, 0, (size_t)n);
    const Config cfg{show_reference, haploid_precision, haploid_sensitive, qual_threshold >= 0, qual_threshold, arith_numpy2};
    return rows_in_parallel(n, out, out_cap, out_len, n_rows, status, nullptr, [&](int i, std::string &buf, bool &consulted) {
        static thread_local Families fam;
        const char *ctg, *seq;
        int ctg_len, seq_len;
        long long position;
        if (!candidate_text(meta, meta_tok + (size_t)i * 6, i, ctg, ctg_len, position, seq, seq_len)) return -1;
        clair_call_t c;
        resolve_one(x + (size_t)i * CLAIR_HOST_VALUES, gt21 + (size_t)i * 21, genotype + (size_t)i * 3, len1 + (size_t)i * 33, len2 + (size_t)i * 33,
                    seq[CENTER], seq_len, fam, c);
        consulted = c.status & CLAIR_CALL_CONSULTED;
        return format_one(c, ctg, ctg_len, position, seq, seq_len, cfg, buf);
    });
}

extern "C" int clair_host_resolve_calls(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                                        const uint8_t *centre, int n, clair_call_t *calls) {
    if (!x || !gt21 || !genotype || !len1 || !len2 || !centre || !calls || n < 0) return clair_host_fail("clair_host_resolve_calls: bad arguments");
    const int nthreads = clair_host_threads(n);
    auto work = [&](int t) {
        static thread_local Families fam;
        const int lo = (int)((int64_t)n * t / nthreads), hi = (int)((int64_t)n * (t + 1) / nthreads);
        for (int i = lo; i < hi; ++i)
            resolve_one(x + (size_t)i * CLAIR_HOST_VALUES, gt21 + (size_t)i * 21, genotype + (size_t)i * 3, len1 + (size_t)i * 33, len2 + (size_t)i * 33,
                        (char)centre[2 * (size_t)i], centre[2 * (size_t)i + 1], fam, calls[i]);
    };
    clair_host_parallel(nthreads, work);
    return 0;
}

extern "C" int clair_host_format_calls(const clair_call_t *calls, const char *meta, const int32_t *meta_tok, int n, int show_reference,
                                       int haploid_precision, int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out,
                                       int64_t out_cap, int64_t *out_len, int *n_rows, uint8_t *status) {
    if (!calls || !meta || !meta_tok || !out || !out_len || !n_rows || n < 0) return clair_host_fail("clair_host_format_calls: bad arguments");
    if (status) memset(status, 0, (size_t)n);
    const Config cfg{show_reference, haploid_precision, haploid_sensitive, qual_threshold >= 0, qual_threshold, arith_numpy2};
    return rows_in_parallel(n, out, out_cap, out_len, n_rows, status, calls, [&](int i, std::string &buf, bool &consulted) {
        const char *ctg, *seq;
        int ctg_len, seq_len;
        long long position;
        if (!candidate_text(meta, meta_tok + (size_t)i * 6, i, ctg, ctg_len, position, seq, seq_len)) return -1;
        consulted = calls[i].status & CLAIR_CALL_CONSULTED;
        return format_one(calls[i], ctg, ctg_len, position, seq, seq_len, cfg, buf);
    });
}

// The same from the COLUMNS of binary tensor records (clair_amd/tensor_binary.py: contig S37 + length, position i64, refseq S33 + length),
// n rows each: no text table has to be built for the batch first.
extern "C" int clair_host_format_calls_records(const clair_call_t *calls, const char *ctg, const uint8_t *ctg_len, const int64_t *pos, const char *seq,
                                               const uint8_t *seq_len, int n, int show_reference, int haploid_precision, int haploid_sensitive,
                                               int qual_threshold, int arith_numpy2, char *out, int64_t out_cap, int64_t *out_len, int *n_rows,
                                               uint8_t *status) {
    if (!calls || !ctg || !ctg_len || !pos || !seq || !seq_len || !out || !out_len || !n_rows || n < 0)
        return clair_host_fail("clair_host_format_calls_records: bad arguments");
    if (status) memset(status, 0, (size_t)n);
    const Config cfg{show_reference, haploid_precision, haploid_sensitive, qual_threshold >= 0, qual_threshold, arith_numpy2};
    return rows_in_parallel(n, out, out_cap, out_len, n_rows, status, calls, [&](int i, std::string &buf, bool &consulted) {
        if (seq_len[i] <= CENTER) { clair_host_fail("candidate %d: reference sequence has %d characters, the centre base is index 16", i, (int)seq_len[i]); return -1; }
        if (ctg_len[i] > 37 || seq_len[i] > 33) { clair_host_fail("candidate %d: record field longer than its column", i); return -1; }
        consulted = calls[i].status & CLAIR_CALL_CONSULTED;
        return format_one(calls[i], ctg + (size_t)i * 37, ctg_len[i], (long long)pos[i], seq + (size_t)i * 33, seq_len[i], cfg, buf);
    });
}

extern "C" int clair_host_centre_bytes(const char *meta, const int32_t *meta_tok, int n, uint8_t *centre) {
    if (!meta || !meta_tok || !centre || n < 0) return clair_host_fail("clair_host_centre_bytes: bad arguments");
    for (int i = 0; i < n; ++i) {
        const int32_t *tk = meta_tok + (size_t)i * 6;
        if (tk[5] <= CENTER) return clair_host_fail("candidate %d: reference sequence has %d characters, the centre base is index 16", i, tk[5]);
        centre[2 * (size_t)i] = (uint8_t)meta[tk[4] + CENTER];
        centre[2 * (size_t)i + 1] = (uint8_t)(tk[5] < 255 ? tk[5] : 255);
    }
    return 0;
}

extern "C" int clair_host_decode_rows(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                                      const char *meta, const int32_t *meta_tok, int n, int show_reference, int haploid_precision,
                                      int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out, int64_t out_cap,
                                      int64_t *out_len, int *n_rows) {
    return clair_host_decode_rows_ex(x, gt21, genotype, len1, len2, meta, meta_tok, n, show_reference, haploid_precision, haploid_sensitive,
                                     qual_threshold, arith_numpy2, out, out_cap, out_len, n_rows, nullptr);
}
