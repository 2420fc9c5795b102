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

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

int clair_host_fail(const char *fmt, ...);   // host_io.cpp
extern "C" int clair_host_threads(int work_items);

namespace {

constexpr int CENTER = 16, NEXT = 17, LONG_INDEL = 16;
constexpr int CH_REF = 0, CH_INS = 1, CH_DEL = 2, CH_SNP = 3;
enum { F_REF, F_HOMO_SNP, F_HET_SNP, F_HOMO_INS, F_ACGT_INS, F_INSINS, F_HOMO_DEL, F_ACGT_DEL, F_DELDEL, F_INSDEL, N_FAM };
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
inline char vote_base(const float votes[8]) {
    int best = 0;
    for (int i = 1; i < 8; ++i)
        if (votes[i] > votes[best]) best = i;
    return "ACGT"[best % 4];
}
// insertion_bases_from (no BAM): tensor-inferred bases (call_var.py:428-447, 464-477, 487-524)
std::string insertion_bases(const float *x, int length) {
    std::string out;
    float votes[8];
    if (length < LONG_INDEL) {
        for (int p = NEXT; p < NEXT + length; ++p) { insertion_votes(x, p, votes); out.push_back(vote_base(votes)); }
        return out;
    }
    for (int p = NEXT; p <= 2 * CENTER; ++p) {
        insertion_votes(x, p, votes);
        if (p < CENTER + LONG_INDEL || (double)sum8(votes) >= 0.125 * (double)sum_rows(x, p, CH_REF)) out.push_back(vote_base(votes));
        else break;
    }
    return out;
}
// deletion_bases_from (no BAM): the reference sequence after the centre (call_var.py:527-565)
inline std::string deletion_bases(const char *seq, int seq_len, int length) {
    const int a = NEXT < seq_len ? NEXT : seq_len, b = NEXT + length < seq_len ? NEXT + length : seq_len;
    return std::string(seq + a, (size_t)(b > a ? b - a : 0));
}

inline std::string homo_snp_alt(const float *g, char ref0) {   // call_var.py:60-62
    int best = 0;
    for (int k = 1; k < 4; ++k)
        if (g[HOMO_SNP_IDX[k]] > g[HOMO_SNP_IDX[best]]) best = k;
    const char *label = GT21[HOMO_SNP_IDX[best]];
    return std::string(1, label[0] != ref0 ? label[0] : label[1]);
}
inline std::string hetero_snp_alt(const float *g, char ref0) {   // call_var.py:65-67
    int best = 0;
    for (int k = 1; k < 6; ++k)
        if (g[HET_SNP_IDX[k]] > g[HET_SNP_IDX[best]]) best = k;
    const char b1 = GT21[HET_SNP_IDX[best]][0], b2 = GT21[HET_SNP_IDX[best]][1];
    if (b1 != ref0 && b2 != ref0) return std::string(1, b1) + "," + std::string(1, b2);
    return std::string(1, b1 != ref0 ? b1 : b2);
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

inline float snp_support(const float *x, char base) {   // call_var.py:1100-1107
    const int b = iupac_num(base);
    return ((xat(x, CENTER, b, CH_SNP) + xat(x, CENTER, b + 4, CH_SNP)) + xat(x, CENTER, b, CH_REF)) + xat(x, CENTER, b + 4, CH_REF);
}

struct Config { int show_ref, haploid_precision, haploid_sensitive, has_qual, qual; int numpy2; };

// One candidate -> appends a row (without '\n') to out; returns 0 = no row, 1 = row, -1 = error (message set)
int decode_one(const float *x, const float *g, const float *z, const float *l1, const float *l2, const char *ctg, int ctg_len,
               long long position, const char *seq, int seq_len, const Config &cfg, Families &fam, std::string &out, bool &consulted) {
    // consulted: the resolution passed a point where the reference asks the BAM when it has one (an indel of LONG_INDEL bases
    // or more, call_var.py:498-524, 540-565; the second allele of an Ins/Ins call, :805-823).  This decoder answers every
    // look-up with "" -- a caller that HAS a BAM re-decodes exactly those candidates on its own (clair_host_decode_rows_ex).
    consulted = false;
    const char ref0 = seq[CENTER];
    if (!(ref0 == 'A' || ref0 == 'C' || ref0 == 'G' || ref0 == 'T' || ref0 == 'U')) return 0;   // call_var.py:1018
    float dsum[8];
    for (int r = 0; r < 8; ++r) dsum[r] = xat(x, CENTER, r, CH_DEL) + xat(x, CENTER, r, CH_REF);
    const float depth = sum8(dsum);                                                              // :1022-1024
    if (depth == 0.0f) return 0;
    // ---- the ten outcome families, products left to right as written in possible_outcome_probabilites_from (:589-690) ----
    const float p_ref = z[0], p_hom = z[1], p_het = z[2];
    const float z1 = l1[16], z2 = l2[16], zero = z1 * z2;
    const int ref_class = HOMO_SNP_IDX[iupac_num(ref0)];
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
        const float c = z1 * del2[i], d = del1[i] * z2, one_del = c > d ? c : d;
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
    // np.maximum propagates NaN where a > b does not; probabilities are finite, but stay exact for the NaN case too
    // (not needed: softmax outputs are finite by construction)
    for (int k = 0; k < N_FAM; ++k)
        for (int i = 0; i < FAM_SIZE[k]; ++i) fam.alive[k][i] = true;

    // ---- iterative arg-max with exact-equality membership (output_from, :693-947) ----
    bool flags[N_FAM];
    std::string ref, alt;
    for (;;) {
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
        if (best == tops[F_REF]) {
            for (int k = 0; k < N_FAM; ++k) flags[k] = k == F_REF;
            ref = alt = std::string(1, iupac_acgt(ref0));
            break;
        }
        flags[F_REF] = false;
        int first[N_FAM];
        for (int k = 1; k < N_FAM; ++k) {
            first[k] = -1;
            for (int i = 0; i < FAM_SIZE[k]; ++i)
                if (fam.alive[k][i] && fam.v[k][i] == best) { first[k] = i; break; }
            flags[k] = first[k] >= 0;
        }
        bool have = false;
        const std::string r0(1, ref0);
        if (flags[F_HOMO_SNP]) { ref = r0; alt = homo_snp_alt(g, ref0); have = true; }
        else if (flags[F_HET_SNP]) { ref = r0; alt = hetero_snp_alt(g, ref0); have = true; }
        else if (flags[F_HOMO_INS]) {
            const int idx = first[F_HOMO_INS];
            fam.alive[F_HOMO_INS][idx] = false;
            consulted |= idx + 1 >= LONG_INDEL;
            const std::string ins = insertion_bases(x, idx + 1);
            if (!ins.empty()) { ref = r0; alt = r0 + ins; have = true; }
        } else if (flags[F_ACGT_INS]) {
            const int idx = first[F_ACGT_INS];
            fam.alive[F_ACGT_INS][idx] = false;
            const int length = idx / 4 + 1;
            const char base = "ACGT"[idx % 4];
            consulted |= length >= LONG_INDEL;
            const std::string ins = insertion_bases(x, length);
            if (!ins.empty()) {
                ref = r0; alt = r0 + ins;
                if (std::string(1, base) != ref) alt = std::string(1, base) + "," + alt;
                have = true;
            }
        } else if (flags[F_INSINS]) {
            const int idx = first[F_INSINS];
            fam.alive[F_INSINS][idx] = false;
            const int i = idx / 16 + 1, j = idx % 16 + 1;
            const int short_ = i <= j ? i : j, long_ = i <= j ? j : i;
            consulted |= long_ >= LONG_INDEL;
            const std::string ins = insertion_bases(x, long_);
            if (!ins.empty()) {
                consulted = true;
                const std::string other = ins.substr(0, (size_t)short_ < ins.size() ? (size_t)short_ : ins.size());   // look-up "" -> ins[0:short]
                const std::string firsts = r0 + other, second = r0 + ins;
                if (firsts != second) { ref = r0; alt = firsts + "," + second; have = true; }
            }
        } else if (flags[F_HOMO_DEL]) {
            const int idx = first[F_HOMO_DEL];
            fam.alive[F_HOMO_DEL][idx] = false;
            consulted |= idx + 1 >= LONG_INDEL;
            const std::string dele = deletion_bases(seq, seq_len, idx + 1);
            if (!dele.empty()) { ref = r0 + dele; alt = r0; have = true; }
        } else if (flags[F_ACGT_DEL]) {
            const int idx = first[F_ACGT_DEL];
            fam.alive[F_ACGT_DEL][idx] = false;
            const int length = idx / 4 + 1;
            const char base = "ACGT"[idx % 4];
            consulted |= length >= LONG_INDEL;
            const std::string dele = deletion_bases(seq, seq_len, length);
            if (!dele.empty()) {
                ref = r0 + dele; alt = r0;
                if (base != ref0) alt = r0 + "," + std::string(1, base) + ref.substr(1);
                have = true;
            }
        } else if (flags[F_DELDEL]) {
            const int idx = first[F_DELDEL];
            fam.alive[F_DELDEL][idx] = false;
            const int i = idx / 15 + 1, jj = idx % 15, j = (jj < i - 1 ? jj : jj + 1) + 1;   // pairs (i, j), j != i, in list order
            const int short_ = i < j ? i : j, long_ = i < j ? j : i;
            consulted |= long_ >= LONG_INDEL;
            const std::string dele = deletion_bases(seq, seq_len, long_);
            if (!dele.empty()) {
                const std::string full = r0 + dele;
                const std::string firsts = r0, second = r0 + (full.size() > (size_t)short_ + 1 ? full.substr((size_t)short_ + 1) : std::string());
                if (firsts != second && full != firsts && full != second) { ref = full; alt = firsts + "," + second; have = true; }
            }
        } else if (flags[F_INSDEL]) {
            const int idx = first[F_INSDEL];
            fam.alive[F_INSDEL][idx] = false;
            const int i = (idx / 2) / 16 + 1, j = (idx / 2) % 16 + 1;
            const int del_len = idx % 2 == 0 ? j : i, ins_len = idx % 2 == 0 ? i : j;
            consulted |= ins_len >= LONG_INDEL || del_len >= LONG_INDEL;
            const std::string ins = insertion_bases(x, ins_len), dele = deletion_bases(seq, seq_len, del_len);
            if (!ins.empty() && !dele.empty()) { ref = r0 + dele; alt = r0 + "," + r0 + ins + ref.substr(1); have = true; }
        }
        if (have) break;
    }
    // ---- output_with (:1002-1196) ----
    const bool is_ref = flags[F_REF];
    if ((!cfg.show_ref && is_ref) || (!is_ref && ref == alt)) return 0;
    const bool is_multi = alt.find(',') != std::string::npos;
    const bool hetero_call = flags[F_HET_SNP] || flags[F_ACGT_INS] || flags[F_INSINS] || flags[F_ACGT_DEL] || flags[F_DELDEL];
    if (cfg.haploid_precision && (hetero_call || flags[F_INSDEL])) return 0;
    if (cfg.haploid_sensitive && is_multi) return 0;
    const char *gt = nullptr;
    if (is_ref) gt = GT_STR[0];
    else if (flags[F_HOMO_SNP] || flags[F_HOMO_INS] || flags[F_HOMO_DEL]) gt = GT_STR[1];
    else if (hetero_call) gt = GT_STR[2];
    if (is_multi) gt = GT_STR[3];
    if (!gt) return clair_host_fail("candidate %.*s:%lld: no genotype string (InsDel call that is not multi-allelic)", ctg_len, ctg, position), -1;
    // supporting reads (:1096-1150), float32 like the NumPy scalars
    float support = 0.0f;
    if (flags[F_REF]) {
        const int b = iupac_num(ref[0]);
        support = xat(x, CENTER, b, CH_REF) + xat(x, CENTER, b + 4, CH_REF);
    } else if (flags[F_HOMO_SNP] || flags[F_HET_SNP]) {
        for (char c : alt)
            if (c != ',') support = support + snp_support(x, c);
    } else {
        const float ins_reads = sum_rows(x, NEXT, CH_INS) - sum_rows(x, NEXT, CH_SNP);
        const float del_reads = sum_rows(x, NEXT, CH_DEL);
        if (flags[F_HOMO_INS] || flags[F_INSINS]) support = ins_reads;
        else if (flags[F_ACGT_INS]) support = is_multi ? ins_reads + snp_support(x, alt[0]) : ins_reads;
        else if (flags[F_HOMO_DEL] || flags[F_DELDEL]) support = del_reads;
        else if (flags[F_ACGT_DEL]) support = is_multi ? del_reads + snp_support(x, alt[alt.find(',') + 1]) : del_reads;
        else if (flags[F_INSDEL]) support = (sum_rows(x, NEXT, CH_INS) + sum_rows(x, NEXT, CH_DEL)) - sum_rows(x, NEXT, CH_SNP);
    }
    double af;
    if (cfg.numpy2) { const float af32 = support / depth; af = af32 > 1.0f ? 1.0 : (double)af32; }
    else { af = (double)support / (double)depth; if (af > 1.0) af = 1.0; }
    // quality_score_from (:568-586)
    const int g1 = gt[0] - '0', g2 = gt[2] - '0';
    const int gi = gt21_index_of_call(ref, alt, g1, g2);
    if (gi < 0) return clair_host_fail("candidate %.*s:%lld: call %s>%s has no gt21 class", ctg_len, ctg, position, ref.c_str(), alt.c_str()), -1;
    const int zi = (g1 == 0 && g2 == 0) ? 0 : (g1 == g2 ? 1 : 2);
    const float p32 = g[gi] * z[zi];
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

}  // namespace

extern "C" int clair_host_decode_rows_ex(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                                         const char *meta, const int32_t *meta_tok, int n, int show_reference, int haploid_precision,
                                         int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out, int64_t out_cap,
                                         int64_t *out_len, int *n_rows, uint8_t *status) {
    if (!x || !gt21 || !genotype || !len1 || !len2 || !meta || !meta_tok || !out || !out_len || !n_rows || n < 0)
        return clair_host_fail("clair_host_decode_rows: bad arguments");
    if (status) memset(status, 0, (size_t)n);
    Config cfg{show_reference, haploid_precision, haploid_sensitive, qual_threshold >= 0, qual_threshold, arith_numpy2};
    // candidates are independent: contiguous ranges per thread, rows concatenated in input order
    const int nthreads = clair_host_threads(n);
    std::vector<std::string> parts((size_t)nthreads), errs((size_t)nthreads);
    std::vector<int> part_rows((size_t)nthreads, 0), err_at((size_t)nthreads, -1);
    auto work = [&](int t) {
        static thread_local Families fam;
        const int lo = (int)((int64_t)n * t / nthreads), hi = (int)((int64_t)n * (t + 1) / nthreads);
        std::string &buf = parts[(size_t)t];
        buf.reserve((size_t)(hi - lo) * 64);
        for (int i = lo; i < hi; ++i) {
            const int32_t *tk = meta_tok + (size_t)i * 6;
            const char *ctg = meta + tk[0], *pos = meta + tk[2], *seq = meta + tk[4];
            int rc;
            if (tk[5] <= CENTER) { clair_host_fail("candidate %d: reference sequence has %d characters, the centre base is index 16", i, tk[5]); rc = -1; }
            else {
                char *endp = nullptr;
                const std::string ptxt(pos, (size_t)tk[3]);
                const long long position = strtoll(ptxt.c_str(), &endp, 10);
                if (endp == ptxt.c_str() || *endp) { clair_host_fail("candidate %d: position %s is not an integer", i, ptxt.c_str()); rc = -1; }
                else {
                    const size_t before = buf.size();
                    bool consulted = false;
                    rc = decode_one(x + (size_t)i * CLAIR_HOST_VALUES, gt21 + (size_t)i * 21, genotype + (size_t)i * 3, len1 + (size_t)i * 33,
                                    len2 + (size_t)i * 33, ctg, tk[1], position, seq, tk[5], cfg, fam, buf, consulted);
                    if (rc == 1) { buf.push_back('\n'); ++part_rows[(size_t)t]; }
                    else buf.resize(before);
                    if (status && rc >= 0) status[i] = (uint8_t)((rc == 1 ? 1 : 0) | (consulted ? 2 : 0));
                }
            }
            if (rc < 0) { errs[(size_t)t] = clair_host_last_error(); err_at[(size_t)t] = i; return; }   // thread-local message -> caller
        }
    };
    if (nthreads <= 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    for (int t = 0; t < nthreads; ++t)                       // the first failing candidate in input order
        if (err_at[(size_t)t] >= 0) return clair_host_fail("%s", errs[(size_t)t].c_str());
    std::string buf;
    int rows = 0;
    for (int t = 0; t < nthreads; ++t) { buf += parts[(size_t)t]; rows += part_rows[(size_t)t]; }
    if ((int64_t)buf.size() > out_cap) return clair_host_fail("clair_host_decode_rows: output needs %lld bytes, buffer has %lld", (long long)buf.size(), (long long)out_cap);
    memcpy(out, buf.data(), buf.size());
    *out_len = (int64_t)buf.size();
    *n_rows = rows;
    return 0;
}

extern "C" int clair_host_decode_rows(const float *x, const float *gt21, const float *genotype, const float *len1, const float *len2,
                                      const char *meta, const int32_t *meta_tok, int n, int show_reference, int haploid_precision,
                                      int haploid_sensitive, int qual_threshold, int arith_numpy2, char *out, int64_t out_cap,
                                      int64_t *out_len, int *n_rows) {
    return clair_host_decode_rows_ex(x, gt21, genotype, len1, len2, meta, meta_tok, n, show_reference, haploid_precision, haploid_sensitive,
                                     qual_threshold, arith_numpy2, out, out_cap, out_len, n_rows, nullptr);
}
