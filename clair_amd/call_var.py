"""call_var: tensors -> MI355X forward pass -> VCF.  Same CLI flags, same record formats.

Counterpart of /root/reference/clair/call_var.py for the inference path:

  C2  driver (load | predict | output pipeline, rows in input order)     call_var.py:1312-1367
  C3  per-candidate decode: 10 outcome families (1 179 float32 products), iterative
      arg-max with exact-equality membership and fall-through when indel bases cannot
      be recovered; REF/ALT, GT, AF, QUAL, filters, row format              :344-425, 568-947, 1002-1196
  C4  VCF header, optional BAM look-ups                                    :102-170, 223-341
  C5  --output_for_ensemble writer / --input_probabilities reader          :950-1000, 1276-1309
  C6  argparse surface                                                     :1370-1435

Design differences (results identical, pinned by tests/golden/* minted from the reference):
  * the outcome families are built for a whole batch at once with float32 array products in
    the reference's operand order; reference / SNP calls -- the bulk of real candidates --
    are resolved from the family maxima alone, only indel calls walk the per-candidate
    removal loop (`_IndelResolver`);
  * predict is asynchronous on the GPU (clair_submit / clair_wait), so the driver overlaps
    parse, forward pass and decode without a Python thread blocked inside predict;
  * BAM / FASTA are optional: without pysam (or without --bam_fn on disk) every BAM look-up
    behaves as "no reads found", which is the reference's own fall-back to tensor-inferred
    bases (:520-524, :562-564).

Arithmetic mode.  The reference's QUAL (:568-586) and AF (:1151) formulas promote float32
scalars to float64 under the NumPy 1.18 it pins (README.md:127) and stay float32 under NumPy 2.
``arith="legacy"`` (default) computes them in float64 as the pinned reference does;
``arith="numpy2"`` reproduces the reference as it runs in the build container (tests/golden/decode_rows.json.gz).
The default is pinned by tests/golden/decode_rows_legacy.json.gz: the reference's own writer with float32 scalars
promoted as NumPy 1.x did (tools/make_ref_goldens.py: L32 / LArr).
"""
import logging
import math
import os
import sys
from argparse import ArgumentParser
from collections import namedtuple
from threading import Event, Thread
from time import time

import numpy as np

from clair_amd import param, task
from clair_amd import utils as ingest

logging.basicConfig(format='%(message)s', level=logging.INFO)

CENTER = param.flankingBaseNum            # 16
NEXT = CENTER + 1
CH_REF, CH_INS, CH_DEL, CH_SNP = 0, 1, 2, 3   # call_var.py:53-57
LONG_INDEL = task.LENGTH_MAX              # lengths >= 16 need a BAM look-up / inference  (:29)
LONG_INDEL_CAP = 50                       # :30
INFER_MIN_AF = 0.125                      # :31
QUAL_SLOPE = -10 * math.log(math.e, 10)

OutputConfig = namedtuple('OutputConfig', [
    'is_show_reference', 'is_debug', 'is_haploid_precision_mode_enabled',
    'is_haploid_sensitive_mode_enabled', 'is_output_for_ensemble', 'quality_score_for_pass'])

# outcome families in the order of the reference's if/elif chain (:733-762)
(F_REF, F_HOMO_SNP, F_HET_SNP, F_HOMO_INS, F_ACGT_INS, F_INSINS,
 F_HOMO_DEL, F_ACGT_DEL, F_DELDEL, F_INSDEL) = range(10)
N_FAMILIES = 10
_REF_CLASS = {"A": task.GT21_INDEX["AA"], "C": task.GT21_INDEX["CC"],
              "G": task.GT21_INDEX["GG"], "T": task.GT21_INDEX["TT"]}
_LEN = np.arange(1, LONG_INDEL + 1)
_INS_COLS = task.LENGTH_OFFSET + _LEN     # 17..32
_DEL_COLS = task.LENGTH_OFFSET - _LEN     # 15..0
_OFF_DIAG = ~np.eye(LONG_INDEL, dtype=bool)


# =============================================================================================
# BAM / FASTA look-ups (optional)
# =============================================================================================
class AlignmentLookup(object):
    """Most frequent inserted / deleted sequence right after a position, from the BAM.

    Counterpart of insertion_bases_using_pysam_from / deletion_bases_using_pysam_from
    (call_var.py:102-170).  When pysam or the files are unavailable every query returns "".
    """

    FILTER_FLAG = 2316   # shared/param.py:6

    def __init__(self, bam_path=None, fasta_path=None):
        self.sam = self.fasta = None
        try:
            import pysam
        except ImportError:
            pysam = None
        if pysam is not None and bam_path and os.path.isfile(bam_path):
            self.sam = pysam.AlignmentFile(bam_path, mode="rb")
        if pysam is not None and fasta_path and os.path.isfile(fasta_path):
            self.fasta = pysam.FastaFile(filename=fasta_path)

    def close(self):
        for handle in (self.sam, self.fasta):
            if handle is not None:
                handle.close()

    def _indel_tokens(self, contig, position, sign):
        """Yield (length, text-after-digits) of every '+'/'-' pileup token at position-1."""
        if self.sam is None:
            return
        try:
            for column in self.sam.pileup(contig, start=position, stop=position + 1,
                                          flag_filter=self.FILTER_FLAG, min_base_quality=0, max_depth=250):
                if column.reference_pos != position - 1:
                    continue
                for token in column.get_query_sequences(mark_matches=False, mark_ends=False, add_indels=True):
                    if len(token) < 4 or token[1] != sign:
                        continue
                    digits = 0
                    while 2 + digits < len(token) and token[2 + digits].isdigit():
                        digits += 1
                    yield int(token[2:2 + digits]), token[2 + digits:]
        except AssertionError:
            return

    @staticmethod
    def _most_frequent(counts):
        best, best_n = "", 0
        for key, n in counts.items():      # first inserted wins ties, like max(dict, key=dict.get)
            if n > best_n:
                best, best_n = key, n
        return best

    def insertion(self, contig, position, min_len=1, max_len=LONG_INDEL_CAP, ignore=""):
        counts = {}
        for length, tail in self._indel_tokens(contig, position, "+"):
            bases = tail.upper()
            if min_len <= length <= max_len and bases != ignore:
                counts[bases] = counts.get(bases, 0) + 1
        return self._most_frequent(counts)

    def deletion(self, contig, position, min_len=1, max_len=LONG_INDEL_CAP):
        counts = {}
        for length, _ in self._indel_tokens(contig, position, "-"):
            if self.fasta is None:
                continue
            bases = self.fasta.fetch(reference=contig, start=position, end=position + length)
            if min_len <= length <= max_len:
                counts[bases] = counts.get(bases, 0) + 1
        return self._most_frequent(counts)


# =============================================================================================
# indel bases from the pileup tensor
# =============================================================================================
def _insertion_votes(x, pos):
    """Per-base insertion evidence at one position (both strands folded, SNP counts removed);
    entries 4..7 are zero exactly as the reference leaves them (call_var.py:430-437, 466-472)."""
    votes = np.zeros(8, dtype=x.dtype)
    votes[:4] = (x[pos, :4, CH_INS] + x[pos, 4:, CH_INS]) - (x[pos, :4, CH_SNP] + x[pos, 4:, CH_SNP])
    return votes


def insertion_bases_from_tensor(x, length):
    """call_var.py:464-477."""
    return "".join("ACGT"[int(np.argmax(_insertion_votes(x, p))) % 4] for p in range(NEXT, NEXT + length))


def inferred_insertion_bases(x):
    """call_var.py:428-447: extend while insert evidence >= 12.5 % of the reference counts
    (the first 15 positions are always taken)."""
    out = []
    for p in range(NEXT, 2 * CENTER + 1):
        votes = _insertion_votes(x, p)
        if p < CENTER + LONG_INDEL or float(votes.sum()) >= INFER_MIN_AF * float(x[p, :, CH_REF].sum()):
            out.append("ACGT"[int(np.argmax(votes)) % 4])
        else:
            break
    return "".join(out)


def _cap(length):
    return LONG_INDEL_CAP if length >= LONG_INDEL else length   # call_var.py:480-484


class IndelBases(object):
    """insertion_bases_from / deletion_bases_from (call_var.py:487-565) over an AlignmentLookup."""

    def __init__(self, lookup, always_use_bam=False):
        self.lookup = lookup
        self.always_use_bam = always_use_bam

    def insertion(self, x, length, contig, position):
        if self.always_use_bam:
            return self.lookup.insertion(contig, position, length, _cap(length))
        if length < LONG_INDEL:
            return insertion_bases_from_tensor(x, length)
        return self.lookup.insertion(contig, position, LONG_INDEL) or inferred_insertion_bases(x)

    def deletion(self, x, length, contig, position, seq):
        if self.always_use_bam:
            return self.lookup.deletion(contig, position, length, _cap(length))
        if length >= LONG_INDEL:
            found = self.lookup.deletion(contig, position, LONG_INDEL)
            if len(found) >= CENTER:
                return found
        return seq[NEXT:NEXT + length]


# =============================================================================================
# outcome families
# =============================================================================================
class OutcomeFamilies(object):
    """The ten outcome families of possible_outcome_probabilites_from (call_var.py:589-690) for a
    whole batch, as float32 arrays whose element order equals the reference's list order and
    whose products are formed left-to-right exactly as written there."""

    def __init__(self, gt21, genotype, len1, len2, ref_class):
        n = gt21.shape[0]
        rows = np.arange(n)
        p_ref, p_hom, p_het = genotype[:, 0], genotype[:, 1], genotype[:, 2]
        zero = len1[:, 16] * len2[:, 16]
        ins1, ins2 = len1[:, _INS_COLS], len2[:, _INS_COLS]
        del1, del2 = len1[:, _DEL_COLS], len2[:, _DEL_COLS]
        z1, z2 = len1[:, 16:17], len2[:, 16:17]
        g = gt21
        fam = [None] * N_FAMILIES
        fam[F_REF] = ((zero * p_ref) * g[rows, ref_class])[:, None]
        fam[F_HOMO_SNP] = (zero * p_hom)[:, None] * g[:, task.HOMO_SNP_IDX]
        fam[F_HET_SNP] = (zero * p_het)[:, None] * g[:, task.HETERO_SNP_IDX]
        fam[F_HOMO_INS] = (ins1 * ins2) * (p_hom * g[:, task.IDX_INSINS])[:, None]
        fam[F_INSINS] = ((ins1[:, :, None] * ins2[:, None, :])
                         * (p_het * g[:, task.IDX_INSINS])[:, None, None]).reshape(n, -1)
        one_ins = np.maximum(z1 * ins2, ins1 * z2)
        fam[F_ACGT_INS] = ((one_ins[:, :, None] * g[:, None, task.INS_BASE_IDX])
                           * p_het[:, None, None]).reshape(n, -1)
        fam[F_HOMO_DEL] = (del1 * del2) * (p_hom * g[:, task.IDX_DELDEL])[:, None]
        deldel = (del1[:, :, None] * del2[:, None, :]) * (p_het * g[:, task.IDX_DELDEL])[:, None, None]
        fam[F_DELDEL] = deldel[:, _OFF_DIAG]                      # i == j is not an outcome (:403-404)
        one_del = np.maximum(z1 * del2, del1 * z2)
        fam[F_ACGT_DEL] = ((one_del[:, :, None] * g[:, None, task.DEL_BASE_IDX])
                           * p_het[:, None, None]).reshape(n, -1)
        e3 = (p_het * g[:, task.IDX_INSDEL])[:, None, None]
        insdel = np.empty((n, LONG_INDEL, LONG_INDEL, 2), dtype=np.float32)
        insdel[..., 0] = (ins1[:, :, None] * del2[:, None, :]) * e3   # len1 = +i, len2 = -j  -> key (j, i)
        insdel[..., 1] = (del1[:, :, None] * ins2[:, None, :]) * e3   # len1 = -i, len2 = +j  -> key (i, j)
        fam[F_INSDEL] = insdel.reshape(n, -1)
        self.fam = fam
        self.top = np.stack([f.max(axis=1) for f in fam], axis=1)     # [n,10]
        self.best = self.top.max(axis=1)
        self.flags = self.top == self.best[:, None]


# keys of the pair families, in list order
_PAIRS_ALL = [(i, j) for i in range(1, 17) for j in range(1, 17)]
_PAIRS_DELDEL = [(i, j) for (i, j) in _PAIRS_ALL if i != j]


class _IndelResolver(object):
    """The `while reference_base is None or alternate_base is None` loop of output_from
    (call_var.py:730-935) for one candidate whose current best outcome is an indel family."""

    INDEL_FAMILIES = (F_HOMO_INS, F_ACGT_INS, F_INSINS, F_HOMO_DEL, F_ACGT_DEL, F_DELDEL, F_INSDEL)

    def __init__(self, families, row, gt21_row, x, seq, contig, position, bases, lookup):
        self.vals = [np.array(f[row], dtype=np.float32, copy=True) for f in families.fam]
        self.alive = [np.ones(v.shape[0], dtype=bool) for v in self.vals]
        self.gt21, self.x, self.seq = gt21_row, x, seq
        self.contig, self.position = contig, position
        self.bases, self.lookup = bases, lookup

    def _family_top(self, k):
        live = self.vals[k][self.alive[k]]
        return live.max() if live.size else 0     # `max(...) if len(...) else 0` (:733-743)

    def _pop_first(self, k, value):
        idx = int(np.flatnonzero(self.alive[k] & (self.vals[k] == value))[0])
        self.alive[k][idx] = False
        return idx

    def run(self):
        seq, x, ref0 = self.seq, self.x, self.seq[CENTER]
        while True:
            tops = [self._family_top(k) for k in range(N_FAMILIES)]
            best = max(tops)
            if best == tops[F_REF]:
                ref_acgt = task.IUPAC_TO_ACGT[ref0]
                return _only(F_REF), ref_acgt, ref_acgt
            flags = tuple(k != F_REF and bool(np.any(self.alive[k] & (self.vals[k] == best)))
                          for k in range(N_FAMILIES))
            ref = alt = None
            if flags[F_HOMO_SNP]:
                ref, alt = ref0, _homo_snp_alt(self.gt21, ref0)
            elif flags[F_HET_SNP]:
                ref, alt = ref0, _hetero_snp_alt(self.gt21, ref0)
            elif flags[F_HOMO_INS]:
                length = int(_LEN[self._pop_first(F_HOMO_INS, best)])
                ins = self.bases.insertion(x, length, self.contig, self.position)
                if ins:
                    ref, alt = ref0, ref0 + ins
            elif flags[F_ACGT_INS]:
                idx = self._pop_first(F_ACGT_INS, best)
                length, base = idx // 4 + 1, "ACGT"[idx % 4]
                ins = self.bases.insertion(x, length, self.contig, self.position)
                if ins:
                    ref, alt = ref0, ref0 + ins
                    if base != ref:
                        alt = "%s,%s" % (base, alt)
            elif flags[F_INSINS]:
                i, j = _PAIRS_ALL[self._pop_first(F_INSINS, best)]
                short, long_ = (i, j) if i <= j else (j, i)
                ins = self.bases.insertion(x, long_, self.contig, self.position)
                if ins:
                    other = (self.lookup.insertion(self.contig, self.position, short, _cap(short), ins)
                             or ins[0:short])
                    first, second = ref0 + other, ref0 + ins
                    if first != second:
                        ref, alt = ref0, "%s,%s" % (first, second)
            elif flags[F_HOMO_DEL]:
                length = int(_LEN[self._pop_first(F_HOMO_DEL, best)])
                dele = self.bases.deletion(x, length, self.contig, self.position, seq)
                if dele:
                    ref, alt = ref0 + dele, ref0
            elif flags[F_ACGT_DEL]:
                idx = self._pop_first(F_ACGT_DEL, best)
                length, base = idx // 4 + 1, "ACGT"[idx % 4]
                dele = self.bases.deletion(x, length, self.contig, self.position, seq)
                if dele:
                    ref, alt = ref0 + dele, ref0
                    if base != ref0:
                        alt = "%s,%s" % (ref0, base + ref[1:])
            elif flags[F_DELDEL]:
                i, j = _PAIRS_DELDEL[self._pop_first(F_DELDEL, best)]
                short, long_ = (i, j) if i < j else (j, i)
                dele = self.bases.deletion(x, long_, self.contig, self.position, seq)
                if dele:
                    full = ref0 + dele
                    first, second = ref0, ref0 + full[short + 1:]
                    if first != second and full != first and full != second:
                        ref, alt = full, "%s,%s" % (first, second)
            elif flags[F_INSDEL]:
                idx = self._pop_first(F_INSDEL, best)
                i, j = _PAIRS_ALL[idx // 2]
                del_len, ins_len = (j, i) if idx % 2 == 0 else (i, j)
                ins = self.bases.insertion(x, ins_len, self.contig, self.position)
                dele = self.bases.deletion(x, del_len, self.contig, self.position, seq)
                if ins and dele:
                    ref = ref0 + dele
                    alt = "%s,%s" % (ref0, ref0 + ins + ref[1:])
            if ref is not None and alt is not None:
                return flags, ref, alt


def _only(k):
    return tuple(i == k for i in range(N_FAMILIES))


def _homo_snp_alt(gt21_row, ref0):
    label = task.HOMO_SNP[int(np.argmax(gt21_row[list(task.HOMO_SNP_IDX)]))]       # call_var.py:60-62
    return label[0] if label[0] != ref0 else label[1]


def _hetero_snp_alt(gt21_row, ref0):
    label = task.HETERO_SNP[int(np.argmax(gt21_row[list(task.HETERO_SNP_IDX)]))]   # call_var.py:65-67
    b1, b2 = label[0], label[1]
    if b1 != ref0 and b2 != ref0:
        return "%s,%s" % (b1, b2)
    return b1 if b1 != ref0 else b2


# =============================================================================================
# per-candidate record
# =============================================================================================
def quality_score(ref, alt, genotype_string, gt21_row, genotype_row, arith):
    """call_var.py:568-586."""
    g1, g2 = int(genotype_string[0]), int(genotype_string[2])
    p32 = gt21_row[task.gt21_index_of_call(ref, alt, g1, g2)] * genotype_row[task.genotype_class_of(g1, g2)]
    if arith == "numpy2":
        ratio = float((np.float32(1.0) - p32) / p32)
    else:
        p = float(p32)
        ratio = ((1.0 - p) + 1e-300) / (p + 1e-300)
    score = max(QUAL_SLOPE * math.log(ratio) + 16, 0)
    return int(round(score * score))


def _snp_support(x, base):
    b = task.IUPAC_TO_NUM[base]
    return (x[CENTER, b, CH_SNP] + x[CENTER, b + 4, CH_SNP] + x[CENTER, b, CH_REF] + x[CENTER, b + 4, CH_REF])


def supporting_reads(x, flags, ref, alt, is_multi):
    """call_var.py:1096-1150 -- same if/elif order over the family flags."""
    if flags[F_REF]:
        b = task.IUPAC_TO_NUM[ref]
        return x[CENTER, b, CH_REF] + x[CENTER, b + 4, CH_REF]
    if flags[F_HOMO_SNP] or flags[F_HET_SNP]:
        total = 0
        for base in alt:
            if base != ",":
                total = total + _snp_support(x, base)
        return total
    ins_reads = x[NEXT, :, CH_INS].sum() - x[NEXT, :, CH_SNP].sum()
    del_reads = x[NEXT, :, CH_DEL].sum()
    if flags[F_HOMO_INS] or flags[F_INSINS]:
        return ins_reads
    if flags[F_ACGT_INS]:
        return ins_reads + (_snp_support(x, alt.split(",")[0][0]) if is_multi else 0)
    if flags[F_HOMO_DEL] or flags[F_DELDEL]:
        return del_reads
    if flags[F_ACGT_DEL]:
        return del_reads + (_snp_support(x, alt.split(",")[1][0]) if is_multi else 0)
    if flags[F_INSDEL]:
        return (x[NEXT, :, CH_INS].sum() + x[NEXT, :, CH_DEL].sum()) - x[NEXT, :, CH_SNP].sum()
    return 0


def _debug_line(contig, position, gt21_row, genotype_row, l1_row, l2_row, note):
    fmt = lambda row: ["{:0.8f}".format(v) for v in row]   # noqa: E731  (call_var.py:239-259)
    return "{}\t{}\t{}\t{}\t{}\t{}\t{}".format(contig, position, fmt(gt21_row), fmt(genotype_row),
                                             fmt(l1_row), fmt(l2_row), note)


class VariantDecoder(object):
    """batch_output / output_with (call_var.py:1002-1236): (X, infos, 4 prob arrays) -> text rows."""

    def __init__(self, config, lookup=None, always_use_bam=False, arith="legacy", native=True):
        if arith not in ("legacy", "numpy2"):
            raise ValueError("arith must be 'legacy' or 'numpy2'")
        self.cfg = config
        self.lookup = lookup if lookup is not None else AlignmentLookup()
        self.bases = IndelBases(self.lookup, always_use_bam)
        self.arith = arith
        self.native = native

    def decode_batch(self, X, infos, Y):
        """Rows of one batch.  The common configuration (no --debug, no --output_for_ensemble, no --pysam_for_all_indel_bases)
        runs in the native decoder (include/clair_host.h: clair_host_decode_rows_ex, byte-identical, ~100x).  With a BAM open the
        reference consults it only for indels of 16 bases or more and for the second allele of Ins/Ins calls
        (call_var.py:498-524, 540-565, 805-823): the native decoder flags exactly those candidates and only they are decoded
        again on the Python look-up path; everything else runs on the Python restatement below."""
        cfg = self.cfg
        if self.native_applies():
            if len(Y[0]) != len(infos):
                sys.exit("Inconsistent shape between input tensor and output predictions %d/%d" % (len(infos), len(Y[0])))
            from clair_amd import _hostapi
            rows, status = _hostapi.decode_rows(X, infos, Y, cfg.is_show_reference, cfg.is_haploid_precision_mode_enabled,
                                                cfg.is_haploid_sensitive_mode_enabled, cfg.quality_score_for_pass,
                                                self.arith == "numpy2", with_status=True)
            if self.lookup.sam is None or not (status & 2).any():
                return rows
            out, at = [], 0
            Y = [np.asarray(a, dtype=np.float32) for a in Y]
            for i, st in enumerate(status):
                if st & 2:
                    out.extend(self.decode_batch_py(X[i:i + 1], infos[i:i + 1], [a[i:i + 1] for a in Y]))
                elif st & 1:
                    out.append(rows[at])
                at += int(st & 1)
            return out
        return self.decode_batch_py(X, infos, Y)

    def native_applies(self):
        """The configuration the native decoders cover (host: clair_host_decode_rows_ex; device: clair_submit_ex + clair_host_format_calls):
        no --debug, no --output_for_ensemble, no --pysam_for_all_indel_bases, an integer --qual."""
        cfg = self.cfg
        return (self.native and not cfg.is_debug and not cfg.is_output_for_ensemble and not self.bases.always_use_bam
                and isinstance(cfg.quality_score_for_pass, (int, type(None))))

    def decode_calls(self, X, infos, calls, Y=None):
        """Rows of one batch (a list, or the finished text as bytes when no BAM is open) from the call records the GPU decode kernel left
        (include/clair_call.h): only text is made here.  With a BAM
        open, the candidates whose record says the reference would have consulted it are decoded again on the Python look-up path --
        which needs their probabilities (Y)."""
        from clair_amd import _hostapi
        cfg = self.cfg
        if len(calls) != len(infos):
            sys.exit("Inconsistent shape between input tensor and output predictions %d/%d" % (len(infos), len(calls)))
        if self.lookup.sam is None:      # nothing to splice in: the rows go to the file as the bytes the formatter wrote
            return _hostapi.format_calls(calls, infos, cfg.is_show_reference, cfg.is_haploid_precision_mode_enabled,
                                         cfg.is_haploid_sensitive_mode_enabled, cfg.quality_score_for_pass, self.arith == "numpy2", as_text=True)
        rows, status = _hostapi.format_calls(calls, infos, cfg.is_show_reference, cfg.is_haploid_precision_mode_enabled,
                                             cfg.is_haploid_sensitive_mode_enabled, cfg.quality_score_for_pass, self.arith == "numpy2",
                                             with_status=True)
        if not (status & 2).any():
            return rows
        if Y is None:
            raise ValueError("decode_calls: candidates that consult the BAM need their probabilities")
        out, at = [], 0
        Y = [np.asarray(a, dtype=np.float32) for a in Y]
        for i, st in enumerate(status):
            if st & 2:
                out.extend(self.decode_batch_py(X[i:i + 1], infos[i:i + 1], [a[i:i + 1] for a in Y]))
            elif st & 1:
                out.append(rows[at])
            at += int(st & 1)
        return out

    def decode_batch_py(self, X, infos, Y):
        gt21, genotype, len1, len2 = [np.asarray(a, dtype=np.float32) for a in Y]
        if len(gt21) != len(infos):
            sys.exit("Inconsistent shape between input tensor and output predictions %d/%d" % (len(infos), len(gt21)))
        if self.cfg.is_output_for_ensemble:
            return self._ensemble_rows(X, infos, gt21, genotype, len1, len2)
        n = len(infos)
        # candidates that reach the outcome computation: centre base in ACGTU and depth > 0
        centre = [inf[2][CENTER] for inf in infos]
        callable_ = np.array([c in task.BASIC_BASES for c in centre], dtype=bool)
        depth = (X[:, CENTER, :, CH_DEL] + X[:, CENTER, :, CH_REF]).sum(axis=1) if n else np.zeros(0, np.float32)
        ref_class = np.array([_REF_CLASS[task.IUPAC_TO_ACGT[c]] if ok else 0 for c, ok in zip(centre, callable_)],
                             dtype=np.int64)
        fams = OutcomeFamilies(gt21, genotype, len1, len2, ref_class) if n else None
        rows = []
        for i in range(n):
            if not callable_[i]:
                continue
            row = self._decode_one(i, X[i], infos[i], depth[i], fams, gt21[i], genotype[i], len1[i], len2[i])
            if row is not None:
                rows.append(row)
        return rows

    # -----------------------------------------------------------------------------------------
    def _decode_one(self, i, x, info, depth, fams, g_row, z_row, l1_row, l2_row):
        cfg = self.cfg
        contig, position, seq = info[0], int(info[1]), info[2]
        dbg = (lambda note: _debug_line(contig, position, g_row, z_row, l1_row, l2_row, note)) if cfg.is_debug else None
        if depth == 0:
            return dbg("Read Depth is zero") if dbg else None
        ref0 = seq[CENTER]
        flags = tuple(bool(f) for f in fams.flags[i])
        if flags[F_REF]:
            flags = _only(F_REF)
            ref = alt = task.IUPAC_TO_ACGT[ref0]
        elif flags[F_HOMO_SNP]:
            ref, alt = ref0, _homo_snp_alt(g_row, ref0)
        elif flags[F_HET_SNP]:
            ref, alt = ref0, _hetero_snp_alt(g_row, ref0)
        else:
            flags, ref, alt = _IndelResolver(fams, i, g_row, x, seq, contig, position, self.bases, self.lookup).run()
        is_ref = flags[F_REF]
        if not cfg.is_debug and ((not cfg.is_show_reference and is_ref) or (not is_ref and ref == alt)):
            return None
        is_multi = "," in alt
        hetero_call = (flags[F_HET_SNP] or flags[F_ACGT_INS] or flags[F_INSINS] or flags[F_ACGT_DEL] or flags[F_DELDEL])
        if cfg.is_haploid_precision_mode_enabled and (hetero_call or flags[F_INSDEL]):
            return None
        if cfg.is_haploid_sensitive_mode_enabled and is_multi:   # the reference's `elif` (:1082-1084)
            return None
        if is_ref:
            gt = task.GENOTYPE_STRINGS[task.HOMO_REFERENCE]
        elif flags[F_HOMO_SNP] or flags[F_HOMO_INS] or flags[F_HOMO_DEL]:
            gt = task.GENOTYPE_STRINGS[task.HOMO_VARIANT]
        elif hetero_call:
            gt = task.GENOTYPE_STRINGS[task.HETERO_VARIANT]
        if is_multi:
            gt = task.GENOTYPE_STRINGS[task.HETERO_VARIANT_MULTI]
        support = supporting_reads(x, flags, ref, alt, is_multi)
        if self.arith == "numpy2":
            af = (support + 0.0) / depth
        else:
            af = float(support) / float(depth)
        if af > 1:
            af = 1
        qual = quality_score(ref, alt, gt, g_row, z_row, self.arith)
        if cfg.is_haploid_precision_mode_enabled or cfg.is_haploid_sensitive_mode_enabled:
            gt = "1" if "1" in gt else "0"
        if cfg.quality_score_for_pass is None:
            filt = "."
        else:
            filt = "PASS" if qual >= cfg.quality_score_for_pass else "LowQual"
        if dbg:
            return dbg("Normal output" if not is_ref else "Reference")
        return "%s\t%d\t.\t%s\t%s\t%d\t%s\t%s\tGT:GQ:DP:AF\t%s:%d:%d:%.4f" % (
            contig, position, ref, alt, qual, filt, ".", gt, qual, depth, af)

    # -----------------------------------------------------------------------------------------
    @staticmethod
    def _ensemble_rows(X, infos, gt21, genotype, len1, len2):
        """call_var.py:950-1000: ctg, pos, seq, 1056 ints, 90 probabilities (%.6f), tab separated."""
        rows = []
        probs = np.concatenate([gt21, genotype, len1, len2], axis=1) if len(infos) else None
        for i, (contig, position, seq) in enumerate(infos):
            if seq[CENTER] not in task.BASIC_BASES:
                continue
            rows.append("\t".join([contig, position, seq] + list(X[i].flatten().astype(int).astype(str))
                                  + ["{:0.6f}".format(p) for p in probs[i]]))
        return rows


# =============================================================================================
# VCF writer
# =============================================================================================
HEADER_LINES = (
    '##fileformat=VCFv4.1',
    '##FILTER=<ID=PASS,Description="All filters passed">',
    '##FILTER=<ID=LowQual,Description="Confidence in this variant being real is below calling threshold.">',
    '##ALT=<ID=DEL,Description="Deletion">',
    '##ALT=<ID=INS,Description="Insertion of novel sequence">',
    '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">',
    '##INFO=<ID=LENGUESS,Number=.,Type=Integer,Description="Best guess of the indel length">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">',
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">',
    '##FORMAT=<ID=AF,Number=1,Type=Float,Description="Estimated allele frequency in the range (0,1)">',
)


class VcfWriter(object):
    """Output side of output_utilties_from (call_var.py:223-341): header + rows to --call_fn."""

    def __init__(self, output_file_path, sample_name="SAMPLE", reference_file_path=None, is_output_for_ensemble=False):
        self.fp = open(output_file_path, "w")
        self.sample_name = sample_name
        self.reference_file_path = reference_file_path
        self.is_output_for_ensemble = is_output_for_ensemble

    def write(self, text):
        print(text, file=self.fp)

    def write_rows(self, rows):
        if isinstance(rows, bytes):      # finished text of a batch (VariantDecoder.decode_calls): rows already '\n'-terminated
            if rows:
                self.fp.flush()                    # what went through the text layer so far comes first
                self.fp.buffer.write(rows)
        elif rows:
            self.fp.write("\n".join(rows))
            self.fp.write("\n")

    def write_header(self):
        if self.is_output_for_ensemble:                     # :305-306
            return
        for line in HEADER_LINES:
            self.write(line)
        if self.reference_file_path is not None:            # :323-329
            with open(self.reference_file_path + ".fai", "r") as fai:
                for row in fai:
                    cols = row.strip().split("\t")
                    self.write("##contig=<ID=%s,length=%s>" % (cols[0], cols[1]))
        self.write('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s' % self.sample_name)

    def close(self):
        self.fp.close()


# =============================================================================================
# drivers
# =============================================================================================
def call_variants(args, m, decoder, writer, batch_size=None, generator=None):
    """call_var.py:1312-1367.  The reference runs three threads per iteration (parse batch k+1, forward pass of batch k, decode +
    write batch k-1) and joins them every iteration; here the three stages are three long-lived threads joined by bounded queues --
    the batch source on one, the GPU (as many batches in flight as the model has slots) on the caller's, decode + write on a
    third -- so a stage never waits for the slowest of the other two to finish ITS current batch.  Rows appear in input order
    (one decoder, first in first out).  `generator` replaces the --tensor_fn reader with another source of (X, infos) batches
    (clair_amd.callVarBam hands pileup arrays over); a third element, the raw int16 counts of the same batch, is sent to the GPU
    instead of X when the model can take it.  A failure on either helper thread (sys.exit of a failing upstream stage included)
    stops the pipeline and is raised here: the reference's callVarBam checks its stages' exit codes (callVarBam.py:218-233)."""
    import queue
    writer.write_header()
    batch_size = batch_size or param.predictBatchSize
    logging.info("Calling variants ...")
    t0 = time()
    use_async = hasattr(m, "submit") and hasattr(m, "wait")
    n_slots = max(1, int(getattr(m, "n_slots", 2))) if use_async else 1
    # The decode on the device (include/clair_amd.h: clair_submit_ex): the outcome products and the arg-max run as one more kernel behind
    # the forward pass and 32-byte call records come back instead of 360 bytes of probabilities; the emit thread only formats text.
    # Same rows, byte for byte (tests/test_e2e_gpu.py).  CLAIR_AMD_DEVICE_DECODE=0 keeps the decode on the host.
    device_decode = (use_async and hasattr(m, "submit_calls") and getattr(decoder, "native_applies", lambda: False)()
                     and os.environ.get("CLAIR_AMD_DEVICE_DECODE", "1") != "0")
    keep_probabilities = device_decode and decoder.lookup.sam is not None      # candidates that consult the BAM are decoded again from them
    pool = None
    if generator is None:      # with the decode on the device nobody on the host reads the float32 tensor (unless a BAM is consulted)
        lean = device_decode and not keep_probabilities
        if lean and hasattr(m, "pinned_buffer"):
            # binary records are read straight into page-locked buffers of the engine and go to the GPU from there (a strided 2-D copy
            # of the counts column): no pass over the batch on the host.  A buffer returns to the pool when its batch leaves the GPU.
            from clair_amd import tensor_binary
            pool = tensor_binary.BufferPool([m.pinned_buffer(batch_size * tensor_binary.RECORD.itemsize) for _ in range(2 * n_slots + 4)])
        generator = ingest.tensor_generator_from(args.tensor_fn, batch_size, with_input=not lean, record_buffers=pool)
    loaded = queue.Queue(maxsize=n_slots + 2)      # batches parsed ahead
    finished = queue.Queue(maxsize=n_slots + 2)    # (batch, prediction) waiting to be decoded and written
    failures = []                                  # exc_info of a stage that died on its helper thread
    stop = Event()
    END = object()

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def get(q):
        while True:
            try:
                return q.get(timeout=0.1)
            except queue.Empty:
                if stop.is_set():
                    return END

    def load():
        try:
            for batch in generator:
                if not put(loaded, batch):
                    return
        except BaseException:                      # a thread would swallow it and the consumer would read "end of input"
            failures.append(sys.exc_info())
            stop.set()
        put(loaded, END)

    def emit():
        try:
            while True:
                item = get(finished)
                if item is END:
                    return
                batch, prediction = item
                if device_decode:
                    calls, probabilities = prediction if keep_probabilities else (prediction, None)
                    writer.write_rows(decoder.decode_calls(batch[0], batch[1], calls, probabilities))
                else:
                    writer.write_rows(decoder.decode_batch(batch[0], batch[1], prediction))
        except BaseException:
            failures.append(sys.exc_info())
            stop.set()

    def reraise():
        if failures:
            _, exc, tb = failures[0]
            raise exc.with_traceback(tb)

    loader, emitter = Thread(target=load, daemon=True), Thread(target=emit, daemon=True)
    loader.start()
    emitter.start()
    inflight = []                                  # (slot, batch) submitted to the GPU, oldest first

    def retire():
        slot, batch = inflight.pop(0)
        prediction = m.wait(slot)
        m.prediction = (prediction[1] if keep_probabilities else None) if device_decode else prediction
        if pool is not None and len(batch) > 3:
            pool.put(batch[3])                         # the GPU has the batch: its record buffer may be refilled
        put(finished, (batch[:2], prediction))

    try:
        k = 0
        while not stop.is_set():
            current = get(loaded)
            if current is END:
                break
            if not use_async:
                prediction = m.predict(current[0])
                m.prediction = prediction
                put(finished, (current, prediction))
                continue
            if len(inflight) == n_slots:
                retire()
            slot = k % n_slots
            has_counts = len(current) > 2 and current[2] is not None
            if device_decode:
                from clair_amd import _hostapi
                m.submit_calls(slot, current[2] if has_counts else current[0], _hostapi.centre_bytes(current[1]), counts=has_counts,
                               with_probabilities=keep_probabilities)
            elif has_counts and hasattr(m, "submit_counts"):
                m.submit_counts(slot, current[2])
            else:
                m.submit(slot, current[0])
            inflight.append((slot, current))
            k += 1
        while inflight and not stop.is_set():
            retire()
    except BaseException:
        stop.set()
        for slot, _ in inflight:                   # leave the engine usable for the caller's next region (callVarBamParallel --run goes on)
            try:
                m.wait(slot)
            except Exception:
                pass
        raise
    finally:
        if pool is not None:
            pool.close()
        put(finished, END)
        emitter.join()
        stop.set()
        loader.join(timeout=5.0)
    reraise()
    logging.info("Total time elapsed: %.2f s" % (time() - t0))


def call_variants_with_probabilities_input(args, decoder, writer, stream=None):
    """call_var.py:1276-1309: rows `ctg pos seq 1056 ints 90 probabilities` on stdin -> VCF."""
    writer.write_header()
    logging.info("Output variants ...")
    t0 = time()
    shape = (param.no_of_positions, param.matrixRow, param.matrixNum)
    nvals = param.input_tensor_size
    for row in (stream if stream is not None else sys.stdin):
        cols = row.split("\t")
        x = np.reshape(np.array(cols[3:3 + nvals], dtype=np.float32), shape)
        probs = np.array(cols[3 + nvals:], dtype=np.float32)
        Y = [probs[None, 0:21], probs[None, 21:24], probs[None, 24:24 + shape[0]], probs[None, 24 + shape[0]:]]
        writer.write_rows(decoder.decode_batch(x[None], [[cols[0], cols[1], cols[2]]], Y))
    logging.info("Total time elapsed: %.2f s" % (time() - t0))


def Run(args):
    """call_var.py:173-220."""
    ingest.setup_environment()
    if args.threads is None:
        if args.tensor_fn == "PIPE":
            param.NUM_THREADS = 4
    else:
        param.NUM_THREADS = max(args.threads - 1, 1)
    config = OutputConfig(
        is_show_reference=args.showRef, is_debug=args.debug,
        is_haploid_precision_mode_enabled=args.haploid_precision,
        is_haploid_sensitive_mode_enabled=args.haploid_sensitive,
        is_output_for_ensemble=args.output_for_ensemble, quality_score_for_pass=args.qual)
    lookup = AlignmentLookup(args.bam_fn, args.ref_fn)
    decoder = VariantDecoder(config, lookup, always_use_bam=args.pysam_for_all_indel_bases, arith=args.arith)
    writer = VcfWriter(args.call_fn, args.sampleName, args.ref_fn, args.output_for_ensemble)
    try:
        if args.input_probabilities:
            call_variants_with_probabilities_input(args, decoder, writer)
            return
        if args.activation_only:
            # dead path in the reference as well: its summary writer factory returns None
            # (clair/model.py:1053-1062), so log_activation returns immediately (call_var.py:1239-1245)
            return
        from clair_amd.model import Clair
        batch = args.batch_size or param.engineBatchSize
        try:
            m = Clair(device=args.device, max_batch=batch, n_slots=param.pipeline_slots())
            m.init()
            m.restore_parameters(os.path.abspath(args.chkpnt_fn))
        except Exception as exc:   # C-ABI errors surface as messages + non-zero exit
            sys.exit("[ERROR] %s" % exc)
        try:
            call_variants(args, m, decoder, writer, batch)
        finally:
            m.close()
    finally:
        writer.close()
        lookup.close()


def build_parser():
    """Same flags and defaults as call_var.py:1370-1429, plus --batch_size / --device / --arith."""
    parser = ArgumentParser(description="Call variants using a trained model and tensors of candididate variants")
    parser.add_argument('--tensor_fn', type=str, default="PIPE", help="Tensor input, use PIPE for standard input")
    parser.add_argument('--chkpnt_fn', type=str, default=None, help="Input a checkpoint for testing")
    parser.add_argument('--call_fn', type=str, default=None, help="Output variant predictions")
    parser.add_argument('--bam_fn', type=str, default="bam.bam", help="BAM file input, default: %(default)s")
    parser.add_argument('--qual', type=int, default=None,
                        help="If set, variant with equal or higher quality will be marked PASS, or LowQual otherwise, optional")
    parser.add_argument('--sampleName', type=str, default="SAMPLE", help="Define the sample name to be shown in the VCF file")
    parser.add_argument('--showRef', action='store_true', help="Show reference calls, optional")
    parser.add_argument('--debug', action='store_true', help="Debug mode, optional")
    parser.add_argument('--ref_fn', type=str, default=None,
                        help="Reference fasta file input, optional, print contig tags in the VCF header if set")
    parser.add_argument('--threads', type=int, default=None, help="Number of threads, optional")
    parser.add_argument('--activation_only', action='store_true', help="Output activation only, no prediction")
    parser.add_argument('--max_plot', type=int, default=10,
                        help="The maximum number of plots output, negative number means no limit (plot all), default: %(default)s")
    parser.add_argument('--log_path', type=str, nargs='?', default=None, help="The path for tensorflow logging, default: %(default)s")
    parser.add_argument('-p', '--parallel_level', type=int, default=2,
                        help="The level of parallelism in plotting (currently available: 0, 2), default: %(default)s")
    parser.add_argument('--fast_plotting', action='store_true', help="Enable fast plotting.")
    parser.add_argument('-w', '--workers', type=int, default=8, help="The number of workers in plotting, default: %(default)s")
    parser.add_argument('--pysam_for_all_indel_bases', action='store_true',
                        help="Always using pysam for outputting indel bases, optional")
    parser.add_argument('--haploid_precision', action='store_true',
                        help="call haploid instead of diploid (output homo-variant only)")
    parser.add_argument('--haploid_sensitive', action='store_true',
                        help="call haploid instead of diploid (output non-multi-variant only)")
    parser.add_argument('--input_probabilities', action='store_true',
                        help="Accept probabilities as input, using those probabilities to call variant")
    parser.add_argument('--output_for_ensemble', action='store_true', help="Output for ensemble")
    # additions of this implementation
    parser.add_argument('--batch_size', type=int, default=None,
                        help="Candidates per forward pass, default: %d" % param.engineBatchSize)
    parser.add_argument('--device', type=int, default=0, help="HIP device ordinal, default: %(default)s")
    parser.add_argument('--arith', type=str, default="legacy", choices=("legacy", "numpy2"),
                        help="QUAL/AF arithmetic: float64 as under the reference's NumPy 1.x (legacy) or float32 (numpy2)")
    return parser


def main():
    parser = build_parser()
    args = parser.parse_args()
    if len(sys.argv[1:]) == 0:
        parser.print_help()
        sys.exit(1)
    Run(args)


if __name__ == "__main__":
    main()
