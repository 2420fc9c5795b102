#!/usr/bin/env python3
"""Which VCF calls are decided by a margin that float32 cannot resolve -- the analysis behind the "bit-identical GT" contract.

The reference's decode (clair/call_var.py:589-690, 733-762) forms 1 179 float32 products of the four heads' probabilities and takes
the largest, family by family in its if/elif order.  Two evaluations of the same network that agree to 1e-5 on every probability
give the same call unless the winner and its runner-up are closer than the probabilities' own error moves them.  This module says,
for a candidate, how close they are and how far an absolute perturbation eps of the probabilities can move them:

  P_k                      the products, in the decoder's own float32 operand order (checked bit for bit against
                           clair_amd.call_var.OutcomeFamilies by tests/test_gt_ties.py);
  R_k = sum_i 1 / p_i      over the factors of product k: |dP_k| / P_k <= eps * R_k to first order for |dp_i| <= eps;
  ambiguous at eps         (P_1 - P_2) / P_1 <= eps * (R_1 + R_2) + ROUNDING        (winner 1, runner-up 2)
  ROUNDING = 6 * 2^-24     three float32 multiplications per product, two products: what the operand order alone can move.

`eps = 0` are the candidates a different multiplication order would already decide differently (what TensorFlow's multithreaded
Eigen does not promise either); `eps = 1e-5` the candidates that ANY implementation within the stated tolerance may call
differently.  A GT flip between the HIP path and the float32 oracle is excused only if (tests/test_parity_gpu.py):
  * the pair (HIP's winner A, the oracle's winner B) is ambiguous at eps = max |p_hip - p_o32| of that candidate, and at
    eps = max |p_o32 - p_o64| -- the float32 oracle's own distance from the float64 evaluation: float32 cannot decide the pair;
  * the float64 evaluation decides it the HIP way (or is itself within ROUNDING of a tie).

TEST / MEASUREMENT INFRASTRUCTURE (used by tools/gt_concordance.py, bench.py's concordance leg and tests/); nothing under clair_amd/ imports it.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from clair_amd import call_var as cvar, task  # noqa: E402

ROUNDING = 6.0 * 2.0 ** -24
FAMILY_SIZES = (1, 4, 6, 16, 64, 256, 16, 64, 240, 512)        # in the decoder's if/elif order (clair_amd/call_var.py: F_REF .. F_INSDEL)
FAMILY_NAMES = ("Ref", "HomoSNP", "HeteroSNP", "HomoIns", "ACGT+Ins", "InsIns", "HomoDel", "ACGT+Del", "DelDel", "InsDel")
OFFSETS = np.concatenate([[0], np.cumsum(FAMILY_SIZES)]).astype(np.int64)
N_OUTCOMES = int(OFFSETS[-1])                                    # 1 179
assert N_OUTCOMES == 1179


class _PR(object):
    """A product and its relative sensitivity: (a * b).r = a.r + b.r."""
    __slots__ = ("p", "r")

    def __init__(self, p, r):
        self.p, self.r = p, r

    def __mul__(self, other):
        return _PR(self.p * other.p, None if self.r is None else self.r + other.r)

    def __getitem__(self, key):
        return _PR(self.p[key], None if self.r is None else self.r[key])

    def reshape(self, *shape):
        return _PR(self.p.reshape(*shape), None if self.r is None else self.r.reshape(*shape))


def _larger(a, b):
    take = a.p >= b.p                        # np.maximum(a, b): the branch float32 takes carries its own factors
    return _PR(np.where(take, a.p, b.p), None if a.r is None else np.where(take, a.r, b.r))


def outcome_table(Y, ref_class, dtype=np.float32, sensitivities=True):
    """(P [n, 1179], R [n, 1179]) for the four heads' probabilities Y = (gt21, genotype, len1, len2).  `dtype` float32: the decoder's
    products, bit for bit, operand order of clair/call_var.py:589-690 as restated in clair_amd/call_var.py: OutcomeFamilies;
    float64: the same products without float32 rounding (for the float64 evaluation's margins).  R is float32 either way."""
    gt21, genotype, len1, len2 = [np.asarray(a, dtype=dtype) for a in Y]
    n = gt21.shape[0]
    rows = np.arange(n)

    def pr(a):
        return _PR(a, (1.0 / np.maximum(a.astype(np.float64), 1e-300)).astype(np.float32) if sensitivities else None)

    g = pr(gt21)
    p_ref, p_hom, p_het = pr(genotype[:, 0]), pr(genotype[:, 1]), pr(genotype[:, 2])
    zero = pr(len1[:, 16]) * pr(len2[:, 16])
    ins1, ins2 = pr(len1[:, cvar._INS_COLS]), pr(len2[:, cvar._INS_COLS])
    del1, del2 = pr(len1[:, cvar._DEL_COLS]), pr(len2[:, cvar._DEL_COLS])
    z1, z2 = pr(len1[:, 16:17]), pr(len2[:, 16:17])
    N = (slice(None), None)
    fam = [None] * cvar.N_FAMILIES
    fam[cvar.F_REF] = ((zero * p_ref) * g[rows, ref_class])[N]
    fam[cvar.F_HOMO_SNP] = (zero * p_hom)[N] * g[:, list(task.HOMO_SNP_IDX)]
    fam[cvar.F_HET_SNP] = (zero * p_het)[N] * g[:, list(task.HETERO_SNP_IDX)]
    fam[cvar.F_HOMO_INS] = (ins1 * ins2) * (p_hom * g[:, task.IDX_INSINS])[N]
    fam[cvar.F_INSINS] = ((ins1[:, :, None] * ins2[:, None, :]) * (p_het * g[:, task.IDX_INSINS])[:, None, None]).reshape(n, -1)
    one_ins = _larger(z1 * ins2, ins1 * z2)
    fam[cvar.F_ACGT_INS] = ((one_ins[:, :, None] * g[:, None, list(task.INS_BASE_IDX)]) * p_het[:, None, None]).reshape(n, -1)
    fam[cvar.F_HOMO_DEL] = (del1 * del2) * (p_hom * g[:, task.IDX_DELDEL])[N]
    fam[cvar.F_DELDEL] = ((del1[:, :, None] * del2[:, None, :]) * (p_het * g[:, task.IDX_DELDEL])[:, None, None])[:, cvar._OFF_DIAG]
    one_del = _larger(z1 * del2, del1 * z2)
    fam[cvar.F_ACGT_DEL] = ((one_del[:, :, None] * g[:, None, list(task.DEL_BASE_IDX)]) * p_het[:, None, None]).reshape(n, -1)
    e3 = (p_het * g[:, task.IDX_INSDEL])[:, None, None]
    a = (ins1[:, :, None] * del2[:, None, :]) * e3          # len1 = +i, len2 = -j
    b = (del1[:, :, None] * ins2[:, None, :]) * e3          # len1 = -i, len2 = +j
    fam[cvar.F_INSDEL] = _PR(np.stack([a.p, b.p], axis=-1).reshape(n, -1), np.stack([a.r, b.r], axis=-1).reshape(n, -1) if sensitivities else None)
    return np.concatenate([f.p for f in fam], axis=1), (np.concatenate([f.r for f in fam], axis=1) if sensitivities else None)


def ref_classes(infos):
    """gt21 class of the homozygous-reference call per candidate (0 where the centre base is not callable; those rows are skipped upstream)."""
    out = np.zeros(len(infos), dtype=np.int64)
    for i, inf in enumerate(infos):
        c = inf[2][cvar.CENTER]
        if c in task.BASIC_BASES:
            out[i] = cvar._REF_CLASS[task.IUPAC_TO_ACGT[c]]
    return out


def family_of(index):
    return int(np.searchsorted(OFFSETS, index, side="right") - 1)


def outcome_name(index):
    k = family_of(index)
    return "%s[%d]" % (FAMILY_NAMES[k], int(index - OFFSETS[k]))


def first_iteration_margins(P, R):
    """Winner (first maximum in the decoder's priority order) and runner-up of the FIRST arg-max of every candidate:
    (winner, runner_up, relative margin (P1 - P2) / P1, R1 + R2)."""
    n = P.shape[0]
    rows = np.arange(n)
    w = P.argmax(axis=1)
    p1 = P[rows, w].astype(np.float64)
    keep = P[rows, w].copy()
    P[rows, w] = -1.0
    r = P.argmax(axis=1)
    p2 = P[rows, r].astype(np.float64)
    P[rows, w] = keep
    margin = (p1 - p2) / np.maximum(p1, 1e-300)
    return w, r, margin, (R[rows, w].astype(np.float64) + R[rows, r].astype(np.float64)) if R is not None else None


def ambiguous(margin, rsum, eps):
    return margin <= eps * rsum + ROUNDING


def _factor_table():
    """For each of the 1 179 outcomes, the positions of its four factors in the packed 90-vector (gt21 | genotype | len1 | len2) -- what
    outcome_table multiplies, as data: FACT[k] = four positions, ALT[k] = the two positions that replace FACT[k][:2] when the other branch
    of np.maximum(z1 * x2, x1 * z2) is the larger one (-1: no such branch).  The reference-call outcome's gt21 factor depends on the
    candidate (its reference base): FACT[0][3] = -1, filled in per candidate."""
    G, Z, L1, L2 = (lambda i: i), (lambda c: 21 + c), (lambda c: 24 + c), (lambda c: 57 + c)
    ins, dele = [int(c) for c in cvar._INS_COLS], [int(c) for c in cvar._DEL_COLS]
    fact, alt = [], []

    def add(f, a=(-1, -1)):
        fact.append(f)
        alt.append(a)
    add((L1(16), L2(16), Z(0), -1))
    for g in task.HOMO_SNP_IDX:
        add((L1(16), L2(16), Z(1), G(g)))
    for g in task.HETERO_SNP_IDX:
        add((L1(16), L2(16), Z(2), G(g)))
    for k in range(16):
        add((L1(ins[k]), L2(ins[k]), Z(1), G(task.IDX_INSINS)))
    for k in range(16):
        for g in task.INS_BASE_IDX:
            add((L1(16), L2(ins[k]), Z(2), G(g)), (L1(ins[k]), L2(16)))
    for i in range(16):
        for j in range(16):
            add((L1(ins[i]), L2(ins[j]), Z(2), G(task.IDX_INSINS)))
    for k in range(16):
        add((L1(dele[k]), L2(dele[k]), Z(1), G(task.IDX_DELDEL)))
    for k in range(16):
        for g in task.DEL_BASE_IDX:
            add((L1(16), L2(dele[k]), Z(2), G(g)), (L1(dele[k]), L2(16)))
    for i in range(16):
        for j in range(16):
            if i != j:
                add((L1(dele[i]), L2(dele[j]), Z(2), G(task.IDX_DELDEL)))
    for i in range(16):
        for j in range(16):
            add((L1(ins[i]), L2(dele[j]), Z(2), G(task.IDX_INSDEL)))
            add((L1(dele[i]), L2(ins[j]), Z(2), G(task.IDX_INSDEL)))
    fact, alt = np.array(fact, dtype=np.int64), np.array(alt, dtype=np.int64)
    assert fact.shape == (N_OUTCOMES, 4) and alt.shape == (N_OUTCOMES, 2)
    return fact, alt


_FACT, _ALT = _factor_table()


def sensitivity_of(packed, k, ref_class):
    """R_k = sum 1 / p over the factors of outcome k[i] of candidate i, from the packed [n, 90] float32 probabilities -- the same numbers as
    outcome_table's R at those positions (tests/test_gt_ties.py), without building the other 1 178 columns."""
    n = packed.shape[0]
    rows = np.arange(n)
    f = _FACT[k].copy()                                   # [n, 4]
    f[:, 3] = np.where(f[:, 3] < 0, ref_class, f[:, 3])
    a = _ALT[k]
    has = a[:, 0] >= 0
    if has.any():                                         # np.maximum(z1 * x2, x1 * z2): the branch float32 takes
        r_ = rows[has]
        first = packed[r_, f[has, 0]] * packed[r_, f[has, 1]]
        other = packed[r_, a[has, 0]] * packed[r_, a[has, 1]]
        swap = other > first
        idx = np.flatnonzero(has)[swap]
        f[idx, 0], f[idx, 1] = a[idx, 0], a[idx, 1]
    p = np.maximum(packed[rows[:, None], f].astype(np.float64), 1e-300)
    return (1.0 / p).sum(axis=1)


def near_tie_counts(Y, infos, eps_list, step=1024, callable_only=True):   # small steps: the temporaries stay in cache (8192: 20x slower, page faults)
    """How many candidates of a batch are ambiguous at each eps (first arg-max only: a candidate whose first choices cannot be
    written as REF / ALT walks on to later maxima, which this count does not follow -- flips are analysed exactly, see analyse_flip).
    The products of all 1 179 outcomes for everybody, the sensitivities of winner and runner-up only (sensitivity_of).
    -> ({eps: count}, boolean mask per eps, relative margins)."""
    n = len(infos)
    rc = ref_classes(infos)
    ok = np.array([inf[2][cvar.CENTER] in task.BASIC_BASES for inf in infos], dtype=bool) if callable_only else np.ones(n, bool)
    margins = np.empty(n, np.float64)
    rsum = np.empty(n, np.float64)
    packed = np.concatenate([np.asarray(a, dtype=np.float32) for a in Y], axis=1)
    for i in range(0, n, step):
        sl = slice(i, i + step)
        P, _ = outcome_table([a[sl] for a in Y], rc[sl], sensitivities=False)
        w, r, margins[sl], _ = first_iteration_margins(P, None)
        rsum[sl] = sensitivity_of(packed[sl], w, rc[sl]) + sensitivity_of(packed[sl], r, rc[sl])
    masks = {eps: ok & ambiguous(margins, rsum, eps) for eps in eps_list}
    return {eps: int(m.sum()) for eps, m in masks.items()}, masks, margins


class _RecordingResolver(cvar._IndelResolver):
    """The decoder's removal loop, remembering which outcome it took last and how many it tried."""

    def __init__(self, *a):
        cvar._IndelResolver.__init__(self, *a)
        self.last, self.tried = None, 0

    def _pop_first(self, k, value):
        idx = cvar._IndelResolver._pop_first(self, k, value)
        self.last, self.tried = (k, idx), self.tried + 1
        return idx


def final_winner(decoder, x, info, Y1):
    """Index (into the 1 179 outcomes) of the outcome the decoder's call for ONE candidate rests on, and the outcomes its removal loop
    tried before settling (0: decided by the first arg-max).  (-1, 0) when the candidate produces no call at all (centre base not in
    ACGTU, read depth 0)."""
    gt21, genotype, len1, len2 = [np.asarray(a, dtype=np.float32).reshape(1, -1) for a in Y1]
    seq = info[2]
    if seq[cvar.CENTER] not in task.BASIC_BASES:
        return -1, 0
    depth = (x[cvar.CENTER, :, cvar.CH_DEL] + x[cvar.CENTER, :, cvar.CH_REF]).sum()
    if depth == 0:
        return -1, 0
    rc = ref_classes([info])
    fams = cvar.OutcomeFamilies(gt21, genotype, len1, len2, rc)
    flags = fams.flags[0]

    def top_of(k, vals=None):
        v = fams.fam[k][0] if vals is None else vals
        return int(OFFSETS[k] + int(np.argmax(v)))

    if flags[cvar.F_REF]:
        return int(OFFSETS[cvar.F_REF]), 0
    if flags[cvar.F_HOMO_SNP]:
        return top_of(cvar.F_HOMO_SNP), 0
    if flags[cvar.F_HET_SNP]:
        return top_of(cvar.F_HET_SNP), 0
    r = _RecordingResolver(fams, 0, gt21[0], x, seq, info[0], int(info[1]), decoder.bases, decoder.lookup)
    out_flags, _, _ = r.run()
    if out_flags[cvar.F_REF]:
        return int(OFFSETS[cvar.F_REF]), r.tried
    for k in (cvar.F_HOMO_SNP, cvar.F_HET_SNP):
        if out_flags[k]:
            return top_of(k), r.tried
    k, idx = r.last
    return int(OFFSETS[k] + idx), r.tried - 1


def pair_margin(Y1, info, a, b, dtype=np.float32):
    """For ONE candidate and two outcome indices: (P_a, P_b, R_a + R_b) from the probabilities Y1 evaluated in `dtype`."""
    P, R = outcome_table([np.asarray(v).reshape(1, -1) for v in Y1], ref_classes([info]), dtype=dtype)
    return float(P[0, a]), float(P[0, b]), float(R[0, a]) + float(R[0, b])


def analyse_flip(decoder, x, info, hip, o32, o64):
    """Everything the contract asks of ONE candidate whose call differs between the HIP probabilities and the float32 oracle's.
    hip / o32: four float32 rows; o64: four float64 rows.  -> dict (JSON-friendly)."""
    o64r = [np.asarray(a, dtype=np.float64).astype(np.float32) for a in o64]
    a, tried_a = final_winner(decoder, x, info, hip)
    b, tried_b = final_winner(decoder, x, info, o32)
    c, _ = final_winner(decoder, x, info, o64r)
    eps_hip = max(float(np.abs(np.asarray(h, np.float64) - np.asarray(o, np.float64)).max()) for h, o in zip(hip, o32))
    eps_32 = max(float(np.abs(np.asarray(o, np.float64) - np.asarray(d, np.float64)).max()) for o, d in zip(o32, o64))
    out = {"hip_outcome": outcome_name(a) if a >= 0 else None, "oracle32_outcome": outcome_name(b) if b >= 0 else None,
           "oracle64_outcome": outcome_name(c) if c >= 0 else None, "hip_index": a, "oracle32_index": b, "oracle64_index": c,
           "outcomes_tried": [tried_a, tried_b], "eps_hip_vs_o32": eps_hip, "eps_o32_vs_o64": eps_32}
    if a < 0 or b < 0 or a == b:
        out.update({"ambiguous_at_eps_hip": False, "ambiguous_at_eps_o32": False, "float64_sides_with_hip": a == c, "float64_tie": False})
        return out
    pa, pb, rsum = pair_margin(o32, info, a, b)                           # the float32 oracle: B over A
    m32 = (pb - pa) / max(pb, 1e-300)
    qa, qb, _ = pair_margin(o64, info, a, b, dtype=np.float64)            # the float64 evaluation, products in float64
    m64 = (qa - qb) / max(qa, qb, 1e-300)                                 # > 0: A (the HIP call) is ahead
    ha, hb, _ = pair_margin(hip, info, a, b)
    out.update({"margin_o32": m32, "margin_o64_towards_hip": m64, "margin_hip": (ha - hb) / max(ha, 1e-300), "sensitivity": rsum,
                "ambiguous_at_eps_hip": bool(ambiguous(m32, rsum, eps_hip)), "ambiguous_at_eps_o32": bool(ambiguous(m32, rsum, eps_32)),
                "float64_sides_with_hip": bool(m64 > 0), "float64_tie": bool(abs(m64) <= ROUNDING),
                "margin_o32_ulps": float(m32 * 2.0 ** 23)})
    return out


def flip_is_excused(rec):
    """The contract (module docstring): float32 cannot decide the pair, and float64 decides it the HIP way (or ties)."""
    return bool(rec["ambiguous_at_eps_hip"] and rec["ambiguous_at_eps_o32"] and (rec["float64_sides_with_hip"] or rec["float64_tie"]))
