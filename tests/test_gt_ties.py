"""tools/gt_ties.py and the concordance tool on the CPU: the analysis behind the "bit-identical GT" contract.

No GPU here: the "device" of the concordance run is a stand-in whose probabilities are the float64 oracle's rounded to float32 -- a
second evaluation of the same network that agrees with the float32 oracle to ~1e-6, which is exactly the situation the contract is about.
"""
import json
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gt_concordance  # noqa: E402
import gt_ties  # noqa: E402
from clair_amd import call_var as cvar, synth, task, weights  # noqa: E402
from oracle import c_oracle  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "gt_ties.npz")


def _random_heads(n, seed, peaky=3.0):
    rng = np.random.default_rng(seed)

    def soft(k):
        z = rng.normal(size=(n, k)) * peaky
        e = np.exp(z - z.max(axis=1, keepdims=True))
        return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)

    return [soft(21), soft(3), soft(33), soft(33)], rng.integers(0, 4, n).astype(np.int64) * 0 + np.array([task.GT21_INDEX[b + b] for b in "ACGT"])[rng.integers(0, 4, n)]


def test_outcome_table_is_the_decoders_products_bit_for_bit():
    Y, rc = _random_heads(300, 1)
    P, R = gt_ties.outcome_table(Y, rc)
    fams = cvar.OutcomeFamilies(*Y, rc)
    assert P.shape == (300, 1179) and P.dtype == np.float32 and R.shape == P.shape
    assert np.array_equal(P, np.concatenate(fams.fam, axis=1))
    assert [f.shape[1] for f in fams.fam] == list(gt_ties.FAMILY_SIZES)
    # the first maximum in concatenation order is the decoder's choice: family priority, then list order
    w, r, margin, rsum = gt_ties.first_iteration_margins(P.copy(), R)
    assert np.array_equal(P[np.arange(300), w], fams.best) and (margin >= 0).all() and (w != r).all()
    fam_of_w = np.array([gt_ties.family_of(i) for i in w])
    assert np.array_equal(fam_of_w, fams.flags.argmax(axis=1))


def test_sensitivities_are_the_derivatives_of_the_products():
    """R_k = sum over the factors of 1 / p_i: scaling ONE probability by (1 + h) scales product k by (1 + h)^(its multiplicity in k);
    summed over all 90 probabilities, (P'/P - 1) / (h p_i) must reproduce R_k."""
    Y, rc = _random_heads(12, 2, peaky=1.5)
    Y = [np.maximum(a, 1e-4).astype(np.float64) for a in Y]
    P, R = gt_ties.outcome_table(Y, rc, dtype=np.float64)
    h = 1e-6
    num = np.zeros_like(P)
    for head in range(4):
        for col in range(Y[head].shape[1]):
            Y2 = [a.copy() for a in Y]
            Y2[head][:, col] *= 1 + h
            P2, _ = gt_ties.outcome_table(Y2, rc, dtype=np.float64)
            num += np.abs(P2 / P - 1) / (h * Y[head][:, col:col + 1])
    # np.maximum(z1 * ins2, ins1 * z2): the derivative follows the branch taken, as R does (entries within 1e-6 of a branch tie are skipped)
    ok = np.abs(num - R) <= 2e-3 * R
    assert ok.mean() > 0.999, float(ok.mean())


def test_exact_and_near_ties_are_counted():
    Y, rc = _random_heads(64, 3)
    infos = [("chr1", str(100 + i), "A" * 16 + "ACGT"[i % 4] + "C" * 16) for i in range(64)]
    rc = gt_ties.ref_classes(infos)
    # candidate 0: an exact tie between two heterozygous SNP outcomes; candidate 1: the same pair one float32 step apart
    for i, bump in ((0, 0.0), (1, 2.0 ** -24)):
        g = np.full(21, 1e-3, np.float32)
        g[task.GT21_INDEX["AC"]] = 0.4
        g[task.GT21_INDEX["AG"]] = np.float32(0.4) * np.float32(1 + bump * 4)
        Y[0][i] = g / g.sum()
        Y[1][i] = (0.05, 0.05, 0.9)
        for L in (Y[2], Y[3]):
            L[i] = 1e-4
            L[i, 16] = 1 - 32e-4
    counts, masks, margins = gt_ties.near_tie_counts(Y, infos, (0.0, 1e-5))
    assert masks[0.0][0] and masks[0.0][1] and margins[0] == 0 and 0 < margins[1] < 1e-6
    assert counts[0.0] >= 2 and counts[1e-5] >= counts[0.0]
    # a clear winner is not ambiguous even at the tolerance
    Y[0][2] = np.full(21, 1e-3, np.float32)
    Y[0][2, rc[2]] = 0.98
    Y[1][2] = (0.98, 0.01, 0.01)
    Y[2][2], Y[3][2] = Y[2][0], Y[3][0]
    counts, masks, _ = gt_ties.near_tie_counts(Y, infos, (1e-5,))
    assert not masks[1e-5][2]


def test_final_winner_names_the_outcome_the_row_was_written_from():
    w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
    raw, infos = synth.synthetic_candidates(400, "ont", seed=5, start=5000)
    x = synth.to_model_input(raw)
    Y = c_oracle.forward(w, x)
    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None), native=False)
    seen = set()
    for i in range(400):
        idx, tried = gt_ties.final_winner(dec, x[i], infos[i], [a[i] for a in Y])
        rows = dec.decode_batch(x[i:i + 1], infos[i:i + 1], [a[i:i + 1] for a in Y])
        if idx < 0:
            assert not rows
            continue
        fam = gt_ties.family_of(idx)
        seen.add(fam)
        gt = rows[0].split("\t")[-1].split(":")[0]
        if fam == cvar.F_REF:
            assert gt == "0/0"
        elif fam in (cvar.F_HOMO_SNP, cvar.F_HOMO_INS, cvar.F_HOMO_DEL):
            assert gt in ("1/1", "1/2")          # 1/2: a multi-allelic ALT overrides the genotype string (call_var.py:1087-1094)
        else:
            assert gt in ("0/1", "1/2")
        assert tried >= 0
    assert len(seen) >= 5          # the synthetic heads exercise most families


class _Float64AsDevice(object):
    """Stand-in for the engine: the float64 oracle rounded to float32."""

    def __init__(self, w):
        self.w = w

    def predict(self, x):
        return [a.astype(np.float32) for a in c_oracle.forward(self.w, x, dtype=np.float64)]


def test_concordance_reports_near_ties_flips_and_honours_its_deadline(tmp_path):
    w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
    eng = _Float64AsDevice(w)
    r = gt_concordance.concordance(eng, w, "illumina", 3000, 777, chunk=1500, log=lambda *a: None, tightest=3)
    assert r["candidates"] == 3000 and "truncated" not in r and r["vcf_rows"] > 2900 and r["max_abs_dp"] < 1e-5
    near = r["near_ties"]
    assert sorted(near) == ["eps_0", "eps_1e-05", "eps_3e-06"] and near["eps_0"] <= near["eps_3e-06"] <= near["eps_1e-05"]
    assert r["gt_flips"] == len(r["flips"]) == r["flips_excused"] + r["flips_not_excused"]
    assert len(r["tightest"]) == 3 and all(t["kind"] == "tight" and 1e-5 in t["near_tie_at"] for t in r["tightest"])
    for f in r["flips"]:           # the stand-in IS the float64 evaluation: every flip is decided "its" way, and must be a pair float32 cannot decide
        assert f["float64_sides_with_hip"] and f["ambiguous_at_eps_o32"] and f["excused"], gt_concordance.strip_arrays(f)
    path = str(tmp_path / "ties.npz")
    gt_concordance.save_ties(path, [r])
    with np.load(path) as z:
        n = len(r["flips"]) + 3
        assert z["x"].shape == (n, 33, 8, 4) and z["hip"].shape == (n, 90) and z["o64"].dtype == np.float64
        assert len(json.loads(str(z["meta"]))) == n
    # a deadline in the past: the first chunk still runs (a count of zero candidates says nothing), the second does not start
    r = gt_concordance.concordance(eng, w, "ont", 3000, 777, chunk=1500, log=lambda *a: None, deadline=time.perf_counter() - 1)
    assert r["candidates"] == 1500 and "1500 of 3000" in r["truncated"]


def test_a_flip_float32_can_decide_is_not_excused():
    """The contract's teeth: a different call on a CLEARLY decided candidate -- the analysis must refuse it."""
    w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
    raw, infos = synth.synthetic_candidates(64, "ont", seed=9, start=9000)
    x = synth.to_model_input(raw)
    o32 = c_oracle.forward(w, x)
    o64 = c_oracle.forward(w, x, dtype=np.float64)
    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None), native=False)
    _, _, margins = gt_ties.near_tie_counts(o32, infos, (1e-5,))
    wins = [gt_ties.final_winner(dec, x[k], infos[k], [a[k] for a in o32])[0] for k in range(64)]
    i = int(np.argmax(margins))                                   # the most clearly decided candidate
    win = wins[i]
    j = [k for k in range(64) if wins[k] >= 0 and gt_ties.family_of(wins[k]) != gt_ties.family_of(win)][0]
    bad = [a[j].copy() for a in o32]                              # "the device" hands back ANOTHER candidate's probabilities (an indexing bug, say)
    rec = gt_ties.analyse_flip(dec, x[i], infos[i], bad, [a[i] for a in o32], [a[i] for a in o64])
    assert rec["hip_index"] != rec["oracle32_index"] == win
    assert not rec["ambiguous_at_eps_o32"] and not rec["float64_sides_with_hip"] and not gt_ties.flip_is_excused(rec)


@pytest.mark.skipif(not os.path.isfile(GOLDEN), reason="tests/golden/gt_ties.npz not minted yet (tools/gt_concordance.py --ties on an MI355X)")
def test_committed_tie_set_is_self_consistent_and_the_oracle_reproduces_it():
    """tests/golden/gt_ties.npz: every flip of the 3 x 200 000 concordance run and the tightest margins per platform, minted on an MI355X.
    On the CPU: the oracle on the committed inputs reproduces the committed float32 / float64 probabilities; decoding the committed
    probabilities reproduces the committed rows; every committed flip is excused by the committed evaluations (float32 cannot decide
    the pair, float64 decides it the HIP way); no committed candidate is decided by a margin float32 can resolve and yet flipped."""
    with np.load(GOLDEN) as z:
        meta = json.loads(str(z["meta"]))
        infos = [tuple(json.loads(str(t))) for t in z["info"]]
        x, hip, o32, o64 = z["x"], z["hip"], z["o32"], z["o64"]
    w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
    again32 = np.concatenate(c_oracle.forward(w, x), axis=1)
    again64 = np.concatenate(c_oracle.forward(w, x, dtype=np.float64), axis=1)
    assert np.abs(again32 - o32).max() <= 1e-7 and np.abs(again64 - o64).max() <= 1e-12
    assert np.abs(hip - o32).max() <= 1e-5 and np.abs(hip - o64).max() <= 4e-6
    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None))
    split = lambda row: [np.ascontiguousarray(row[a:b].reshape(1, -1)) for a, b in ((0, 21), (21, 24), (24, 57), (57, 90))]  # noqa: E731
    flips = 0
    for i, m in enumerate(meta):
        for name, probs in (("hip", hip[i]), ("oracle32", o32[i]), ("oracle64_rounded", o64[i].astype(np.float32))):
            rows = dec.decode_batch(x[i:i + 1], infos[i:i + 1], split(probs))
            assert (rows[0] if rows else None) == m[name], (i, name)
        rec = gt_ties.analyse_flip(dec, x[i], infos[i], split(hip[i]), split(o32[i]), [a[0] for a in split(o64[i])])
        differs = gt_concordance.key(m["hip"]) != gt_concordance.key(m["oracle32"])
        assert differs == (m["kind"] == "flip")
        if differs:
            flips += 1
            assert gt_ties.flip_is_excused(rec) and m["excused"], rec
            assert rec["float64_sides_with_hip"] or rec["float64_tie"]
    assert flips >= 1 and {m["platform"] for m in meta} == {"ont", "pacbio_ccs", "illumina"}
