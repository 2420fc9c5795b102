#!/usr/bin/env python3
"""Mint golden vectors from the REAL reference for the Python halves of the call_var path.

Runs only in the build container (needs /root/reference).  The reference's pure-Python code
(clair/utils.py ingest, clair/call_var.py decode / VCF writer / driver) is imported with its
missing third-party modules (tensorflow, pysam, blosc, intervaltree) replaced by MagicMock in
sys.modules; its network (TensorFlow) is NOT available, so probabilities
fed to the decode are either crafted here or come from oracle/model_np.py.

Committed outputs (data only -- inputs and expected outputs):
  tests/golden/ingest_cases.npz + ingest_*.txt.gz    G1  text records -> (X, infos) batches, stderr progress
  tests/golden/decode_cases.npz / decode_rows.json.gz   G2  (x, seq, probs) -> VCF rows, 6 output configs
  tests/golden/header_*.vcf                           G3  VCF headers without / with a .fai
  tests/golden/decode_rows.json.gz ["ensemble"]       G4  --output_for_ensemble lines (the sixth output config of G2; read back
                                                          through --input_probabilities in tests/test_decode.py)
  tests/golden/e2e_*.{txt.gz,vcf}                     C2  whole driver: tensor file -> VCF (oracle probabilities)

NumPy note: this container has NumPy 2.2; the reference pins NumPy 1.18 (README.md:127).  Under
NumPy 2 `quality_score_from` (call_var.py:568-586) and the AF division (:1151) run in float32 instead
of float64.  The goldens therefore pin clair_amd's decode in its ``numpy2`` arithmetic mode
byte-for-byte; the shipped default (``legacy``) differs only in those two formulas (float64, as
NumPy 1.x promotes) -- see clair_amd/call_var.py -- and is pinned by
  tests/golden/decode_rows_legacy.json.gz + decode_cases_legacy.npz   the reference's writer with NumPy 1.x's scalar promotion put back
                                                          (L32 / LArr below; `--legacy-decode-only` mints just these).
"""
import gc
import gzip
import io
import json
import os
import sys
import tempfile
from contextlib import redirect_stderr
from unittest import mock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, "/root/reference")
for name in ['pysam', 'blosc', 'intervaltree', 'tensorflow', 'tensorflow.python', 'tensorflow.python.util',
             'tensorflow.python.util.deprecation', 'tensorflow.python.client', 'tensorflow.python.client.device_lib',
             'tensorflow.python.ops', 'tensorflow.python.ops.array_ops', 'tensorflow.python.ops.math_ops',
             'tensorflow.python.ops.random_ops', 'tensorflow.python.framework', 'tensorflow.python.framework.ops',
             'tensorflow.python.framework.tensor_shape', 'tensorflow.python.framework.tensor_util',
             'tensorflow.contrib', 'tensorflow.contrib.layers', 'tensorflow.contrib.layers.python',
             'tensorflow.contrib.layers.python.layers', 'tensorflow.contrib.layers.python.layers.utils']:
    sys.modules[name] = mock.MagicMock(name=name)
import clair.call_var as cv  # noqa: E402  (the reference)
import clair.utils as cu  # noqa: E402
import shared.param as rparam  # noqa: E402

from clair_amd import synth, weights  # noqa: E402
from oracle import model_np  # noqa: E402

CONFIGS = {
    # name: (showRef, debug, haploid_precision, haploid_sensitive, ensemble, qual)
    "default": (False, False, False, False, False, None),
    "showref_qual": (True, False, False, False, False, 100),
    "haploid_precision": (False, False, True, False, False, None),
    "haploid_sensitive": (True, False, False, True, False, 50),
    "debug": (False, True, False, False, False, None),
    "ensemble": (False, False, False, False, True, None),
}


def close_utilities(ou):
    """The reference's close_opened_files dereferences a None fasta handle when --ref_fn is not given
    (call_var.py:299-302): swallow that, then drop the closures so the output file is flushed."""
    try:
        ou.close_opened_files()
    except AttributeError:
        pass


# ---------------------------------------------------------------------------------------------
def write_gz(path, lines):
    with gzip.open(path, "wt") as f:
        for ln in lines:
            f.write(ln + "\n")


def ingest_goldens():
    raw, infos = synth.synthetic_candidates(11, "ont", seed=31)
    infos[4][2] = infos[4][2][:16] + "N" + infos[4][2][17:]        # IUPAC 'N' centre: kept (in BASE2NUM)
    infos[7][2] = infos[7][2][:16] + "-" + infos[7][2][17:]        # not an IUPAC key: dropped (utils.py:90-91)
    infos[9][2] = infos[9][2][:16] + "R" + infos[9][2][17:]
    lines = list(synth.tensor_records(raw, infos))
    out = {}
    cases = {"a": (lines, 4), "b": (lines[:8], 4), "c": (lines[:3], 1000), "d": ([], 5), "e": ([lines[7]], 2)}
    for tag, (lns, batch) in cases.items():
        path = os.path.join(GOLD, "ingest_%s.txt.gz" % tag)
        write_gz(path, lns)
        err = io.StringIO()
        batches = []
        with redirect_stderr(err):
            for X, inf in cu.tensor_generator_from(path, batch):
                batches.append((np.array(X, copy=True), [list(i) for i in inf]))
        out["%s_batch" % tag] = np.int64(batch)
        out["%s_nbatches" % tag] = np.int64(len(batches))
        for k, (X, inf) in enumerate(batches):
            assert X.dtype == np.float32
            out["%s_X%d" % (tag, k)] = X
            out["%s_info%d" % (tag, k)] = np.array(json.dumps(inf))
        out["%s_stderr" % tag] = np.array(err.getvalue())
    np.savez_compressed(os.path.join(GOLD, "ingest_cases.npz"), **out)
    print("ingest goldens:", {k: int(out[k]) for k in out if k.endswith("_nbatches")})


# ---------------------------------------------------------------------------------------------
from golden_probs import crafted_probabilities  # noqa: E402,F401


def reference_rows(X, infos, P, cfg, pysam_all=False, ref_path=None):
    """Drive the real reference writer (file-backed, pysam mocked: every BAM look-up sees no reads);
    return per-candidate lists of output lines (positions are unique) and the whole text."""
    show_ref, debug, hp, hs, ens, qual = cfg
    oc = cv.OutputConfig(show_ref, debug, hp, hs, ens, qual)
    with tempfile.TemporaryDirectory() as td:
        out_path = os.path.join(td, "out.vcf")
        ou = cv.output_utilties_from(sample_name="SAMPLE", is_debug=debug, is_output_for_ensemble=ens,
                                     is_using_pysam_for_all_indel_bases_output=pysam_all,
                                     bam_file_path="none.bam", reference_file_path=ref_path,
                                     output_file_path=out_path)
        method = cv.batch_output_for_ensemble if ens else cv.batch_output
        Y = [P[:, 0:21], P[:, 21:24], P[:, 24:57], P[:, 57:90]]
        method((X, infos), Y, oc, ou)
        close_utilities(ou)
        del ou, method
        gc.collect()
        with open(out_path) as f:
            text = f.read()
    by_pos = {}
    for ln in text.splitlines():
        by_pos.setdefault(ln.split("\t")[1], []).append(ln)
    return [by_pos.get(infos[i][1], []) for i in range(len(infos))], text


def decode_goldens():
    rng = np.random.default_rng(2025)
    n = 720
    raw, infos = synth.synthetic_candidates(n, "ont", seed=77)
    # unique positions so rows map back to candidates
    for i, inf in enumerate(infos):
        inf[1] = str(500000 + 11 * i)
    # edge cases on the inputs
    raw[5, 16, :, :] = 0                                             # read depth 0 at the centre
    infos[6][2] = infos[6][2][:16] + "N" + infos[6][2][17:]          # centre not in ACGTU: skipped
    infos[7][2] = infos[7][2][:16] + "R" + infos[7][2][17:]
    raw[8] = 0                                                       # empty tensor
    X = synth.to_model_input(raw)
    P, tags = crafted_probabilities(rng, n)
    # a slice with network-shaped probabilities from the oracle (random-init weights, peaky heads)
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)
    po = np.concatenate(model_np.forward(w, X[600:]), axis=1)
    P[600:] = np.minimum(po, np.float32(0.99999))
    tags[600:] = ["oracle"] * (n - 600)
    rows = {}
    for name, cfg in CONFIGS.items():
        per, _ = reference_rows(X, infos, P, cfg)
        rows[name] = per
    per, _ = reference_rows(X, infos, P, CONFIGS["default"], pysam_all=True)
    rows["default_pysam_all"] = per
    np.savez_compressed(os.path.join(GOLD, "decode_cases.npz"), x=X.astype(np.int16), probs=P,
                        infos=np.array(json.dumps(infos)), tags=np.array(json.dumps(tags)))
    with gzip.open(os.path.join(GOLD, "decode_rows.json.gz"), "wt") as f:
        json.dump(rows, f, separators=(",", ":"))
    cover = {}
    for t, r in zip(tags, rows["default"]):
        cover.setdefault(t, [0, 0])
        cover[t][0] += 1
        cover[t][1] += bool(r)
    gts = {}
    for r in rows["showref_qual"]:
        for ln in r:
            gts[ln.split("\t")[-1].split(":")[0]] = gts.get(ln.split("\t")[-1].split(":")[0], 0) + 1
    print("decode goldens: rows per kind", cover, "GT histogram", gts)


# ---- the decode as the reference's pinned NumPy (1.18, README.md:127) would run it ------------------------------------------------------
# NumPy 2 (NEP 50) changed ONE thing the decode can see: a float32 SCALAR combined with a Python int / float stays float32, where NumPy 1.x
# promoted it to float64 (NEP 50's own table: `np.float32(3) + 3.` was float64; Python ints counted as int64, and int64 with float32 is
# float64).  Arrays never promoted on a Python scalar, float32 with float32 stays float32, in both.  L32 / LArr below put exactly that rule
# back under the REFERENCE's own code: every float32 scalar the reference takes out of its tensors or probabilities is an L32, whose
# arithmetic with Python numbers and float64 yields float64.  What comes out is the reference's control flow and formulas with NumPy 1.x's
# scalar promotion restated by hand -- not a run under NumPy 1.18 (not installable here), and the fixture says so.
class L32(np.float32):
    def _with(self, other, op, swapped):
        if isinstance(other, np.float32):           # float32 with float32 (L32 included): float32, in every NumPy
            r = op(np.float32(other), np.float32(self)) if swapped else op(np.float32(self), np.float32(other))
            return L32(r)
        if isinstance(other, (bool, int, float, np.float64, np.integer)):
            a, b = np.float64(float(self)), np.float64(float(other))
            return op(b, a) if swapped else op(a, b)
        return NotImplemented

    def __add__(self, o): return self._with(o, lambda a, b: a + b, False)
    def __radd__(self, o): return self._with(o, lambda a, b: a + b, True)
    def __sub__(self, o): return self._with(o, lambda a, b: a - b, False)
    def __rsub__(self, o): return self._with(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._with(o, lambda a, b: a * b, False)
    def __rmul__(self, o): return self._with(o, lambda a, b: a * b, True)
    def __truediv__(self, o): return self._with(o, lambda a, b: a / b, False)
    def __rtruediv__(self, o): return self._with(o, lambda a, b: a / b, True)
    def __neg__(self): return L32(-np.float32(self))


class LArr(np.ndarray):
    """float32 array whose scalars come out as L32 (indexing and iteration)."""
    def __getitem__(self, key):
        r = np.ndarray.__getitem__(self, key)
        if isinstance(r, np.float32) and not isinstance(r, L32):
            return L32(r)
        return r

    def __iter__(self):
        for i in range(self.shape[0]):
            yield self[i]


def legacy_decode_goldens():
    """tests/golden/decode_rows_legacy.json.gz: the cases of decode_cases.npz through the reference's writer with NumPy 1.x's scalar promotion
    (L32 / LArr above).  Sanity checks here: the emulator is inert where NumPy 2 and 1.x agree (same rows except QUAL / AF), and it does
    change what it should (some QUAL / AF values differ)."""
    with np.load(os.path.join(GOLD, "decode_cases.npz")) as z:
        X = z["x"].astype(np.float32)
        P = z["probs"]
        infos = json.loads(str(z["infos"]))
    with gzip.open(os.path.join(GOLD, "decode_rows.json.gz"), "rt") as f:
        plain = json.load(f)
    assert type(L32(np.float32(0.5)) * np.float32(0.5)) is L32 and type(1.0 - L32(0.25)) is np.float64 and type(0 + L32(3)) is np.float64
    assert type(sum(X[0, 16, :, 0].view(LArr))) is np.float64 and type(X.view(LArr)[0, 16, 0, 0]) is L32
    rows = {}
    for name in ("default", "showref_qual", "haploid_sensitive", "debug"):
        per, _ = reference_rows(X.view(LArr), infos, P.view(LArr), CONFIGS[name])
        rows[name] = per
        changed = 0
        assert len(per) == len(plain[name])
        for a, b in zip(per, plain[name]):
            assert len(a) == len(b)
            for ra, rb in zip(a, b):
                ca, cb = ra.split("\t"), rb.split("\t")
                if name != "debug":
                    assert ca[:5] == cb[:5] and ca[6:9] == cb[6:9] and ca[9].split(":")[0] == cb[9].split(":")[0] and ca[9].split(":")[2] == cb[9].split(":")[2]
                changed += ra != rb
        print("legacy decode goldens [%s]: %d rows, %d differ from the NumPy-2 rows" % (name, sum(len(a) for a in per), changed))
    # Cases in which the two arithmetics part: read depth 160 (s / 160 sits on a %.4f rounding tie for 32 values of s, and float32 / float64
    # fall on different sides of it) and calls the network is certain of (p = 1: QUAL from the 1e-300 guards; the reference under NumPy 2
    # raises there, which is why decode_cases.npz caps its probabilities at 0.99999).
    rng = np.random.default_rng(160)
    n = 192
    raw, xinfos = synth.synthetic_candidates(n, "ont", seed=160)
    for i, inf in enumerate(xinfos):
        inf[1] = str(900000 + 13 * i)
    XE = synth.to_model_input(raw)
    for i in range(n):      # centre column: depth = sum(delete + reference channels) = 160 exactly
        d = float(np.sum(XE[i, 16, :, 2] + XE[i, 16, :, 0]))
        XE[i, 16, 0, 0] += 160.0 - d
    assert all(float(np.sum(XE[i, 16, :, 2] + XE[i, 16, :, 0])) == 160.0 for i in range(n))
    PE, _ = crafted_probabilities(rng, n)
    for i in range(0, n, 12):     # every twelfth: a certain reference call / a certain homozygous SNP
        PE[i] = 0
        PE[i, [0, 4, 7, 9][(i // 12) % 4]] = 1.0      # gt21: AA, CC, GG, TT
        PE[i, 21 + (0 if "ACGT"[(i // 12) % 4] == xinfos[i][2][16] else 2)] = 1.0
        PE[i, 24 + 16] = 1.0
        PE[i, 57 + 16] = 1.0
    extra = {}
    for name in ("default", "showref_qual"):
        per, _ = reference_rows(XE.view(LArr), xinfos, PE.view(LArr), CONFIGS[name])
        extra[name] = per
    plain_rows = None
    try:
        plain_rows, _ = reference_rows(XE, xinfos, PE, CONFIGS["showref_qual"])
    except (ValueError, ZeroDivisionError) as exc:
        print("legacy decode goldens: the reference under NumPy %s raises on the certain calls (%s: %s), as expected" % (np.__version__, type(exc).__name__, exc))
    keep = [i for i in range(n) if i % 12]
    a, _ = reference_rows(XE[keep].view(LArr), [xinfos[i] for i in keep], PE[keep].view(LArr), CONFIGS["showref_qual"])
    b, _ = reference_rows(XE[keep], [xinfos[i] for i in keep], PE[keep], CONFIGS["showref_qual"])
    ndiff = sum(x != y for x, y in zip(a, b))
    print("legacy decode goldens [extra]: %d rows (showref_qual), %d of the %d candidates without a certain call give different rows under NumPy 2"
          % (sum(len(r) for r in extra["showref_qual"]), ndiff, len(keep)))
    assert ndiff >= 10
    np.savez_compressed(os.path.join(GOLD, "decode_cases_legacy.npz"), x=XE.astype(np.int16), probs=PE, infos=np.array(json.dumps(xinfos)))
    rows["extra_default"], rows["extra_showref_qual"] = extra["default"], extra["showref_qual"]
    meta = {"minted_with": "the reference's clair/call_var.py (batch_output) under NumPy %s with float32 scalars promoted as NumPy 1.x did "
                           "(tools/make_ref_goldens.py: L32 / LArr); not a run under the pinned NumPy 1.18" % np.__version__}
    with gzip.open(os.path.join(GOLD, "decode_rows_legacy.json.gz"), "wt") as f:
        json.dump({"meta": meta, "rows": rows}, f, separators=(",", ":"))


def header_goldens():
    for tag, fai in (("nofai", None), ("fai", "chr20\t64444167\t7\t60\t61\nchr21\t46709983\t65518251\t60\t61\n")):
        with tempfile.TemporaryDirectory() as td:
            ref_path = None
            if fai:
                ref_path = os.path.join(td, "ref.fa")
                open(ref_path + ".fai", "w").write(fai)
            out_path = os.path.join(td, "h.vcf")
            ou = cv.output_utilties_from("HG002", False, False, False, "none.bam", ref_path, out_path)
            ou.output_header()
            close_utilities(ou)
            del ou
            gc.collect()
            text = open(out_path).read()
        open(os.path.join(GOLD, "header_%s.vcf" % tag), "w").write(text)
        if fai:
            open(os.path.join(GOLD, "header_fai.fai"), "w").write(fai)
    print("header goldens written")


class OracleModel(object):
    """Stands in for clair.model.Clair inside the reference driver: same predict/prediction surface,
    probabilities from the float32 oracle."""

    def __init__(self, w):
        self.w = w
        self.prediction = None

    def predict(self, batchX):
        self.prediction = model_np.forward(self.w, batchX)
        return self.prediction


def e2e_goldens():
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)
    raw, infos = synth.synthetic_candidates(230, "ont", seed=123)
    path = os.path.join(GOLD, "e2e_230.txt.gz")
    write_gz(path, synth.tensor_records(raw, infos))
    saved = rparam.predictBatchSize
    rparam.predictBatchSize = 100   # 3 batches: 100 + 100 + 30
    try:
        for tag, cfg in (("default", CONFIGS["default"]), ("showref", CONFIGS["showref_qual"])):
            with tempfile.TemporaryDirectory() as td:
                out_path = os.path.join(td, "o.vcf")
                oc = cv.OutputConfig(*cfg)
                ou = cv.output_utilties_from("SAMPLE", False, False, False, "none.bam", None, out_path)
                args = mock.MagicMock()
                args.tensor_fn = path
                with redirect_stderr(io.StringIO()):
                    cv.call_variants(args, OracleModel(w), oc, ou._replace(close_opened_files=lambda: None))
                close_utilities(ou)
                del ou
                gc.collect()
                text = open(out_path).read()
            open(os.path.join(GOLD, "e2e_230_%s.vcf" % tag), "w").write(text)
            print("e2e", tag, "lines", text.count("\n"))
    finally:
        rparam.predictBatchSize = saved


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--legacy-decode-only" in sys.argv:
        legacy_decode_goldens()
        return
    ingest_goldens()
    header_goldens()
    decode_goldens()
    legacy_decode_goldens()
    e2e_goldens()


if __name__ == "__main__":
    main()
