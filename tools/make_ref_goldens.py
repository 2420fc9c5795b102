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
NumPy 1.x promotes) -- see clair_amd/call_var.py.
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
    ingest_goldens()
    header_goldens()
    decode_goldens()
    e2e_goldens()


if __name__ == "__main__":
    main()
