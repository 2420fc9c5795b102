#!/usr/bin/env python3
"""Mint golden VCF rows for the BAM look-up paths of the decode from the REAL reference, with tests/fake_pysam.py installed as
`pysam` (build container only; needs /root/reference).

The reference consults the BAM for indels of 16 bases and more, for the second allele of every Ins/Ins call, and -- with
--pysam_for_all_indel_bases -- for every indel (clair/call_var.py:498-524, 540-565, 805-823).  Committed (data only):
  tests/golden/pysam_bam.json, pysam_ref.json     the fake alignment columns and reference sequence
  tests/golden/pysam_cases.npz                     x (int16 counts form), probabilities, infos
  tests/golden/pysam_rows.json.gz                  rows the reference wrote: default and --pysam_for_all_indel_bases
"""
import gc
import gzip
import json
import os
import sys
import tempfile
from unittest import mock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")

import fake_pysam  # noqa: E402

sys.path.insert(0, "/root/reference")
sys.modules["pysam"] = fake_pysam
for name in ['blosc', 'intervaltree', 'tensorflow', 'tensorflow.python', 'tensorflow.python.util',
             'tensorflow.python.util.deprecation', 'tensorflow.python.client', 'tensorflow.python.client.device_lib',
             'tensorflow.python.ops', 'tensorflow.python.ops.array_ops', 'tensorflow.python.ops.math_ops',
             'tensorflow.python.ops.random_ops', 'tensorflow.python.framework', 'tensorflow.python.framework.ops',
             'tensorflow.python.framework.tensor_shape', 'tensorflow.python.framework.tensor_util',
             'tensorflow.contrib', 'tensorflow.contrib.layers', 'tensorflow.contrib.layers.python',
             'tensorflow.contrib.layers.python.layers', 'tensorflow.contrib.layers.python.layers.utils']:
    sys.modules[name] = mock.MagicMock(name=name)
import clair.call_var as cv  # noqa: E402  (the reference)

from clair_amd import synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from golden_probs import crafted_probabilities  # noqa: E402


def main():
    rng = np.random.default_rng(909)
    n = 360
    raw, infos = synth.synthetic_candidates(n, "ont", seed=515)
    contig = "chr20"
    ref_len = 60000
    ref_seq = "".join("ACGT"[b] for b in rng.integers(0, 4, size=ref_len))
    for i, inf in enumerate(infos):
        inf[0] = contig
        inf[1] = str(1000 + 41 * i)
        p = int(inf[1])
        inf[2] = ref_seq[p - 17:p + 16]                     # seq[16] is the base AT the 1-based position
    X = synth.to_model_input(raw)
    P, tags = crafted_probabilities(rng, n, long_bias=True)
    # alignment columns: for most candidates a handful of reads with insertions / deletions after the position
    cols = {}
    for i, inf in enumerate(infos):
        p = int(inf[1])
        kind = i % 6
        if kind == 5:
            continue                                        # no reads at all: the look-up answers ""
        toks = []
        base = ref_seq[p - 1]
        for _ in range(int(rng.integers(3, 12))):
            r = rng.random()
            if r < 0.45:
                ln = int(rng.choice([1, 2, 3, 5, 9, 15, 16, 17, 23, 40, 51]))
                ins = "".join("ACGTacgt"[b] for b in rng.integers(0, 8, size=ln))
                toks.append("%s+%d%s" % (base, ln, ins))
            elif r < 0.8:
                ln = int(rng.choice([1, 2, 4, 8, 15, 16, 18, 30, 50, 55]))
                toks.append("%s-%d%s" % (base, ln, "N" * ln))
            elif r < 0.9:
                toks.append("*")
            else:
                toks.append(base if rng.random() < 0.5 else base.lower())
        if kind == 4 and toks:                              # repeated sequences: a clear most-frequent one
            toks = toks + [t for t in toks if "+" in t][:1] * 3 + [t for t in toks if "-" in t][:1] * 3
        cols[str(p - 1)] = toks
        cols[str(p)] = ["%s+3AAA" % ref_seq[p]] * 2        # a neighbouring column that must be ignored
    bam = {contig: cols}
    json.dump(bam, open(os.path.join(GOLD, "pysam_bam.json"), "w"), separators=(",", ":"))
    json.dump({contig: ref_seq}, open(os.path.join(GOLD, "pysam_ref.json"), "w"))
    rows = {}
    for name, pysam_all in (("default", False), ("pysam_all", True)):
        oc = cv.OutputConfig(False, False, False, False, False, None)
        with tempfile.TemporaryDirectory() as td:
            out_path = os.path.join(td, "out.vcf")
            ou = cv.output_utilties_from(sample_name="SAMPLE", is_debug=False, is_output_for_ensemble=False,
                                         is_using_pysam_for_all_indel_bases_output=pysam_all,
                                         bam_file_path=os.path.join(GOLD, "pysam_bam.json"),
                                         reference_file_path=os.path.join(GOLD, "pysam_ref.json"), output_file_path=out_path)
            Y = [P[:, 0:21], P[:, 21:24], P[:, 24:57], P[:, 57:90]]
            cv.batch_output((X, infos), Y, oc, ou)
            ou.close_opened_files()
            del ou
            gc.collect()
            text = open(out_path).read()
        by_pos = {}
        for ln in text.splitlines():
            by_pos.setdefault(ln.split("\t")[1], []).append(ln)
        rows[name] = [by_pos.get(infos[i][1], []) for i in range(n)]
    np.savez_compressed(os.path.join(GOLD, "pysam_cases.npz"), x=X.astype(np.int16), probs=P,
                        infos=np.array(json.dumps(infos)), tags=np.array(json.dumps(tags)))
    with gzip.open(os.path.join(GOLD, "pysam_rows.json.gz"), "wt") as f:
        json.dump(rows, f, separators=(",", ":"))
    long_rows = sum(1 for r in rows["default"] for ln in r if max(len(a) for a in (ln.split("\t")[3] + "," + ln.split("\t")[4]).split(",")) > 16)
    diff = sum(a != b for a, b in zip(rows["default"], rows["pysam_all"]))
    print("pysam goldens: %d candidates, %d rows, %d rows with an allele longer than 16, %d candidates differ under --pysam_for_all_indel_bases"
          % (n, sum(len(r) for r in rows["default"]), long_rows, diff))


if __name__ == "__main__":
    main()
