#!/usr/bin/env python3
"""Mint golden vectors for the pileup front end from the REAL reference scripts (build container only).

Runs /root/reference/dataPrepScripts/{CreateTensor,ExtractVariantCandidates}.py as sub-processes, unmodified, on the
synthetic inputs of tests/pileup_synth.py, with `--samtools "python tests/fake_samtools.py"` standing in for the samtools
binary the image lacks (the scripts only read the text `samtools view` / `samtools faidx` print).  Committed output: data only,

    tests/golden/pileup_<case>.json.gz = {"tool", "args", "fasta", "sam", "candidates", "bed", "expected"}

where `expected` is the reference's stdout (CreateTensor: tensor records; ExtractVariantCandidates: candidate rows).
`intervaltree` (bed regions, ExtractVariantCandidates only) is absent from the image: cases with --bed_fn are minted with a
minimal stand-in package on PYTHONPATH that implements the three members shared/interval_tree.py uses.
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pileup_synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FAKE = "%s %s" % (sys.executable, os.path.join(ROOT, "tests", "fake_samtools.py"))

INTERVALTREE_STUB = '''
class Interval(object):
    def __init__(self, begin, end, data=None):
        self.begin, self.end, self.data = begin, end, data
class IntervalTree(object):
    def __init__(self):
        self.items = []
    def addi(self, begin, end, data=None):
        self.items.append(Interval(begin, end, data))
    def at(self, p):
        return set(i for i in self.items if i.begin <= p < i.end)
    def overlap(self, begin, end):
        return set(i for i in self.items if i.begin < end and i.end > begin)
    def __getitem__(self, p):
        return self.at(p)
    def __len__(self):
        return len(self.items)
'''

CT_CASES = {
    # name: (synth kwargs, extra CLI args, candidates via file?, shuffle candidates?)
    "ct_default": (dict(seed=11), [], False, False),
    "ct_region_file": (dict(seed=12, n_reads=350), ["--ctgStart", "500", "--ctgEnd", "2200"], True, False),
    "ct_noleft_mq_cov": (dict(seed=13), ["--stop_consider_left_edge", "--minMQ", "20", "--minCoverage", "4"], False, False),
    "ct_dcov": (dict(seed=14, dup_burst=14), ["--dcov", "3"], False, False),
    "ct_dense_long": (dict(seed=15, n_reads=120, read_len=(300, 1200), cand_step=(1, 4), ref_len=2500), [], False, False),
    "ct_unsorted_candidates": (dict(seed=16, n_reads=200), [], False, True),
}


def run_reference(module, args, stdin_text, cwd, extra_path=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(["/root/reference"] + ([extra_path] if extra_path else []))
    r = subprocess.run([sys.executable, "-m", module] + args, input=stdin_text, capture_output=True, text=True, cwd=cwd, env=env)
    if r.returncode != 0:
        raise RuntimeError("%s failed: %s" % (module, r.stderr[-2000:]))
    return r.stdout


def mint_create_tensor():
    for name, (kw, extra, via_file, shuffle) in CT_CASES.items():
        case = pileup_synth.synth_case(**kw)
        cands = case["candidates"]
        if shuffle:
            import numpy as np
            rows = cands.splitlines()
            rng = np.random.default_rng(99)
            # local disorder: swap neighbours here and there, one far-away outlier up front
            for i in range(0, len(rows) - 1, 3):
                if rng.random() < 0.5:
                    rows[i], rows[i + 1] = rows[i + 1], rows[i]
            rows.insert(2, rows.pop(len(rows) // 2))
            cands = "\n".join(rows) + "\n"
        with tempfile.TemporaryDirectory() as tmp:
            fa, sam, can = (os.path.join(tmp, n) for n in ("ref.fa", "reads.sam", "cands.txt.gz"))
            open(fa, "w").write(case["fasta"])
            open(sam, "w").write(case["sam"])
            args = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE] + extra
            if via_file:
                with gzip.open(can, "wt") as f:
                    f.write(cands)
                args += ["--can_fn", can]
            out = run_reference("dataPrepScripts.CreateTensor", args, None if via_file else cands, tmp)
        doc = {"tool": "CreateTensor", "args": extra, "candidates_via_file": via_file, "ctg": case["ctg"],
               "fasta": case["fasta"], "sam": case["sam"], "candidates": cands, "expected": out}
        with gzip.open(os.path.join(GOLD, "pileup_%s.json.gz" % name), "wt", compresslevel=9) as f:
            json.dump(doc, f)
        print(name, "records:", out.count("\n"), "bytes:", len(out))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    mint_create_tensor()
