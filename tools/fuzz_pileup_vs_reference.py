#!/usr/bin/env python3
"""Differential test of the native pileup front end against the REAL reference scripts (build container only).

    python tools/fuzz_pileup_vs_reference.py [n_cases] [first_seed]

Per case: fresh synthetic alignments (tests/pileup_synth.py) and a random choice of switches; the reference's
dataPrepScripts/ExtractVariantCandidates.py and CreateTensor.py run unmodified as sub-processes (samtools =
tests/fake_samtools.py) and their stdout is compared byte for byte with `python -m clair_amd.extract_variant_candidates` /
`python -m clair_amd.create_tensor`; CreateTensor is fed the candidates the reference's own ExtractVariantCandidates printed.
Nothing is committed by this tool; the committed fixtures come from tools/make_pileup_goldens.py.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pileup_synth  # noqa: E402
import make_pileup_goldens as g  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    for k in range(n):
        seed = seed0 + k
        rng = np.random.default_rng(seed)
        ref_len = int(rng.integers(600, 4000))
        case = pileup_synth.synth_case(seed=seed, ref_len=ref_len, n_reads=int(rng.integers(20, 400)),
                                       read_len=(30, int(rng.integers(60, 900))), sub_rate=float(rng.choice([0.01, 0.05, 0.12])),
                                       ins_rate=float(rng.choice([0.0, 0.02, 0.06])), del_rate=float(rng.choice([0.0, 0.02, 0.06])),
                                       dup_burst=int(rng.choice([0, 0, 8])), iupac=bool(rng.random() < 0.7))
        evc_args, ct_args = [], []
        if rng.random() < 0.5:
            a = int(rng.integers(1, ref_len // 2))
            b = int(rng.integers(a, ref_len + 200))
            evc_args += ["--ctgStart", str(a), "--ctgEnd", str(b)]
            ct_args += ["--ctgStart", str(a), "--ctgEnd", str(b)]
        if rng.random() < 0.5:
            evc_args += ["--threshold", str(float(rng.choice([0.0, 0.05, 0.2, 0.5]))), "--minCoverage", str(int(rng.choice([0, 1, 4, 9])))]
        if rng.random() < 0.3:
            mq = str(int(rng.choice([1, 10, 30, 60])))
            evc_args += ["--minMQ", mq]
            ct_args += ["--minMQ", mq]
        if rng.random() < 0.3:
            ct_args += ["--stop_consider_left_edge"]
        if rng.random() < 0.3:
            ct_args += ["--dcov", str(int(rng.choice([1, 2, 5])))]
        if rng.random() < 0.3:
            ct_args += ["--minCoverage", str(int(rng.choice([1, 3, 8])))]
        bed = None
        if rng.random() < 0.3:
            cuts = sorted(int(v) for v in rng.integers(0, ref_len, 6))
            bed = "".join("%s\t%d\t%d\n" % (case["ctg"], cuts[i], cuts[i + 1]) for i in (0, 2, 4)) + "%s\t%d\t%d\n" % (case["ctg"], cuts[1], cuts[1])
        with tempfile.TemporaryDirectory() as tmp:
            fa, sam, bedf = (os.path.join(tmp, x) for x in ("ref.fa", "reads.sam", "r.bed"))
            open(fa, "w").write(case["fasta"])
            open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
            open(sam, "w").write(case["sam"])
            stub = os.path.join(tmp, "stub", "intervaltree")
            os.makedirs(stub)
            open(os.path.join(stub, "__init__.py"), "w").write(g.INTERVALTREE_STUB)
            common = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", g.FAKE]
            if bed is not None:
                open(bedf, "w").write(bed)
                evc_args += ["--bed_fn", bedf]
            want1 = g.run_reference("dataPrepScripts.ExtractVariantCandidates", common + evc_args, None, tmp, os.path.dirname(stub))
            got1 = subprocess.run([sys.executable, "-m", "clair_amd.extract_variant_candidates"] + common + evc_args, capture_output=True,
                                  text=True, cwd=ROOT)
            want2 = g.run_reference("dataPrepScripts.CreateTensor", common + ct_args, want1, tmp)
            got2 = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor"] + common + ct_args, input=want1, capture_output=True,
                                  text=True, cwd=ROOT)
        ok1, ok2 = got1.stdout == want1, got2.stdout == want2
        bad += (not ok1) + (not ok2)
        print("seed %d: candidates %5d %s | tensors %5d %s | %s %s" % (seed, want1.count("\n"), "ok" if ok1 else "MISMATCH",
                                                                      want2.count("\n"), "ok" if ok2 else "MISMATCH",
                                                                      " ".join(evc_args[:8]), " ".join(ct_args)), flush=True)
        if not ok1:
            print(got1.stderr[-500:])
        if not ok2:
            print(got2.stderr[-500:])
    print("%d cases, %d mismatches" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
