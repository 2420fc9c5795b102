#!/usr/bin/env python3
"""Host-side front end rates: candidate extraction and pileup tensors, native vs the reference scripts.

    python tools/bam_pipeline_bench.py [--ref]      # --ref: also time /root/reference's scripts (build container only)

Synthetic 200 kb contig, 2000 reads of 2-9 kb (~50x), candidates every ~22 bp (tests/pileup_synth.py); `samtools` is
tests/fake_samtools.py over text files.  Single thread each.  Results: profiles/r01_host_pipeline.txt.
"""
import io
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pileup_synth  # noqa: E402
from clair_amd import _hostapi, create_tensor as ct  # noqa: E402

FAKE = "%s %s" % (sys.executable, os.path.join(ROOT, "tests", "fake_samtools.py"))


def main():
    t0 = time.time()
    case = pileup_synth.synth_case(seed=5, ref_len=200000, n_reads=2000, read_len=(2000, 9000), cand_step=(5, 40), iupac=False,
                                   second_ctg=False)
    print("synthetic case: %.1f MB SAM, %d listed candidates (%.0f s to generate)"
          % (len(case["sam"]) / 1e6, case["candidates"].count("\n"), time.time() - t0))
    tmp = tempfile.mkdtemp()
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"])
    common = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE]
    full = "".join(case["fasta"].splitlines()[1:]).upper()
    samb = "".join(l + "\n" for l in case["sam"].splitlines() if not l.startswith("@") and not int(l.split("\t")[1]) & 2316).encode()

    # in-process native cores
    t0 = time.time()
    f = _hostapi.CandidateFinder(case["ctg"], full, 0, min_coverage=4, threshold=0.125)
    f.feed(samb)
    f.finish()
    pos = f.take_positions()
    t_evc = time.time() - t0
    print("native candidate finder : %.3f s  %6.1f MB SAM/s  -> %d candidates" % (t_evc, len(samb) / t_evc / 1e6, len(pos)))
    cands = ct.candidate_positions_from(io.StringIO(case["candidates"]), None, None)
    t0 = time.time()
    b = _hostapi.PileupBuilder(case["ctg"], full, 0, cands)
    b.feed(samb)
    b.finish()
    n = b.pending()
    t_ct = time.time() - t0
    print("native pileup builder   : %.3f s  %6.1f MB SAM/s  -> %d windows (%.0f windows/s)" % (t_ct, len(samb) / t_ct / 1e6, n, n / t_ct))
    t0 = time.time()
    centres, seqs, counts = b.take_arrays()
    print("   array hand-off       : %.3f s (%d x 1056 int32)" % (time.time() - t0, len(centres)))
    b2 = _hostapi.PileupBuilder(case["ctg"], full, 0, cands)
    b2.feed(samb)
    b2.finish()
    t0 = time.time()
    text = b2.take_text(1 << 28)
    print("   text records instead : %.3f s (%.1f MB, what the reference pipes to call_var)" % (time.time() - t0, len(text) / 1e6))

    # command lines (interpreter start-up + fake samtools included)
    t0 = time.time()
    r1 = subprocess.run([sys.executable, "-m", "clair_amd.extract_variant_candidates"] + common, capture_output=True, text=True, cwd=ROOT)
    t_cli_evc = time.time() - t0
    t0 = time.time()
    r2 = subprocess.run([sys.executable, "-m", "clair_amd.create_tensor"] + common, input=case["candidates"], capture_output=True, text=True, cwd=ROOT)
    t_cli_ct = time.time() - t0
    print("CLI extract_variant_candidates: %.2f s;  CLI create_tensor: %.2f s" % (t_cli_evc, t_cli_ct))
    if "--ref" in sys.argv:
        env = dict(os.environ)
        stub = os.path.join(tmp, "stub", "intervaltree")
        os.makedirs(stub)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_pileup_goldens as g
        open(os.path.join(stub, "__init__.py"), "w").write(g.INTERVALTREE_STUB)
        env["PYTHONPATH"] = "/root/reference" + os.pathsep + os.path.dirname(stub)
        t0 = time.time()
        q1 = subprocess.run([sys.executable, "-m", "dataPrepScripts.ExtractVariantCandidates"] + common, capture_output=True, text=True, cwd=tmp, env=env)
        t_ref_evc = time.time() - t0
        t0 = time.time()
        q2 = subprocess.run([sys.executable, "-m", "dataPrepScripts.CreateTensor"] + common, input=case["candidates"], capture_output=True, text=True, cwd=tmp, env=env)
        t_ref_ct = time.time() - t0
        print("reference ExtractVariantCandidates.py (CPython %d.%d): %.1f s, output identical: %s" % (sys.version_info[0], sys.version_info[1], t_ref_evc, q1.stdout == r1.stdout))
        print("reference CreateTensor.py             (CPython %d.%d): %.1f s, output identical: %s" % (sys.version_info[0], sys.version_info[1], t_ref_ct, q2.stdout == r2.stdout))
        print("speed-up of the native cores over the reference scripts: candidates %.0fx, pileup %.0fx" % (t_ref_evc / t_evc, t_ref_ct / t_ct))


if __name__ == "__main__":
    main()
