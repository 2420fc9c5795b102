#!/usr/bin/env python3
"""End to end on one GPU: SAM text -> candidates -> pileup windows -> network -> VCF, one process (clair_amd.callVarBam).

    python tools/e2e_bam_bench.py [n_reads]

Synthetic 200 kb contig at ~50x with 2-9 kb reads (tests/pileup_synth.py; 4 % substitutions, so most covered positions pass the
default 0.125 allele-frequency threshold: ~200 k candidates).  `samtools` is a shell stand-in that prints the SAM file
(`view`) and a FASTA slice (`faidx`), so the time measured is this pipeline's, not BAM decompression.  Random-weights model.
"""
import os
import stat
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pileup_synth  # noqa: E402
from clair_amd import weights  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    tmp = tempfile.mkdtemp()
    t0 = time.time()
    case = pileup_synth.synth_case(seed=5, ref_len=200000, n_reads=n_reads, read_len=(2000, 9000), cand_step=(5, 40), iupac=False,
                                   second_ctg=False)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    body = "".join(l + "\n" for l in case["sam"].splitlines() if not l.startswith("@") and not int(l.split("\t")[1]) & 2316)
    open(sam, "w").write(body)
    seq = "".join(case["fasta"].splitlines()[1:])
    open(os.path.join(tmp, "seq.txt"), "w").write(">%s\n%s\n" % (case["ctg"], seq))
    fake = os.path.join(tmp, "samtools")
    open(fake, "w").write("#!/bin/sh\nif [ \"$1\" = view ]; then exec cat %s; fi\nexec cat %s\n" % (sam, os.path.join(tmp, "seq.txt")))
    os.chmod(fake, os.stat(fake).st_mode | stat.S_IEXEC)
    ck = weights.save_weights(os.path.join(tmp, "model"), weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1))[:-4]
    print("inputs: %.1f MB SAM, %d reads (%.0f s to generate)" % (len(body) / 1e6, body.count("\n"), time.time() - t0))
    out = os.path.join(tmp, "out.vcf")
    for batch, workers in ((1024, 1), (4096, 1), (4096, 4)):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "clair_amd.callVarBam", "--chkpnt_fn", ck, "--bam_fn", sam, "--ref_fn", fa, "--ctgName",
                            case["ctg"], "--samtools", fake, "--call_fn", out, "--batch_size", str(batch), "--front_end_workers", str(workers)],
                           cwd=ROOT, capture_output=True, text=True)
        dt = time.time() - t0
        if r.returncode != 0:
            print(r.stderr[-2000:])
            return 1
        tensors = [l for l in r.stderr.splitlines() if l.startswith("Processed")]
        n = int(tensors[-1].split()[1]) if tensors else 0
        rows = sum(1 for l in open(out) if not l.startswith("#"))
        print("callVarBam, batch %d, %d front-end worker(s): %.2f s wall for %d candidate windows -> %d VCF rows: %.0f candidates/s end to end (one process, one GPU)"
              % (batch, workers, dt, n, rows, n / dt))
        for l in r.stderr.splitlines():
            if "candidate sites" in l or "Total time" in l:
                print("   ", l)
    return 0


if __name__ == "__main__":
    sys.exit(main())
