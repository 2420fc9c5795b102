#!/usr/bin/env python3
"""End to end on one GPU: SAM text -> candidates -> pileup windows -> network -> VCF, one process (clair_amd.callVarBam), with the
candidate search and the pileup on the device (--front_end device) and on the host (--front_end host, 1 and 4 workers).

    python tools/e2e_bam_bench.py [ref_len] [noisy_every] [depth]

Synthetic contig at 50x with 2-9 kb reads (tools/fast_reads.py; one candidate site per ~2 x noisy_every bases).  `samtools` is a shell
stand-in that prints the SAM file (`view`) and the region of the contig asked for (`faidx`), so the time measured is this pipeline's, not BAM decompression.
Random-weights model.  The VCFs of the front ends are compared byte for byte.
"""
import hashlib
import os
import stat
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fast_reads  # noqa: E402
from clair_amd import weights  # noqa: E402


def main():
    ref_len = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    noisy_every = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    tmp = tempfile.mkdtemp()
    t0 = time.time()
    case = fast_reads.make(ref_len=ref_len, depth=depth, noisy_every=noisy_every, seed=5)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    open(sam, "wb").write(case["sam"])
    fake = os.path.join(tmp, "samtools")
    # view: the whole file whatever the region (alignments beyond a sub-range touch none of its positions); faidx: the region asked for
    open(fake, "w").write("#!/bin/sh\nif [ \"$1\" = view ]; then exec cat %s; fi\nexec %s %s \"$@\"\n"
                          % (sam, sys.executable, os.path.join(ROOT, "tests", "fake_samtools.py")))
    os.chmod(fake, os.stat(fake).st_mode | stat.S_IEXEC)
    ck = weights.save_weights(os.path.join(tmp, "model"), weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1))[:-4]
    print("inputs: %.1f MB SAM, %d reads over %d bases at %dx (%.0f s to generate)" % (len(case["sam"]) / 1e6, case["n_reads"], ref_len, depth, time.time() - t0))
    digests = {}
    for front_end, batch, workers in (("device", 4096, 1), ("device", 4096, 1), ("host", 4096, 1), ("host", 4096, 4)):
        out = os.path.join(tmp, "out_%s_%d.vcf" % (front_end, workers))
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "clair_amd.callVarBam", "--chkpnt_fn", ck, "--bam_fn", sam, "--ref_fn", fa, "--ctgName",
                            case["ctg"], "--samtools", fake, "--call_fn", out, "--batch_size", str(batch), "--front_end", front_end,
                            "--front_end_workers", str(workers)], cwd=ROOT, capture_output=True, text=True)
        dt = time.time() - t0
        if r.returncode != 0:
            print(r.stderr[-2000:])
            return 1
        tensors = [l for l in r.stderr.splitlines() if l.startswith("Processed")]
        n = int(tensors[-1].split()[1]) if tensors else 0
        rows = sum(1 for l in open(out) if not l.startswith("#"))
        digests[(front_end, workers)] = hashlib.sha256(open(out, "rb").read()).hexdigest()[:16]
        print("callVarBam --front_end %s, batch %d, %d front-end worker(s): %.2f s wall for %d windows -> %d VCF rows: %.0f candidates/s, %.1f MB/s of SAM end to end"
              % (front_end, batch, workers, dt, n, rows, n / dt, len(case["sam"]) / 1e6 / dt))
        for l in r.stderr.splitlines():
            if "candidate sites" in l or "Total time" in l or "front end" in l:
                print("   ", l)
    same = len(set(digests.values())) == 1
    print("VCFs byte-identical across the front ends: %s %s" % (same, sorted(set(digests.values()))))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
