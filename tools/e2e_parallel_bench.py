#!/usr/bin/env python3
"""callVarBamParallel --run against the printed commands run one after the other, one GPU (run on the GPU box):

    python tools/e2e_parallel_bench.py [n_chunks] [chunk_len] [depth] [readers]

A synthetic contig of n_chunks x chunk_len bases (tools/fast_reads.py, 2-9 kb reads) cut into --refChunkSize chunk_len pieces.  `samtools`
is a shell stand-in: `view` prints the pre-cut SAM text of the chunk the region falls in, `faidx` the region asked for.  Measures what
keeping one engine up per GPU and reading the next chunks' alignments ahead buys over one process per chunk; the VCFs are compared.
"""
import hashlib
import os
import shlex
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
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    readers = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    tmp = tempfile.mkdtemp()
    t0 = time.time()
    case = fast_reads.make(ref_len=n_chunks * chunk, depth=depth, seed=9)
    fa = os.path.join(tmp, "ref.fa")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\n" % (case["ctg"], case["ref_len"]))
    bam = os.path.join(tmp, "reads.sam")
    open(bam, "w").close()
    # the alignments a region query of chunk k returns: those that overlap it (a read may go to two files)
    files = [open(os.path.join(tmp, "chunk_%d.sam" % k), "wb") for k in range(n_chunks)]
    total = 0
    for line in case["sam"].splitlines(keepends=True):
        col = line.split(b"\t", 10)
        pos, span = int(col[3]), len(col[9]) + 64
        for k in range(max(0, (pos - 64) // chunk), min(n_chunks - 1, (pos + span) // chunk) + 1):
            files[k].write(line)
            total += len(line)
    for f in files:
        f.close()
    fake = os.path.join(tmp, "samtools")
    open(fake, "w").write("#!/bin/sh\nif [ \"$1\" = view ]; then\n  for a in \"$@\"; do r=$a; done\n  s=${r#*:}; s=${s%%-*}\n  exec cat %s/chunk_$(( (s + 2) / %d )).sam\nfi\n"
                          "exec %s %s \"$@\"\n" % (tmp, chunk, sys.executable, os.path.join(ROOT, "tests", "fake_samtools.py")))
    os.chmod(fake, os.stat(fake).st_mode | stat.S_IEXEC)
    ck = weights.save_weights(os.path.join(tmp, "model"), weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1))[:-4]
    print("inputs: %d chunks of %d bases at %dx, %.1f MB of SAM text in all (%.0f s to generate)" % (n_chunks, chunk, depth, total / 1e6, time.time() - t0))
    common = ["--chkpnt_fn", ck, "--bam_fn", bam, "--ref_fn", fa, "--samtools", fake, "--includingAllContigs", "--refChunkSize", str(chunk),
              "--batch_size", "4096", "--python", sys.executable]
    from clair_amd import callVarBamParallel as par
    os.makedirs(os.path.join(tmp, "one"))
    lines = par.commands(par.build_parser().parse_args(common + ["--output_prefix", os.path.join(tmp, "one", "var")]))
    t0 = time.time()
    for line in lines:
        argv = shlex.split(line)
        r = subprocess.run(argv, cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            return 1
    t_one = time.time() - t0
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "clair_amd.callVarBamParallel", "--run", "--readers", str(readers), "--output_prefix", os.path.join(tmp, "all", "var")] + common,
                       cwd=ROOT, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        return 1
    t_all = time.time() - t0
    rows, same = 0, True
    for n in sorted(os.listdir(os.path.join(tmp, "one"))):
        a, b = open(os.path.join(tmp, "one", n), "rb").read(), open(os.path.join(tmp, "all", n), "rb").read()
        same = same and hashlib.sha256(a).digest() == hashlib.sha256(b).digest()
        rows += sum(1 for x in a.splitlines() if not x.startswith(b"#"))
    print("one process per chunk, one after the other: %.2f s   |   --run, one engine, %d regions read ahead: %.2f s   (%d chunks, %d VCF rows, %.0f MB/s of SAM text; "
          "VCFs byte-identical: %s)" % (t_one, readers, t_all, len(lines), rows, total / 1e6 / t_all, same))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
