"""Mint tests/golden/get_truth.json by running the reference dataPrepScripts/GetTruth.py (build container only; `samtools` on PATH = tests/fake_samtools.py)."""
import os, sys, subprocess, tempfile, json, stat
ROOT="/root/repo"
vcf = ("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
       "chrS\t100\t.\tA\tG\t.\t.\t.\tGT\t0/1\n"
       "chrS\t250\t.\tAC\tA,*\t.\t.\t.\tGT\t1/2\n"
       "chrT\t5\t.\tA\tG\t.\t.\t.\tGT\t1/1\n"
       "chrS\t300\t.\tG\t*,T\t.\t.\t.\tGT\t2|1\n"
       "chrS\t400\t.\tG\tT\t.\t.\t.\tGT\t0/1\n"
       "chrS\t400\t.\tG\tGA\t.\t.\t.\tGT\t0/1\n"
       "chrS\t500\t.\tT\tA,C,*\t.\t.\t.\tGT\t1/3\n"
       "chrS\t900\t.\tA\tG\t.\t.\t.\tGT:DP\t./1:7\n")
fasta = ">chrS\n" + "ACGT" * 300 + "\n"
out = {}
with tempfile.TemporaryDirectory() as tmp:
    open(os.path.join(tmp, "t.vcf"), "w").write(vcf)
    open(os.path.join(tmp, "ref.fa"), "w").write(fasta)
    sam = os.path.join(tmp, "samtools")
    open(sam, "w").write("#!/bin/sh\nexec %s %s/tests/fake_samtools.py \"$@\"\n" % (sys.executable, ROOT))
    os.chmod(sam, os.stat(sam).st_mode | stat.S_IEXEC)
    env = dict(os.environ); env["PYTHONPATH"] = "/root/reference"; env["PATH"] = tmp + os.pathsep + env["PATH"]
    for name, extra in (("all", []), ("region", ["--ctgStart", "200", "--ctgEnd", "499"])):
        r = subprocess.run([sys.executable, "-m", "dataPrepScripts.GetTruth", "--vcf_fn", os.path.join(tmp, "t.vcf"), "--ref_fn",
                            os.path.join(tmp, "ref.fa"), "--ctgName", "chrS"] + extra, capture_output=True, text=True, cwd=tmp, env=env)
        assert r.returncode == 0, r.stderr
        out[name] = {"extra": extra, "stdout": r.stdout}
        print(name, r.stdout)
json.dump({"vcf": vcf, "fasta": fasta, "cases": out}, open(ROOT + "/tests/golden/get_truth.json", "w"), indent=1)
