"""Widened rows on the MI355X: the single-process BAM -> VCF driver, the binary tensor
records and the BAM look-up path, each through the HIP forward pass, compared with the text pipeline over the same inputs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
FAKE_SAMTOOLS = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))


def _bam_case(tmp, seed=91, **kw):
    import pileup_synth
    case = pileup_synth.synth_case(seed=seed, **kw)
    fa, sam = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.sam")
    open(fa, "w").write(case["fasta"])
    open(fa + ".fai", "w").write("%s\t%d\t6\t60\t61\nchrOther\t120\t3100\t120\t121\n" % (case["ctg"], case["ref_len"]))
    open(sam, "w").write(case["sam"])
    return case, fa, sam


def _model(tmp):
    from clair_amd import weights
    w = weights.synthetic_weights(seed=4242, head_gain=6.0, lstm_bias_scale=0.1)
    return weights.save_weights(os.path.join(tmp, "model"), w)[:-4]


def _run(argv, **kw):
    r = subprocess.run([sys.executable, "-m"] + argv, cwd=ROOT, capture_output=True, text=True, **kw)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


def _rows(path):
    return [ln for ln in open(path).read().splitlines() if not ln.startswith("#")]


@pytest.mark.parametrize("front_end", ["device", "host"])
@pytest.mark.parametrize("region", [[], ["--ctgStart", "300", "--ctgEnd", "2500"]], ids=["contig", "region"])
def test_callVarBam_vcf_equals_the_three_stage_text_pipeline(tmp_path, region, front_end):
    """clair_amd.callVarBam (one process; --front_end device: candidate search and pileup on the GPU, windows never leave HBM;
    --front_end host: arrays handed between the host stages, int16 counts to the GPU) writes byte for byte the VCF of
    extract_variant_candidates | create_tensor | call_var over their text interfaces (clair/callVarBam.py:185-201)."""
    tmp = str(tmp_path)
    case, fa, sam = _bam_case(tmp)
    ck = _model(tmp)
    common = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE_SAMTOOLS] + region
    one = os.path.join(tmp, "one.vcf")
    r = _run(["clair_amd.callVarBam", "--chkpnt_fn", ck, "--call_fn", one, "--threshold", "0.15", "--minCoverage", "5", "--batch_size", "64",
              "--front_end", front_end] + common)
    assert "device front end not used" not in r.stderr and "candidate sites" in r.stderr
    r1 = _run(["clair_amd.extract_variant_candidates", "--threshold", "0.15", "--minCoverage", "5"] + common)
    tensors = os.path.join(tmp, "t.gz")
    _run(["clair_amd.create_tensor", "--tensor_fn", tensors] + common, input=r1.stdout)
    three = os.path.join(tmp, "three.vcf")
    _run(["clair_amd.call_var", "--chkpnt_fn", ck, "--tensor_fn", tensors, "--call_fn", three, "--batch_size", "64", "--ref_fn", fa])
    assert len(_rows(three)) > 20
    assert open(one).read() == open(three).read()


def test_binary_tensor_records_give_the_same_vcf_as_text_records(tmp_path):
    """create_tensor --binary -> call_var (counts go to the GPU as int16, clair_submit_counts) == the text records' VCF."""
    tmp = str(tmp_path)
    case, fa, sam = _bam_case(tmp, seed=303)
    ck = _model(tmp)
    common = ["--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE_SAMTOOLS]
    outs = {}
    for tag, extra in (("text", []), ("binary", ["--binary"])):
        tensors = os.path.join(tmp, "t_%s.gz" % tag)
        _run(["clair_amd.create_tensor", "--tensor_fn", tensors] + common + extra, input=case["candidates"])
        outs[tag] = os.path.join(tmp, "o_%s.vcf" % tag)
        _run(["clair_amd.call_var", "--chkpnt_fn", ck, "--tensor_fn", tensors, "--call_fn", outs[tag], "--batch_size", "50", "--showRef"])
    assert len(_rows(outs["text"])) > 50
    assert open(outs["text"]).read() == open(outs["binary"]).read()


@pytest.mark.parametrize("front_end", ["device", "host"])
def test_failing_upstream_stage_fails_the_run(tmp_path, front_end):
    """A `samtools view` that dies in the middle of the pileup stage must fail callVarBam (the reference checks its stages' exit
    codes, clair/callVarBam.py:218-233) instead of ending the VCF early with status 0."""
    tmp = str(tmp_path)
    first_bad = 0 if front_end == "device" else 1
    case, fa, sam = _bam_case(tmp)
    ck = _model(tmp)
    flaky = os.path.join(tmp, "flaky_samtools.py")
    marker = os.path.join(tmp, "views")
    open(flaky, "w").write(
        "import os, subprocess, sys\n"
        "if sys.argv[1] == 'view':\n"
        "    n = int(open(%r).read()) if os.path.exists(%r) else 0\n"
        "    open(%r, 'w').write(str(n + 1))\n"
        "    if n >= %d:\n"                                  # host: the first view feeds the candidate finder, the second the pileup; device: there is one
        "        out = subprocess.run([sys.executable, %r] + sys.argv[1:], capture_output=True).stdout\n"
        "        sys.stdout.buffer.write(out[:len(out) // 2]); sys.stdout.flush(); sys.exit(3)\n"
        "sys.exit(subprocess.run([sys.executable, %r] + sys.argv[1:]).returncode)\n"
        % (marker, marker, marker, first_bad, os.path.join(HERE, "fake_samtools.py"), os.path.join(HERE, "fake_samtools.py")))
    out = os.path.join(tmp, "o.vcf")
    r = subprocess.run([sys.executable, "-m", "clair_amd.callVarBam", "--chkpnt_fn", ck, "--call_fn", out, "--bam_fn", sam, "--ref_fn", fa,
                        "--ctgName", case["ctg"], "--samtools", "%s %s" % (sys.executable, flaky), "--batch_size", "64", "--front_end", front_end],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0, "callVarBam returned 0 although samtools view failed: %s" % r.stderr[-500:]
    assert "samtools view" in r.stderr


def test_call_var_with_bam_lookups_through_fake_pysam(tmp_path):
    """call_var --bam_fn / --ref_fn with a pysam in place: candidates that pass a look-up point are decoded on the BAM path, the
    rest natively; rows of candidates the BAM cannot change are identical to the run without --bam_fn."""
    tmp = str(tmp_path)
    from clair_amd import synth
    ck = _model(tmp)
    raw, infos = synth.synthetic_candidates(300, "ont", seed=123)
    tensors = os.path.join(tmp, "t.txt")
    open(tensors, "w").write("".join(ln + "\n" for ln in synth.tensor_records(raw, infos)))
    env = dict(os.environ, PYTHONPATH=HERE + os.pathsep + os.environ.get("PYTHONPATH", ""))
    shim = os.path.join(tmp, "pysam.py")
    open(shim, "w").write("from fake_pysam import *  # noqa\n")
    env["PYTHONPATH"] = tmp + os.pathsep + env["PYTHONPATH"]
    import shutil
    bam, fa = os.path.join(tmp, "reads.bam.json"), os.path.join(tmp, "ref.fa.json")
    shutil.copy(os.path.join(HERE, "golden", "pysam_bam.json"), bam)
    shutil.copy(os.path.join(HERE, "golden", "pysam_ref.json"), fa)
    open(fa + ".fai", "w").write("chr20\t60000\t7\t60\t61\n")          # the VCF header reads the contig lines from it
    a, b = os.path.join(tmp, "a.vcf"), os.path.join(tmp, "b.vcf")
    base = ["clair_amd.call_var", "--chkpnt_fn", ck, "--tensor_fn", tensors, "--batch_size", "128", "--showRef"]
    _run(base + ["--call_fn", a])
    _run(base + ["--call_fn", b, "--bam_fn", bam, "--ref_fn", fa], env=env)
    ra, rb = _rows(a), _rows(b)
    assert len(ra) == len(rb) > 100
    same = sum(x == y for x, y in zip(ra, rb))
    assert same >= len(ra) * 0.8          # the fake BAM has no reads at these positions: only look-up fall-backs may differ


def test_call_var_vcf_does_not_depend_on_the_kernel_selection(tmp_path):
    """call_var over 5 000 candidates at batch 1024 writes the same VCF bytes with layer 2 as one fused launch (opt-in since round 5),
    with the two-tile LSTM2 forced on, and from text, plain text and binary records of the same candidates."""
    import gzip
    from clair_amd import synth, tensor_binary
    tmp = str(tmp_path)
    ck = _model(tmp)
    n = 5000
    raw, infos = synth.synthetic_candidates(n, "ont", seed=31)
    text_gz, plain, binary = os.path.join(tmp, "t.txt.gz"), os.path.join(tmp, "t.txt"), os.path.join(tmp, "t.bin")
    lines = [l if l.endswith("\n") else l + "\n" for l in synth.tensor_records(raw, infos)]
    with gzip.open(text_gz, "wt", compresslevel=1) as f:
        f.writelines(lines)
    with open(plain, "w") as f:
        f.writelines(lines)
    with open(binary, "wb") as f:
        f.write(tensor_binary.MAGIC)
        f.write(tensor_binary.pack_records(infos[0][0], [int(i[1]) for i in infos], [i[2] for i in infos], raw))
    outs = {}
    for name, source, env in (("default", text_gz, {}), ("unfused", text_gz, {"CLAIR_AMD_LSTM2_FUSED": "0"}), ("fused", text_gz, {"CLAIR_AMD_LSTM2_FUSED": "1", "CLAIR_AMD_SLOTS": "2"}),
                              ("pair", text_gz, {"CLAIR_AMD_LSTM2_FUSED": "0", "CLAIR_AMD_LSTM2_PAIR": "1"}), ("plain", plain, {}), ("binary", binary, {})):
        out = os.path.join(tmp, name + ".vcf")
        _run(["clair_amd.call_var", "--chkpnt_fn", ck, "--tensor_fn", source, "--call_fn", out, "--batch_size", "1024", "--sampleName", "S", "--showRef"],
             env=dict(os.environ, **env))
        outs[name] = open(out).read()
    assert len(outs["default"].splitlines()) > 4000
    for name in ("unfused", "fused", "pair", "plain", "binary"):
        assert outs[name] == outs["default"], name


def test_callVarBam_front_end_workers_write_the_single_pass_vcf(tmp_path, monkeypatch):
    """callVarBam --front_end_workers 3 (both host stages over three sub-ranges on threads, batches consumed in position order)
    writes the VCF of the single pass, whole contig and region."""
    from clair_amd import callVarBam
    tmp = str(tmp_path)
    case, fa, sam = _bam_case(tmp, seed=97)
    ck = _model(tmp)
    monkeypatch.setattr(callVarBam, "MIN_SPAN_PER_WORKER", 300)       # the test contig is 3 kb; the CLI asks for 50 kb per sub-range
    for region in ([], ["--ctgStart", "300", "--ctgEnd", "2500"]):
        outs = []
        for workers in ("1", "3"):
            out = os.path.join(tmp, "w%s.vcf" % workers)
            callVarBam.main(["--chkpnt_fn", ck, "--call_fn", out, "--threshold", "0.15", "--minCoverage", "5", "--batch_size", "64", "--bam_fn", sam,
                             "--ref_fn", fa, "--ctgName", case["ctg"], "--samtools", FAKE_SAMTOOLS, "--front_end_workers", workers] + region)
            outs.append(open(out).read())
        assert len(outs[0].splitlines()) > 30 and outs[0] == outs[1]


def test_callVarBam_device_front_end_options_and_fall_back(tmp_path, monkeypatch, caplog):
    """--vcf_fn sites, --bed_fn and a lifted --dcov through the device front end = the host stages' VCF; an input outside the device
    formulation's regime (here: a tuple budget reported as binding) runs the host stages under `auto`, with a message, and is an
    error under `device`."""
    import gzip
    import logging
    from clair_amd import _capi, callVarBam
    tmp = str(tmp_path)
    case, fa, sam = _bam_case(tmp, seed=97, dup_burst=5)
    ck = _model(tmp)
    bed = os.path.join(tmp, "r.bed")
    open(bed, "w").write("%s\t100\t1500\n%s\t1400\t2900\nchrOther\t1\t50\n" % (case["ctg"], case["ctg"]))
    sites = os.path.join(tmp, "sites.vcf.gz")
    with gzip.open(sites, "wt") as f:
        for p in range(150, 2800, 37):
            f.write("%s\t%d\t.\tA\tC\t.\t.\t.\n" % (case["ctg"], p))
    base = ["--chkpnt_fn", ck, "--threshold", "0.15", "--minCoverage", "5", "--batch_size", "64", "--bam_fn", sam, "--ref_fn", fa, "--ctgName", case["ctg"],
            "--samtools", FAKE_SAMTOOLS]
    for extra in (["--bed_fn", bed], ["--vcf_fn", sites], ["--dcov", "2", "--ctgStart", "200", "--ctgEnd", "2700"], ["--qual", "30"],
                  ["--stop_consider_left_edge"], ["--view_readers", "3", "--samtools_threads", "2"], ["--debug"]):                          # --debug: the Python decode reads the tensors (host copies of the windows)
        outs = {}
        for fe in ("device", "host"):
            out = os.path.join(tmp, "%s.vcf" % fe)
            callVarBam.main(base + ["--call_fn", out, "--front_end", fe] + extra)
            outs[fe] = open(out).read()
        assert outs["device"] == outs["host"] and len(outs["host"].splitlines()) > 30, extra
    # the text in many small chunks (parsed on the device), then packed on the host in many small slabs
    callVarBam.main(base + ["--call_fn", os.path.join(tmp, "h0.vcf"), "--front_end", "host"])
    monkeypatch.setattr(callVarBam, "TEXT_CHUNK", 20000)
    monkeypatch.setattr(callVarBam, "SLAB_BYTES", 3000)
    for pack in ("device", "host"):
        monkeypatch.setenv("CLAIR_AMD_FE_PACK", pack)
        out = os.path.join(tmp, "slabs_%s.vcf" % pack)
        with caplog.at_level(logging.INFO):
            caplog.clear()
            callVarBam.main(base + ["--call_fn", out, "--front_end", "device"])
        assert ("text parsed on the device" in caplog.text) == (pack == "device")
        assert open(out).read() == open(os.path.join(tmp, "h0.vcf")).read()
    monkeypatch.delenv("CLAIR_AMD_FE_PACK")
    monkeypatch.setattr(_capi.Frontend, "budget_binds", lambda self, available_slots=5000000: True)
    out = os.path.join(tmp, "fallback.vcf")
    with caplog.at_level(logging.INFO):
        callVarBam.main(base + ["--call_fn", out, "--front_end", "auto"])
    assert "device front end not used (the reference's budget of 5 M outstanding tuples would run out)" in caplog.text
    callVarBam.main(base + ["--call_fn", os.path.join(tmp, "h.vcf"), "--front_end", "host"])
    assert open(out).read() == open(os.path.join(tmp, "h.vcf")).read()
    with pytest.raises(SystemExit, match="--front_end device"):
        callVarBam.main(base + ["--call_fn", out, "--front_end", "device"])


def test_callVarBamParallel_run_writes_the_vcfs_of_the_printed_commands(tmp_path):
    """callVarBamParallel --run (one worker process per GPU, one engine for all of its chunks, the next chunks' alignments read ahead)
    leaves the per-chunk VCFs that running the printed commands one by one leaves."""
    import shlex
    from clair_amd import callVarBamParallel as par
    tmp = str(tmp_path)
    case, fa, sam = _bam_case(tmp, seed=55)
    ck = _model(tmp)
    common = ["--chkpnt_fn", ck, "--bam_fn", sam, "--ref_fn", fa, "--samtools", FAKE_SAMTOOLS, "--includingAllContigs", "--refChunkSize", "700",
              "--threshold", "0.15", "--minCoverage", "5", "--batch_size", "64", "--python", sys.executable]
    lines = par.commands(par.build_parser().parse_args(common + ["--output_prefix", os.path.join(tmp, "one", "var")]))
    lines = [l for l in lines if '--ctgName "%s"' % case["ctg"] in l]
    assert len(lines) == 5
    os.makedirs(os.path.join(tmp, "one"))
    for line in lines:
        argv = shlex.split(line)
        _run(argv[argv.index("-m") + 1:])
    r = _run(["clair_amd.callVarBamParallel", "--run", "--readers", "2", "--output_prefix", os.path.join(tmp, "all", "var")] + common)
    assert r.stderr.count("device front end:") >= 5
    names = sorted(n for n in os.listdir(os.path.join(tmp, "one")) if case["ctg"] in n)
    assert len(names) == 5 and sorted(n for n in os.listdir(os.path.join(tmp, "all")) if case["ctg"] in n) == names
    rows = 0
    for n in names:
        a, b = open(os.path.join(tmp, "one", n)).read(), open(os.path.join(tmp, "all", n)).read()
        assert a == b, n
        rows += len([x for x in a.splitlines() if not x.startswith("#")])
    assert rows > 30
