"""Print one `callVarBam` command per reference chunk (to be run by `parallel`, a scheduler or a shell loop).

Host-side mirror of the reference's clair/callVarBamParallel.py (:19-116): contigs from `<ref>.fai` (chr1-22/X/Y and
1-22/X/Y unless --includingAllContigs), cut into --refChunkSize pieces, skipped when a --bed_fn has nothing in the chunk,
output `<prefix>.<contig>_<start>_<end>.vcf`.  This is also the multi-GPU front door: candidates are independent
(SURVEY.md 8e), so chunks are dealt round-robin over --devices GPUs (addition; one process per chunk and GPU, no
collective) and the per-chunk VCFs concatenate in order exactly as the reference's do (README: vcfcat + bcftools sort).

--run (addition) executes the chunks instead of printing them: one worker process per GPU (--devices), which keeps ONE engine up for all
of its chunks and reads the alignments of the next --readers chunks (`samtools view` into page-locked buffers, parsed and piled up on
the device: clair_amd.callVarBam.DeviceFrontEnd) while the current one goes through the network -- the process start-up and the engine
set-up are paid once per GPU instead of once per chunk, and the GPU is never waiting for a single samtools.  Same per-chunk VCFs.
"""
import os
import sys
from argparse import ArgumentParser

from .extract_variant_candidates import BedRegions, bed_regions_from

major_contigs = {"chr" + str(a) for a in list(range(1, 23)) + ["X", "Y"]}.union({str(a) for a in list(range(1, 23)) + ["X", "Y"]})


def _opt(name, value):
    return None if value is None else '--%s "%s"' % (name, value)


def _flag(name, on):
    return "--%s" % name if on else None


def _existing(path, suffix=""):
    if not isinstance(path, str) or not os.path.isfile(path + suffix):
        return None
    return os.path.abspath(path)


def _must_exist(path, suffix=""):
    p = _existing(path, suffix)
    if p is None:
        sys.exit("[ERROR] file %s not found" % (str(path) + suffix))
    return p


def chunk_overlaps_bed(regions, start, end):
    """IntervalTree.overlap(start, end) non-empty (shared/interval_tree.py:52-54)."""
    import bisect
    i = bisect.bisect_left(regions.end, start + 1)        # first interval with end > start
    return i < len(regions.start) and regions.start[i] < end


def commands(args):
    chkpnt_fn = os.path.abspath(args.chkpnt_fn) if args.chkpnt_fn else None
    if chkpnt_fn is None or not any(os.path.isfile(chkpnt_fn + s) for s in (".meta", ".npz", ".index")):
        sys.exit("[ERROR] file %s not found" % (str(args.chkpnt_fn) + ".meta"))
    bam_fn, ref_fn = _must_exist(args.bam_fn), _must_exist(args.ref_fn)
    fai_fn = _must_exist(args.ref_fn, ".fai") + ".fai"
    bed_fn, vcf_fn = _existing(args.bed_fn), _existing(args.vcf_fn)
    tree = bed_regions_from(bed_fn)
    tree = None if tree is None else {k: BedRegions(v) for k, v in tree.items()}
    head = " ".join(x for x in [
        "%s -m clair_amd.callVarBam" % (args.python or sys.executable),
        _opt("chkpnt_fn", chkpnt_fn), _opt("ref_fn", ref_fn), _opt("bam_fn", bam_fn), _opt("threshold", args.threshold),
        _opt("minCoverage", args.minCoverage), _opt("pypy", args.pypy), _opt("samtools", args.samtools), _opt("delay", args.delay),
        _opt("threads", args.tensorflowThreads), _opt("sampleName", args.sampleName), _opt("vcf_fn", vcf_fn), _opt("qual", args.qual),
        _flag("stop_consider_left_edge", args.stop_consider_left_edge), _flag("debug", args.debug),
        _flag("pysam_for_all_indel_bases", args.pysam_for_all_indel_bases), _flag("haploid_precision", args.haploid_precision),
        _flag("haploid_sensitive", args.haploid_sensitive), _flag("output_for_ensemble", args.output_for_ensemble),
        _opt("front_end", args.front_end), _opt("batch_size", args.batch_size), _opt("samtools_threads", args.samtools_threads), _opt("view_readers", args.view_readers),
        # as ONE token: a value that is a single option ("-x", "--no-PG") would otherwise be read as the next flag
        None if args.samtools_view_args is None else '--samtools_view_args="%s"' % args.samtools_view_args,
    ] if x is not None)
    out, k = [], 0
    commands.chunks = []           # (device, output file) per command, for --run
    with open(fai_fn) as fai:
        for row in fai:
            col = row.strip().split("\t")
            contig, length = col[0], int(col[1])
            if not args.includingAllContigs and contig not in major_contigs:
                continue
            end = 0
            while end < length:
                start, end = end, min(end + args.refChunkSize, length)
                in_bed = tree is not None and contig in tree and chunk_overlaps_bed(tree[contig], start, end)
                if tree is not None and not in_bed:
                    continue
                tail = [_opt("ctgName", contig), _opt("ctgStart", start), _opt("ctgEnd", end),
                        _opt("call_fn", "%s.%s_%d_%d.vcf" % (args.output_prefix, contig, start, end)),
                        _opt("bed_fn", bed_fn) if in_bed else None,
                        _opt("device", k % args.devices) if args.devices > 1 else None]
                out.append(head + " " + " ".join(x for x in tail if x is not None))
                commands.chunks.append((k % args.devices, "%s.%s_%d_%d.vcf" % (args.output_prefix, contig, start, end)))
                k += 1
    return out


def run_worker(command_lines, device, readers, all_devices=None):
    """All chunks of one GPU in this process: one engine, `readers` front ends reading ahead.  With more than one GPU in use the worker
    first pins itself to the host cores next to ITS GPU (clair_amd/shard.py: bind_worker; the reference pins its stages with taskset,
    clair/callVarBam.py:103-115): the `samtools view` pipes it reads and its page-locked text buffers then live on that socket."""
    import logging
    from . import shard
    placed = shard.bind_worker(device, all_devices or [device])
    import queue
    import shlex
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from . import call_var as cv
    from . import callVarBam
    logging.basicConfig(format="%(message)s", level=logging.INFO)
    if placed is not None:
        logging.info("[INFO] worker of device %d: PCI %s, NUMA node %s, bound to cores %s%s" % (
            device, placed["pci"], placed["numa_node"], placed["cpus_bound"], " (%s)" % placed["note"] if placed["note"] else ""))
    cv.ingest.setup_environment()
    parser = callVarBam.build_parser()
    jobs = []
    for line in command_lines:
        argv = shlex.split(line)
        jobs.append(callVarBam.normalise(parser.parse_args(argv[argv.index("clair_amd.callVarBam") + 1:] + ["--device", str(device)])))
    if not jobs:
        return 0
    m = callVarBam.load_model(jobs[0])
    try:
        buffers = queue.Queue()
        if hasattr(m, "pinned_buffer"):
            for _ in range(readers):
                buffers.put(m.pinned_buffer(callVarBam.TEXT_CHUNK + 16))
        # a front end holds its region in HBM until it has been called: chunk i is read only once chunk i - readers has been called
        turn = threading.Condition()
        state = {"called": 0, "stop": False}

        def prepare(i, args):
            with turn:
                turn.wait_for(lambda: state["stop"] or i < state["called"] + readers)
                if state["stop"]:
                    return None
            if not callVarBam.wants_device_front_end(args):
                return None
            buf = buffers.get() if hasattr(m, "pinned_buffer") else None
            try:
                fe = callVarBam.DeviceFrontEnd(args, device, pinned=(lambda n: buf) if buf is not None else None)
                fe.run()
                return fe
            finally:
                if buf is not None:
                    buffers.put(buf)

        failed = []
        with ThreadPoolExecutor(max_workers=readers) as pool:
            ahead = [pool.submit(prepare, i, a) for i, a in enumerate(jobs)]
            try:
                for args, fut in zip(jobs, ahead):
                    # One chunk's failure is that chunk's: the printed commands run a process per chunk and the others would have finished
                    # (clair/callVarBamParallel.py:90-119 under GNU parallel).  sys.exit(message) of a stage is such a failure too.
                    fe = None
                    try:
                        fe = fut.result()
                        callVarBam.call_region(args, m, prepared=fe)             # closes the front end
                    except KeyboardInterrupt:
                        raise
                    except BaseException as err:
                        if fe is not None:
                            fe.close()                                           # (idempotent) its region leaves the GPU's memory now
                        failed.append(args.call_fn)
                        logging.error("[ERROR] %s: %s" % (args.call_fn, err if str(err) else repr(err)))
                    with turn:
                        state["called"] += 1
                        turn.notify_all()
            finally:
                with turn:
                    state["stop"] = True        # interrupted: let the readers that wait for their turn run out
                    turn.notify_all()
                for fut in ahead:               # ... and give back what those that had finished hold on the GPU
                    if fut.done() and not fut.cancelled() and fut.exception() is None and fut.result() is not None:
                        fut.result().close()
    finally:
        m.close()
    if failed:
        logging.error("[ERROR] %d of %d chunk(s) failed on device %d: %s" % (len(failed), len(jobs), device, " ".join(failed)))
        return 1
    return 0


def run(args, lines):
    """--run: the command lines dealt over --devices worker processes (this module, --worker)."""
    import json
    import subprocess
    import tempfile
    per_device = {}
    for line, (device, _) in zip(lines, commands.chunks):
        per_device.setdefault(device, []).append(line)
    os.makedirs(os.path.dirname(os.path.abspath(args.output_prefix)) or ".", exist_ok=True)
    procs = []
    with tempfile.TemporaryDirectory() as tmp:
        for device, mine in sorted(per_device.items()):
            spec = os.path.join(tmp, "device_%d.json" % device)
            with open(spec, "w") as f:
                json.dump({"device": device, "readers": args.readers, "commands": mine, "devices": sorted(per_device)}, f)
            procs.append(subprocess.Popen([args.python or sys.executable, "-m", "clair_amd.callVarBamParallel", "--worker", spec]))
        codes = [p.wait() for p in procs]
    if any(codes):
        sys.exit("[ERROR] callVarBamParallel --run: worker exit codes %r" % codes)
    return 0


def build_parser():
    """Flag names and defaults of clair/callVarBamParallel.py:119-204 (help texts are this build's), plus --devices / --python."""
    parser = ArgumentParser(description="Print one callVarBam command per reference chunk")
    add = parser.add_argument
    add('--chkpnt_fn', type=str, default=None, help="model checkpoint prefix")
    add('--ref_fn', type=str, default="ref.fa", help="reference FASTA (its .fai lists the contigs)")
    add('--bed_fn', type=str, default=None, help="restrict calling to these intervals; chunks without any are skipped")
    add('--refChunkSize', type=int, default=10000000, help="chunk length in bp")
    add('--bam_fn', type=str, default="bam.bam", help="sorted alignments")
    add('--vcf_fn', type=str, default=None, help="call only at the sites of this VCF")
    add('--output_prefix', type=str, default=None, help="per-chunk VCFs are <prefix>.<contig>_<start>_<end>.vcf")
    add('--includingAllContigs', action='store_true', help="every contig of the .fai, not only 1-22, X, Y (with or without chr)")
    add('--tensorflowThreads', type=int, default=4, help="passed on as --threads (host threads per process)")
    add('--threshold', type=float, default=0.2, help="minimum allele frequency of a candidate site")
    add('--minCoverage', type=float, default=4, help="minimum depth of a candidate site")
    add('--qual', type=int, default=None, help="PASS / LowQual cut-off, optional")
    add('--sampleName', type=str, default="SAMPLE", help="sample column of the VCF")
    add('--stop_consider_left_edge', action='store_true', help="passed on to the pileup stage")
    add('--samtools', type=str, default="samtools", help="samtools executable")
    add('--pypy', type=str, default="pypy3", help="passed on; callVarBam ignores it")
    add('--delay', type=int, default=10, help="passed on; callVarBam ignores it")
    add('--debug', action='store_true', help="passed on")
    add('--pysam_for_all_indel_bases', action='store_true', help="passed on")
    add('--haploid_precision', action='store_true', help="passed on")
    add('--haploid_sensitive', action='store_true', help="passed on")
    add('--activation_only', action='store_true', help="kept for flag compatibility (plotting is a dead path)")
    add('--max_plot', type=int, default=10, help="kept for flag compatibility")
    add('--log_path', type=str, nargs='?', default=None, help="kept for flag compatibility")
    add('-p', '--parallel_level', type=int, default=2, help="kept for flag compatibility")
    add('-w', '--workers', type=int, default=8, help="kept for flag compatibility")
    add('--fast_plotting', action='store_true', help="kept for flag compatibility")
    add('--output_for_ensemble', action='store_true', help="passed on")
    # additions of this implementation
    add('--devices', type=int, default=1, help="deal the chunks round-robin over this many GPUs (--device k)")
    add('--python', type=str, default=None, help="interpreter to put in the commands, default: the running one")
    add('--run', action='store_true', help="run the chunks instead of printing the commands: one worker process per GPU, its engine kept up for all of its chunks")
    add('--readers', type=int, default=4, help="with --run: regions whose alignments are read ahead per GPU (one `samtools view` each), default: %(default)s")
    add('--worker', type=str, default=None, help="internal: a job file written by --run")
    add('--front_end', type=str, default=None, choices=("auto", "device", "host"), help="passed on (callVarBam: where the candidate search and the pileup run)")
    add('--batch_size', type=int, default=None, help="passed on (callVarBam: candidates per forward pass)")
    add('--samtools_threads', type=int, default=None, help="passed on (callVarBam: -@ of `samtools view`)")
    add('--view_readers', type=int, default=None, help="passed on (callVarBam: `samtools view` processes per region)")
    add('--samtools_view_args', type=str, default=None, help="passed on (callVarBam: extra options for `samtools view`)")
    return parser


def main(argv=None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 0:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args(argv)
    if args.worker is not None:
        import json
        spec = json.load(open(args.worker))
        sys.exit(run_worker(spec["commands"], spec["device"], spec["readers"], spec.get("devices")))
    if args.run:
        sys.exit(run(args, commands(args)))
    if not args.includingAllContigs:
        print("echo \"[INFO] --includingAllContigs not enabled, use chr{1..22,X,Y,M,MT} and {1..22,X,Y,MT} by default\"\n")
    else:
        print("echo \"[INFO] --includingAllContigs enabled\"\n")
    for line in commands(args):
        print(line)


if __name__ == "__main__":
    main()
