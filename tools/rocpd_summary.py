#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace --stats) as a text table.

usage: tools/rocpd_summary.py gpurun_out/prof/bench_results.db [--skip-first N] [--phases W,K,I] [--resource-usage FILE] > profiles/rNN_....txt
Per kernel: calls, total / mean / min / max duration (us) and share of GPU kernel time; grid, LDS and the register counts.
The tracer's own register columns are NOT usable for kernels that pin operands in the accumulation half of the register file:
rocprofv3 7.2 records arch_vgpr_count as the allocation granule (e.g. 244 for a 226-VGPR kernel) and accum_vgpr_count as 0 even
when the kernel descriptor reserves 256 AGPRs.  --resource-usage takes the compiler's own -Rpass-analysis=kernel-resource-usage
output (profiles/rNN_kernel_resource_usage.txt) and prints ITS VGPR / AGPR / SGPR / scratch numbers beside the tracer's.  --skip-first drops the first N dispatches of each
kernel (warm-up launches) so the means line up with bench.py's timed region.  --phases W,K,I splits each kernel's dispatches the way bench.py issues
them -- W warm-up launches, K timed steps (batches in flight on the slots), K steps with events around every kernel, I launches alone on
one stream (`roofline.alone_kernel_ms`), K steps with HIP events around the dominant kernel only (`roofline.kernel_ms` is the dominant
kernel's mean there), one launch of the parity check -- and prints the mean of each phase.  --phases-from BENCH.json takes the split
from the bench line itself (`launch_phases`: every leg of the run in launch order, bench.py) and prints one row per leg.
"""
import sqlite3
import sys


def compiler_resources(path):
    """mangled kernel name -> (vgprs, agprs, sgprs, scratch, occupancy) from a -Rpass-analysis=kernel-resource-usage log"""
    import re
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, "
                      "sgpr_count, lds_size from kernels order by start").fetchall()
    per = {}
    for r in rows:
        per.setdefault(r[0], []).append(r)
    total = 0.0
    stats = []
    for name, rs in per.items():
        rs = rs[skip:] if len(rs) > skip else rs
        d = [(r[2] - r[1]) / 1e3 for r in rs]
        total += sum(d)
        g = rs[-1]
        stats.append((sum(d), name, len(d), sum(d) / len(d), min(d), max(d), g))
    stats.sort(reverse=True)
    print("# source: %s   (durations in microseconds; first %d dispatches per kernel skipped)" % (path, skip))
    print("%-58s %6s %12s %10s %10s %10s %6s  %s" % ("kernel", "calls", "total_us", "mean_us", "min_us", "max_us", "pct", "grid/wg vgpr+agpr(tracer) sgpr lds"))
    for tot, name, n, mean, mn, mx, g in stats:
        short = name if len(name) <= 58 else name[:55] + "..."
        print("%-58s %6d %12.1f %10.2f %10.2f %10.2f %6.2f  %dx%dx%d/%d %d+%d %d %d"
              % (short, n, tot, mean, mn, mx, 100.0 * tot / total, g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10]))
    print("# total kernel time %.1f us over %d dispatches" % (total, sum(s[2] for s in stats)))
    if "--resource-usage" in sys.argv:
        res = compiler_resources(sys.argv[sys.argv.index("--resource-usage") + 1])
        print("#\n# registers as the compiler allocated them (%s); the tracer's vgpr+agpr column above is the descriptor's granule and 0"
              % sys.argv[sys.argv.index("--resource-usage") + 1])
        for mangled, r in res.items():
            print("#   %-62s VGPRs %3d  AGPRs %3d  SGPRs %3d  scratch %d  LDS %6d  waves/SIMD %d"
                  % (mangled[:62], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("TotalSGPRs", 0), r.get("ScratchSize", 0), r.get("LDS", 0), r.get("Occupancy", 0)))
    if "--phases-from" in sys.argv:
        phases_from(per, sys.argv[sys.argv.index("--phases-from") + 1])
    if "--phases" in sys.argv:
        w, k, i = (int(v) for v in sys.argv[sys.argv.index("--phases") + 1].split(","))
        print("#\n# per phase of bench.py (mean us): %d warm-up launches | %d timed steps, batches in flight | %d steps, events on every kernel | "
              "%d launches alone on one stream | %d steps, events on the dominant kernel only (+ the parity check's launch)" % (w, k, k, i, k))
        print("%-58s %10s %10s %10s %10s %10s" % ("kernel", "warm-up", "timed", "ev.all", "alone", "ev.dominant"))
        for name, rs in per.items():
            d = [(r[2] - r[1]) / 1e3 for r in rs]
            if len(d) not in (w + 3 * k + i, w + 3 * k + i + 1):
                continue
            cut = [d[:w], d[w:w + k], d[w + k:w + 2 * k], d[w + 2 * k:w + 2 * k + i], d[w + 2 * k + i:w + 3 * k + i]]
            short = name if len(name) <= 58 else name[:55] + "..."
            print("%-58s %10.2f %10.2f %10.2f %10.2f %10.2f" % ((short,) + tuple(sum(c) / max(len(c), 1) for c in cut)))


def phases_from(per, bench_json):
    import json
    line = [l for l in open(bench_json).read().splitlines() if l.lstrip().startswith("{")][-1]
    phases = [(name, n) for name, n in (json.loads(line).get("launch_phases") or []) if n is not None]      # a leg on a handle of its own (the GT concordance) has no count
    total = sum(n for _, n in phases)
    # one column per kernel of the pass; LSTM2 is two kernels by batch size (the one-tile kernel below 2 048 candidates -- e.g. the parity check's
    # 1 024-candidate launch inside a batch-4096 run -- the two-tile kernel from there on): merged in launch order
    merged = {}
    for name, rs in per.items():
        kid = next((v for k, v in BENCH_IDS if k in name), None)
        merged.setdefault(kid if kid == "lstm2" else name, []).extend(rs)
    merged = {("lstm32_kernel<false> / lstm32_pair_kernel" if k == "lstm2" else k): sorted(rs, key=lambda r: r[1]) for k, rs in merged.items()}
    cols = [(name, [(r[2] - r[1]) / 1e3 for r in rs]) for name, rs in merged.items() if len(rs) == total]
    print("#\n# per leg of bench.py, in launch order (mean us per launch; `launch_phases` of %s, %d forward passes)" % (bench_json.split("/")[-1], total))
    if not cols:
        print("# no kernel has exactly %d dispatches: %s" % (total, {k.split("(")[0][-24:]: len(v) for k, v in per.items()}))
        return
    short = [c[0].replace("clair::", "").replace("void ", "").split("(")[0][:22] for c in cols]
    print("%-66s %6s " % ("leg", "passes") + " ".join("%22s" % s_ for s_ in short))
    at = 0
    timed = {}
    for name, n in phases:
        if n > 0:
            print("%-66s %6d " % (name[:66], n) + " ".join("%22.2f" % (sum(c[1][at:at + n]) / n) for c in cols))
            if name.startswith("timed (value)"):          # the contract's timed region: what roofline.kernel_ms_rocprof quotes
                for full, d in cols:
                    kid = next((v for k, v in BENCH_IDS if k in full), None)
                    if kid:
                        timed[kid] = round(sum(d[at:at + n]) / n / 1e3, 5)
        at += n
    if "--timed-json" in sys.argv:
        with open(sys.argv[sys.argv.index("--timed-json") + 1], "w") as f:
            json.dump(timed, f)


# substring of the traced kernel name -> bench.py's kernel id (clair_amd/_capi.py: KERNEL_NAMES)
BENCH_IDS = (("lstm32_kernel<false> / lstm32_pair_kernel", "lstm2"), ("lstm32_kernel<true>", "lstm1"), ("lstm32_kernelILb1", "lstm1"), ("gemm_split_kernel", "proj2"), ("lstm32_kernel<false>", "lstm2"), ("lstm32_kernelILb0", "lstm2"),
             ("lstm32_pair_kernel", "lstm2"), ("lstm2_fused_kernel", "lstm2"), ("l3l4_kernel", "l4"), ("tail_kernel", "tail"))


if __name__ == "__main__":
    main()
