#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace --stats) as a text table.

usage: tools/rocpd_summary.py gpurun_out/prof/bench_results.db [--skip-first N] [--phases W,K,I] > profiles/rNN_....txt
Per kernel: calls, total / mean / min / max duration (us) and share of GPU kernel time;
VGPR/AGPR/LDS as recorded by the tracer.  --skip-first drops the first N dispatches of each
kernel (warm-up launches) so the means line up with bench.py's timed region.  --phases W,K,I splits each kernel's dispatches the way bench.py issues
them -- W warm-up launches, K timed steps (batches overlapped on the slots), K instrumented steps, I launches alone on one stream
-- and prints the mean of each phase: the K-step mean is `roofline.overlapped_kernel_ms`, the last-I mean is `roofline.kernel_ms`.
"""
import sqlite3
import sys


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
    print("%-58s %6s %12s %10s %10s %10s %6s  %s" % ("kernel", "calls", "total_us", "mean_us", "min_us", "max_us", "pct", "grid/wg vgpr+agpr sgpr lds"))
    for tot, name, n, mean, mn, mx, g in stats:
        short = name if len(name) <= 58 else name[:55] + "..."
        print("%-58s %6d %12.1f %10.2f %10.2f %10.2f %6.2f  %dx%dx%d/%d %d+%d %d %d"
              % (short, n, tot, mean, mn, mx, 100.0 * tot / total, g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10]))
    print("# total kernel time %.1f us over %d dispatches" % (total, sum(s[2] for s in stats)))
    if "--phases" in sys.argv:
        w, k, i = (int(v) for v in sys.argv[sys.argv.index("--phases") + 1].split(","))
        print("#\n# per phase of bench.py (mean us): %d warm-up launches | %d timed steps, batches overlapped | %d instrumented steps | %d "
              "launches alone on one stream" % (w, k, k, i))
        print("%-58s %10s %10s %10s %10s" % ("kernel", "warm-up", "timed", "instrum.", "alone"))
        for name, rs in per.items():
            d = [(r[2] - r[1]) / 1e3 for r in rs]
            if len(d) != w + 2 * k + i:
                continue
            cut = [d[:w], d[w:w + k], d[w + k:w + 2 * k], d[w + 2 * k:]]
            short = name if len(name) <= 58 else name[:55] + "..."
            print("%-58s %10.2f %10.2f %10.2f %10.2f" % ((short,) + tuple(sum(c) / max(len(c), 1) for c in cut)))


if __name__ == "__main__":
    main()
