#!/usr/bin/env python3
"""Where the chip's idle CU-time goes, from a rocprofv3 --kernel-trace database (rocpd SQLite) of bench.py's resident loop:
per hardware queue (= compute lane) the gap between the end of one forward-pass kernel and the start of the next, and the chip's
occupancy in workgroup slots over time (every forward-pass workgroup holds a whole CU).

usage: tools/lane_gaps.py bench_results.db [first_dispatch_fraction=0.5]     (looks at the second half of the dispatches: the steady state)"""
import sqlite3
import sys

import numpy as np

WGS = {"lstm32_kernel<true>": None, "lstm32_kernel<false>": None, "gemm_split": None, "l3l4": None, "tail_kernel": None, "lstm32_pair": None}


def main():
    db = sqlite3.connect(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 0.5
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = db.execute("select name, start, end, grid_x, workgroup_x%s from kernels order by start" % (", " + qcol if qcol else "")).fetchall()
    rows = [r for r in rows if any(k in r[0] for k in WGS)]
    if "--passes" in sys.argv:          # forward passes [a, b) in launch order (five kernels each): e.g. the timed leg behind bench.py's warm-up
        a, b = (int(v) for v in sys.argv[sys.argv.index("--passes") + 1:sys.argv.index("--passes") + 3])
        rows = rows[5 * a:5 * b]
    else:
        rows = rows[int(len(rows) * frac):int(len(rows) * 0.95)]
    if not rows:
        print("no forward-pass kernels in", sys.argv[1])
        return
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    print("# %d dispatches over %.2f ms, queue column: %s" % (len(rows), (t1 - t0) / 1e6, qcol))
    per_q = {}
    for r in rows:
        per_q.setdefault(r[5] if qcol else 0, []).append(r)
    for q, rs in sorted(per_q.items()):
        gaps = np.array([max(0, b[1] - a[2]) for a, b in zip(rs, rs[1:])]) / 1e3
        busy = sum(r[2] - r[1] for r in rs) / 1e3
        span = (rs[-1][2] - rs[0][1]) / 1e3
        by = {}
        for a, b in zip(rs, rs[1:]):
            key = "%s -> %s" % (a[0].split("(")[0].replace("void clair::", "").replace("clair::", "")[:22], b[0].split("(")[0].replace("void clair::", "").replace("clair::", "")[:22])
            by.setdefault(key, []).append(max(0, b[1] - a[2]) / 1e3)
        for key, v in sorted(by.items()):
            print("    %-50s n %4d  gap median %6.2f us  mean %6.2f" % (key, len(v), float(np.median(v)), float(np.mean(v))))
        print("queue %s: %d kernels, busy %.0f us of %.0f (%.1f %%), gap between consecutive kernels: median %.2f us, mean %.2f, p90 %.2f; gaps = %.1f %% of the lane's time"
              % (q, len(rs), busy, span, 100 * busy / span, np.median(gaps), gaps.mean(), np.percentile(gaps, 90), 100 * gaps.sum() / span))
    # occupancy: workgroups resident is unknown per instant, but an upper bound per kernel is its grid; integrate min(256, sum of grids of running kernels)
    ev = []
    for r in rows:
        wgs = r[3] // max(r[4], 1)
        ev.append((r[1], wgs)); ev.append((r[2], -wgs))
    ev.sort()
    cur, last, area, full = 0, ev[0][0], 0.0, 0.0
    hist = {}
    for t, d in ev:
        dt = t - last
        area += min(cur, 256) * dt
        full += dt
        hist[min(cur, 256) // 32] = hist.get(min(cur, 256) // 32, 0) + dt
        cur += d; last = t
    print("# sum of the grids of the kernels in flight, capped at 256 CUs: mean %.0f (%.1f %% of the chip); time share by demand: %s"
          % (area / full, 100 * area / full / 256, {"%d-%d" % (32 * k, 32 * k + 31): "%.0f%%" % (100 * v / full) for k, v in sorted(hist.items())}))


if __name__ == "__main__":
    main()
