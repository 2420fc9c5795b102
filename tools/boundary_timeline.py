#!/usr/bin/env python3
"""Timeline digest of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/gpu/boundary_trace.py (rocpd SQLite): splits the trace at
the long idle gaps between the script's legs and prints, per leg: wall time, kernel-busy time of the chip (union of kernel intervals),
per-kernel mean duration, the gaps between consecutive kernels on each queue, and the copies (count, mean duration, busy union)."""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    db = sqlite3.connect(sys.argv[1])
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kcols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in kcols else ("stream_id" if "stream_id" in kcols else None)
    ks = db.execute("select name, start, end%s from kernels order by start" % (", " + qcol if qcol else "")).fetchall()
    cp = []
    for t in ("memory_copies", "memory_copy"):
        if t in names:
            cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
            namecol = "name" if "name" in cols else cols[0]
            sizecol = "size" if "size" in cols else None
            cp = db.execute("select %s, start, end%s from %s order by start" % (namecol, ", " + sizecol if sizecol else "", t)).fetchall()
            break
    print("# %s: %d kernel dispatches, %d copies; kernel columns %s" % (sys.argv[1], len(ks), len(cp), kcols))
    # legs: split where the chip sees no kernel for > 3 ms
    legs, cur = [], [ks[0]]
    for k in ks[1:]:
        if k[1] - max(x[2] for x in cur[-8:]) > 2e7:
            legs.append(cur)
            cur = []
        cur.append(k)
    legs.append(cur)
    for li, leg in enumerate(legs):
        if len(leg) < 200:
            continue
        t0, t1 = leg[0][1], max(k[2] for k in leg)
        fwd = sum(1 for k in leg if "tail_kernel" in k[0])
        mine = [c for c in cp if t0 - 2e5 <= c[1] <= t1]
        print("\n## leg %d: %d dispatches, %d forward passes, wall %.1f us = %.1f us per pass; chip busy with kernels %.1f %% of it; %d copies"
              % (li, len(leg), fwd, (t1 - t0) / 1e3, (t1 - t0) / 1e3 / max(fwd, 1), 100.0 * union([(k[1], k[2]) for k in leg]) / (t1 - t0), len(mine)))
        per = {}
        for k in leg:
            per.setdefault(k[0].split("(")[0][-40:], []).append((k[2] - k[1]) / 1e3)
        for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            print("   %-42s %5d x %8.1f us (min %7.1f max %7.1f)" % (n, len(d), sum(d) / len(d), min(d), max(d)))
        if qcol:
            byq = {}
            for k in leg:
                byq.setdefault(k[3], []).append(k)
            for q, kk in sorted(byq.items()):
                gaps = [(b[1] - a[2]) / 1e3 for a, b in zip(kk, kk[1:])]
                busy = union([(k[1], k[2]) for k in kk])
                pos = [g for g in gaps if g > 0]
                print("   queue %s: %d kernels, busy %.1f %% of the leg; gaps between consecutive kernels: mean %.1f us, median %.1f, > 20 us: %d, sum %.0f us"
                      % (q, len(kk), 100.0 * busy / (t1 - t0), sum(pos) / max(len(pos), 1), sorted(pos)[len(pos) // 2] if pos else 0, sum(1 for g in pos if g > 20), sum(pos)))
        if mine:
            kinds = {}
            for c in mine:
                kinds.setdefault((c[0], c[3] if len(c) > 3 else 0), []).append((c[2] - c[1]) / 1e3)
            for (n, size), d in sorted(kinds.items(), key=lambda kv: -sum(kv[1]))[:8]:
                print("   copy %-28s %9d B  %5d x %7.1f us (min %6.1f max %7.1f)%s" % (n, size, len(d), sum(d) / len(d), min(d), max(d),
                      "  = %.1f GB/s" % (size / (sum(d) / len(d)) / 1e3) if size else ""))
            print("   copies busy (union) %.1f %% of the leg" % (100.0 * union([(c[1], c[2]) for c in mine]) / (t1 - t0)))
        if "--dump" in sys.argv and mine:     # 2.5 ms from the middle of a leg with copies, every event in start order
            mid = (t0 + t1) // 2
            ev = [(k[1], k[2], "q%s %s" % (k[3] if qcol else "", k[0].split("(")[0][-28:])) for k in leg if mid <= k[1] < mid + 2500000]
            ev += [(c[1], c[2], "copy %s %d" % (c[0][12:], c[3] if len(c) > 3 else 0)) for c in mine if mid <= c[1] < mid + 2500000]
            for a, b, what in sorted(ev):
                print("      %9.1f .. %9.1f  (%6.1f us)  %s" % ((a - mid) / 1e3, (b - mid) / 1e3, (b - a) / 1e3, what))


if __name__ == "__main__":
    main()
