#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), as profiles/rNN_pmc_hbm_traffic.txt.

usage: tools/pmc_traffic.py <fetch results.db> <write results.db> [--skip-first N]
Counters are collected in their own runs (`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py ...`, then WRITE_SIZE),
never together with the hip/hsa trace domains.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950 correction per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled; WRITE_SIZE is
reported as read.  Algorithmic bytes: bench.py KERNEL_BYTES x batch 1024.
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter, skip):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ? order by start", (counter,)).fetchall()
    out = {}
    for name, v in rows:
        out.setdefault(name, []).append(float(v))
    return {k: sum(v[skip:]) / max(1, len(v[skip:])) for k, v in out.items() if len(v) > skip}


def main():
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 2
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE", skip)
    write = per_kernel(sys.argv[2], "WRITE_SIZE", skip)
    import bench
    alg = {"gemm_split": bench.KERNEL_BYTES["proj2"], "l3l4": bench.KERNEL_BYTES["l4"], "tail": bench.KERNEL_BYTES["tail"],
           "lstm32_kernel<false>": bench.KERNEL_BYTES["lstm2"], "lstm32_kernel<true>": bench.KERNEL_BYTES["lstm1"]}
    print("%-54s %14s %14s %14s %14s %8s" % ("kernel", "fetch_MB(x2)", "write_MB", "total_MB", "algorithmic_MB", "ratio"))
    for name in sorted(fetch):
        key = next((k for k in alg if k in name), None)
        if key is None:
            continue
        f, w = 2 * fetch[name] * 1024 / 1e6, write.get(name, 0.0) * 1024 / 1e6
        a = alg[key] * 1024 / 1e6
        print("%-54s %14.1f %14.1f %14.1f %14.1f %8.2f" % (name[:54], f, w, f + w, a, (f + w) / a))


if __name__ == "__main__":
    main()
