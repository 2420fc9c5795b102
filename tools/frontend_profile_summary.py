#!/usr/bin/env python3
"""Per-kernel summary of the device front end from three rocprofv3 runs of tools/gpu/frontend_bench.py (rocpd SQLite): a kernel
trace and the FETCH_SIZE / WRITE_SIZE counter passes (each in its own run, never with the hip/hsa trace domains).

    tools/frontend_profile_summary.py <trace.db> <fetch.db> <write.db> --elements N --positions P --windows W

Counters per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): KiB per dispatch, FETCH_SIZE doubled on gfx950.  The bench runs
the front end four times (host-packed and device-parsed, twice each): launches are averaged.  "algorithmic" = bytes a launch has to move at least once.
"""
import re
import sqlite3
import sys


def arg(flag, default):
    return type(default)(sys.argv[sys.argv.index(flag) + 1]) if flag in sys.argv else default


def main():
    trace, fetch, write = (sqlite3.connect(p) for p in sys.argv[1:4])
    elements, positions, windows = arg("--elements", 0), arg("--positions", 0), arg("--windows", 0)
    dur = {}
    for name, start, end, gx in trace.execute("select name, start, end, grid_x from kernels order by start"):
        if "fe_" in name:
            dur.setdefault(name, []).append(((end - start) / 1e3, gx))

    def counter(db, which):
        out = {}
        for name, v in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (which,)):
            out.setdefault(name, []).append(float(v))
        return out
    f, w = counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
    # what one launch must move at least once: the packed slab it walks, the table rows it touches, the windows it writes
    algorithmic = {
        "fe_tally_kernel": elements * (1 + 16 / 8.0 + 4) + elements * 3 * 4,                 # SEQ byte + its share of an operation (16 B per ~8 bases) + prefix entry; ~3 counters of 4 B
        "fe_tally_tile_kernel": elements * (1 + 16 / 8.0) + positions * 64 * 2,               # SEQ byte + its share of an operation; the tile's 64 B of counters read and written once
        "fe_windows_per_base_kernel": elements * (1 + 16 / 8.0 + 4) + elements * 2 * 4,      # + two prefix look-ups
        "fe_candidate_flags_kernel": positions * (32 + 1),
        "fe_block_count_kernel": positions * 1,
        "fe_scan_write_kernel": positions * (1 + 4),
        "fe_window_flags_kernel": windows * 35 * 32,
        "fe_assemble_kernel": windows * (2112 + 33 * (32 + 8) + 1056),
        "fe_text_lines_kernel": elements * 2.03,             # every byte of the text once (SEQ + QUAL + the rest ~ 2 bytes per aligned base)
        "fe_text_emit_kernel": elements * (16 + 4) / 8.0,     # the operations written
    }
    print("%-34s %8s %10s %10s %12s %12s %12s %14s %10s" % ("kernel", "launches", "mean_us", "grid", "fetch_MB(x2)", "write_MB", "GB/s moved", "algorithmic_MB", "GB/s alg."))
    for name in sorted(dur, key=lambda k: -sum(d for d, _ in dur[k])):
        short = re.search(r"fe_\w+", name).group(0)
        d = [x for x, _ in dur[name]]
        big = max(d)
        sel = [x for x in d if x > 0.25 * big]              # the bench's full-size launches (drop the tiny warm-up front end)
        mean = sum(sel) / len(sel)
        fm = [2 * v * 1024 / 1e6 for v in f.get(name, [])]
        wm = [v * 1024 / 1e6 for v in w.get(name, [])]
        fmb = max(fm) if fm else 0.0
        wmb = max(wm) if wm else 0.0
        alg = algorithmic.get(short, 0) / 1e6
        print("%-34s %8d %10.1f %10d %12.1f %12.1f %12.0f %14.1f %10.0f"
              % (short, len(sel), mean, max(g for _, g in dur[name]), fmb, wmb, (fmb + wmb) / mean * 1e3 if mean else 0, alg, alg / mean * 1e3 if mean else 0))
    if elements:
        for key, label in (("fe_tally", "pass 1"), ("fe_windows_per_", "pass 2")):   # pass 2: per operation, or per base with --stop_consider_left_edge
            d = [x for k in dur for x, _ in dur[k] if key in k]
            runs = arg("--runs", 4)                     # the bench builds the front end four times (twice per packing path)
            per_run = sum(d) / runs
            if not d or per_run <= 0:
                print("# %s: no launches in this trace" % label)
                continue
            print("# %s: %.0f M read bases in %.0f us over %d launches = %.1f G read bases/s" % (label, elements / 1e6, per_run, len(d) // runs, elements / per_run / 1e3))
        print("# %d positions, %d windows" % (positions, windows))


if __name__ == "__main__":
    main()
