#!/usr/bin/env python3
"""Per-kernel summaries of rocprofv3 PMC passes (rocpd SQLite), averaged per launch.

  tools/pmc_summary.py traffic <fetch.db> <write.db> --batch B      HBM bytes per launch (profiles/rNN_*_pmc_hbm_traffic.txt)
  tools/pmc_summary.py mfma <sq.db> --batch B --groups G            matrix-pipe utilisation (profiles/rNN_*_pmc_mfma_util.txt)

Counters are collected in their own runs (`rocprofv3 --kernel-trace --pmc ... -- python bench.py ...`), never together with the
hip/hsa trace domains.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md
(HBM section): FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled; WRITE_SIZE is reported as read.  The counters sit on the
L2's fabric side: Infinity-Cache hits are included, so "traffic" is L2 <-> fabric bytes, an upper bound on what HBM saw.
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHORT = (("lstm2_fused", "layer2_fused"), ("gemm_split", "proj2"), ("l3l4", "l4"), ("tail", "tail"), ("lstm32_pair_kernel", "lstm2"), ("lstm32_kernel<false>", "lstm2"), ("lstm32_kernel<true>", "lstm1"))


def per_kernel(path, counter, skip):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ? order by start", (counter,)).fetchall()
    out = {}
    for name, v in rows:
        out.setdefault(name, []).append(float(v))
    return {k: sum(v[skip:]) / max(1, len(v[skip:])) for k, v in out.items() if len(v) > skip}


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return None


def arg(flag, default):
    return type(default)(sys.argv[sys.argv.index(flag) + 1]) if flag in sys.argv else default


def traffic():
    import bench
    skip, batch = arg("--skip-first", 2), arg("--batch", 1024)
    fetch = per_kernel(sys.argv[2], "FETCH_SIZE", skip)
    write = per_kernel(sys.argv[3], "WRITE_SIZE", skip)
    print("%-54s %14s %14s %14s %14s %8s" % ("kernel", "fetch_MB(x2)", "write_MB", "total_MB", "design_MB", "ratio"))
    total = 0.0
    table = {}
    for name in sorted(fetch):
        s = short(name)
        if s is None:
            continue
        f, w = 2 * fetch[name] * 1024 / 1e6, write.get(name, 0.0) * 1024 / 1e6
        a = (33 * 256 * 4 + 33 * 1024 * 4 + 33 * 256 * 4 if s == "layer2_fused" else bench.DESIGN_BYTES[s]) * batch / 1e6   # fused: a1 in, zx out (its reads stay in L2), a2 out
        total += f + w
        table[s] = (f + w) * 1e6
        print("%-54s %14.1f %14.1f %14.1f %14.1f %8.2f" % (name[:54], f, w, f + w, a, (f + w) / a if a else 0))
    print("# sum over the forward pass: %.1f MB per batch of %d = %.0f B per candidate (SURVEY 8(d) algorithmic bytes: 4 584 B per candidate = %.1f MB)"
          % (total, batch, total * 1e6 / batch, 4584 * batch / 1e6))
    if "--json" in sys.argv:      # stamp the table with the sources it was measured on and merge it into profiles/pmc_traffic.json (bench.py reads it)
        import json
        import subprocess
        from clair_amd import build
        path = arg("--json", "")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        doc = json.load(open(path)) if os.path.isfile(path) else {"entries": {}}
        fused = {k: v for k, v in table.items() if k == "layer2_fused"}
        try:
            head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip() or None
        except OSError:
            head = None
        prev = doc["entries"].get(str(batch), {})
        entry = {"csrc_digest": build.csrc_digest(), "git": head, "source": arg("--source", "")}
        same = prev.get("csrc_digest") == entry["csrc_digest"]
        entry["kernels"] = dict(prev.get("kernels", {}) if same else {}, **{k: v for k, v in table.items() if k != "layer2_fused"})
        if fused or (same and prev.get("fused")):
            entry["fused"] = dict(prev.get("fused", {}) if same else {}, **fused)
        doc["entries"][str(batch)] = entry
        json.dump(doc, open(path, "w"), indent=1)


def mfma():
    skip, batch, groups = arg("--skip-first", 2), arg("--batch", 1024), arg("--groups", 8)
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"]
    c = {n: per_kernel(sys.argv[2], n, skip) for n in names}
    n_pad = (batch + 31) // 32 * 32
    wgs = {"proj2": 32 * min(groups, (33 * n_pad // 64 + 7) // 8), "l4": (n_pad + 63) // 64 * 8, "tail": n_pad // 32, "lstm1": n_pad // 16,
           "lstm2": (n_pad // 32 + 1) // 2 * 2 if n_pad >= 2048 else n_pad // 16}         # the two-tile kernel from 64 tiles on
    wgs["layer2_fused"] = 128 + 32 * ((n_pad // 64 + 7) // 8)      # 4 projection groups per XCD + the recurrent workgroups (lstm2_fused.hip.h)
    per_cu = {"proj2": 1, "l4": 1, "tail": 1, "lstm1": 1, "lstm2": 1, "layer2_fused": 1}
    print("%-46s %14s %12s %8s %12s %12s %10s %10s" % ("kernel", "mfma_busy_cyc", "duration_cyc", "SIMDs", "mfma_util", "chip_util", "wait_any", "wait_inst"))
    for name in sorted(c["GRBM_GUI_ACTIVE"]):
        s = short(name)
        if s is None:
            continue
        dur = c["GRBM_GUI_ACTIVE"][name] / 8.0
        simds = 4 * min(256, (wgs[s] + per_cu[s] - 1) // per_cu[s])
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"].get(name, 0.0)
        wave = max(c["SQ_WAVE_CYCLES"].get(name, 0.0), 1.0)
        print("%-46s %14.0f %12.0f %8d %11.1f%% %11.1f%% %9.1f%% %9.1f%%"
              % (name[:46], busy, dur, simds, 100 * busy / (dur * simds), 100 * busy / (dur * 1024),
                 100 * c["SQ_WAIT_ANY"].get(name, 0.0) / wave, 100 * c["SQ_WAIT_INST_ANY"].get(name, 0.0) / wave))
    print("# SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipes were busy, summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = duration")
    print("# in shader cycles); util = MFMA_BUSY / (duration x SIMDs the grid occupies); wait_any = share of wave cycles parked on s_waitcnt / barriers,")
    print("# wait_inst = issue stalls (matrix pipe busy, dependencies).")


def counters():
    """Per-kernel averages of whatever counters a pass collected: pmc_summary.py counters <db> [--skip-first N]"""
    db = sqlite3.connect(sys.argv[2])
    names = [r[0] for r in db.execute("select distinct counter_name from counters_collection order by 1")]
    skip = arg("--skip-first", 2)
    cols = {n: per_kernel(sys.argv[2], n, skip) for n in names}
    kernels = sorted({k for c in cols.values() for k in c})
    print("%-56s %s" % ("kernel", " ".join("%18s" % n[-18:] for n in names)))
    for k in kernels:
        if short(k) is None:
            continue
        print("%-56s %s" % (k[:56], " ".join("%18.0f" % cols[n].get(k, float("nan")) for n in names)))
        if "TCC_HIT_sum" in cols and "TCC_MISS_sum" in cols:
            h, m = cols["TCC_HIT_sum"].get(k, 0.0), cols["TCC_MISS_sum"].get(k, 0.0)
            print("%-56s L2 hit rate %.1f %% of %.0f requests" % ("", 100.0 * h / max(h + m, 1.0), h + m))


if __name__ == "__main__":
    {"traffic": traffic, "mfma": mfma, "counters": counters}[sys.argv[1]]()
