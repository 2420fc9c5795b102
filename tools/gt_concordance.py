#!/usr/bin/env python3
"""VCF GT concordance at scale on the GPU box: decode the HIP probabilities and the float32 oracle's probabilities of the
same synthetic candidates with the same decoder (clair/call_var.py:733-762: arg-max over float32 products with
exact-equality membership tests) and count the rows whose CHROM/POS/REF/ALT/GT differ.  Every flip is then re-examined with
the float64 evaluation and with tools/gt_ties.py: a flip is EXCUSED only when the float32 oracle's margin between the two calls is
smaller than its own distance from float64 can move it (float32 cannot decide the pair) and float64 decides it the HIP way.  The
count of such inherently ambiguous candidates (`near_ties`, at eps = 0 / 3e-6 / 1e-5 on the probabilities) is printed beside the flips:
the reference itself (multithreaded Eigen, no fixed reduction order) would not be bit-stable on them either.

Instrumented (round 3; one of nine round-2 runs showed chunks beyond the 1e-5 tolerance on a box nobody recorded):
  * the box is identified first (GPU unique id, PCI bus, clocks, power, ECC/RAS counters, kernel selection of the handle);
  * with --taps the LSTM1 / LSTM2 outputs of every batch are kept until its chunk has been checked;
  * a chunk beyond the tolerance is dissected on the spot -- which candidates (batch, tile, lane), the HIP path re-run three times
    on the same batch (bit-compared with the first pass), the oracle re-evaluated on the offending candidates alone with ONE thread
    (float32 and float64), the first layer whose tap leaves the oracle's -- and everything needed to replay the worst 16 candidates
    is written to <dump-dir>/excursion_<platform>_<chunk>.npz, which comes back in gpurun_out/.

Usage:  python tools/gt_concordance.py [--n 200000] [--platforms ont,pacbio_ccs,illumina] [--json out.json] [--taps] [--dump-dir gpurun_out]
        [--tightest 8 --ties gpurun_out/gt_ties.npz]     (the recipe of tests/golden/gt_ties.npz)
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from clair_amd import _capi, call_var as cvar, synth, weights  # noqa: E402

PROB_TOL = 1e-5


def key(row):
    f = row.split("\t")
    return (f[0], f[1], f[3], f[4], f[-1].split(":")[0])


def _run(cmd, timeout=30):
    try:
        return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout).stdout.decode(errors="replace").strip()
    except Exception as e:   # noqa: BLE001 -- diagnostics only
        return "(%s: %s)" % (" ".join(cmd), e)


_PCI = []        # sysfs directory of the GPU this process sees (found once by box_info)


def box_info(full=True):
    """Who ran this.  A box holds eight GPUs and the container sees one: rocm-smi (which honours the container's view) names its
    unique id and PCI bus; clocks and power are then read from THAT device's sysfs directory."""
    import re
    info = {"host": os.uname().nodename, "unique_ids": [], "pci": None}
    smi = _run(["rocm-smi", "--showuniqueid", "--showbus", "--showclocks", "--showpower", "--showtemp"])
    info["unique_ids"] = re.findall(r"Unique ID:\s*(0x[0-9a-fA-F]+)", smi)
    m = re.search(r"PCI Bus:\s*([0-9a-fA-F:.]+)", smi)
    if m:
        info["pci"] = m.group(1)
        path = "/sys/bus/pci/devices/%s" % m.group(1).lower()
        if os.path.isdir(path) and not _PCI:
            _PCI.append(path)
    if not info["unique_ids"]:          # no rocm-smi: every card the kernel shows
        for p in sorted(glob.glob("/sys/class/drm/card*/device/unique_id")):
            try:
                info["unique_ids"].append("0x" + open(p).read().strip())
            except OSError:
                pass
    if full:
        info["rocm_smi"] = smi
        info["ras"] = _run(["rocm-smi", "--showrasinfo", "all"], timeout=60)
    return info


def gpu_state():
    """One line of clocks / power while the run is hot (cheap: sysfs of the device box_info found)."""
    out = []
    for card in _PCI[:1]:
        try:
            sclk = [line for line in open(os.path.join(card, "pp_dpm_sclk")).read().splitlines() if line.endswith("*")]
            out.append("sclk %s" % (sclk[0].split(":")[1].strip(" *") if sclk else "?"))
        except OSError:
            pass
        for hw in glob.glob(os.path.join(card, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(card, "hwmon/hwmon*/power1_input")):
            try:
                out.append("%.0f W" % (int(open(hw).read()) / 1e6))
                break
            except (OSError, ValueError):
                pass
    return ", ".join(out)


def dissect(eng, w, platform, c0, x, got, want, batch, taps, dump_dir, log, worst_k=16):
    """A chunk beyond the tolerance: say which side moved, where, and leave the evidence on disk."""
    from oracle import c_oracle
    err = np.max([np.abs(g - t).max(axis=1) for g, t in zip(got, want)], axis=0)       # per candidate
    bad = np.flatnonzero(err > PROB_TOL)
    order = bad[np.argsort(-err[bad])][:worst_k]
    log("  !! %s chunk at %d: %d candidates beyond %.0e (worst %.3e); positions (batch, tile, lane) of the worst: %s"
        % (platform, c0, len(bad), PROB_TOL, float(err.max()), [(int(i) // batch, int(i) % batch // 32, int(i) % 32) for i in order[:8]]))
    log("     state: %s" % gpu_state())
    rec = {"platform": platform, "chunk_start": int(c0), "index": order, "err": err[order], "x": x[order]}
    for k, (g, t) in enumerate(zip(got, want)):
        rec["hip_first_%d" % k], rec["oracle32_chunk_%d" % k] = g[order], t[order]
    # (1) the HIP side again, three times, on each offending batch
    reruns_differ = 0
    for b in sorted({int(i) // batch for i in order}):
        b0 = b * batch
        mine = [int(i) for i in order if int(i) // batch == b]
        for rep in range(3):
            again = eng.predict(x[b0:b0 + batch])
            d_first = max(float(np.abs(a[np.array(mine) - b0] - g[mine]).max()) for a, g in zip(again, got))
            whole = sum(int((a != g[b0:b0 + a.shape[0]]).any(axis=1).sum()) for a, g in zip(again, got))
            d_or = max(float(np.abs(a[np.array(mine) - b0] - t[mine]).max()) for a, t in zip(again, want))
            reruns_differ += int(whole > 0)
            log("     HIP re-run %d of batch %d: |again - first| on the offenders %.3e (%d rows of the batch differ bitwise); |again - oracle32| %.3e"
                % (rep, b, d_first, whole, d_or))
            for k, a in enumerate(again):
                rec["hip_again%d_b%d_%d" % (rep, b, k)] = a[np.array(mine) - b0]
        if taps is not None and b in taps:      # the re-run's taps against the first pass's
            a1r, a2r = eng.debug_read(0, 1, taps[b][0].shape), eng.debug_read(0, 2, taps[b][1].shape)
            log("     taps of batch %d, re-run vs first pass: a1 differs in %d values, a2 in %d" % (b, int((a1r != taps[b][0]).sum()), int((a2r != taps[b][1]).sum())))
    # (2) the oracle side again: the offenders alone, one thread, float32 and float64
    xs = x[order]
    o32_1, inter32 = c_oracle.forward(w, xs, threads=1, keep_intermediates=True)
    o64_1, inter64 = c_oracle.forward(w, xs, threads=1, keep_intermediates=True, dtype=np.float64)
    d_or = max(float(np.abs(a - t[order]).max()) for a, t in zip(o32_1, want))
    d_h32 = max(float(np.abs(a - g[order]).max()) for a, g in zip(o32_1, got))
    d_h64 = max(float(np.abs(a - g[order]).max()) for a, g in zip(o64_1, got))
    d_3264 = max(float(np.abs(a - b_).max()) for a, b_ in zip(o32_1, o64_1))
    log("     oracle on the offenders alone, 1 thread: |alone32 - chunk32| %.3e;  |hip - alone32| %.3e;  |hip - alone64| %.3e;  |alone32 - alone64| %.3e"
        % (d_or, d_h32, d_h64, d_3264))
    for k in range(4):
        rec["oracle32_alone_%d" % k], rec["oracle64_alone_%d" % k] = o32_1[k], o64_1[k]
    # (3) first diverging layer, from the taps of the FIRST pass (kept only with --taps)
    if taps is not None:
        for j, i in enumerate(order):
            b, r = int(i) // batch, int(i) % batch
            if b not in taps:
                continue
            a1, a2 = taps[b][0][:, r, :], taps[b][1][:, r, :]            # [33][256]
            rec["a1_first_%d" % j], rec["a2_first_%d" % j] = a1, a2
            if j < 8:
                log("     candidate %d (batch %d row %d): |a1 - o32| %.3e (o32 vs o64 %.3e);  |a2 - o32| %.3e (o32 vs o64 %.3e);  |p - o32| %.3e"
                    % (int(i), b, r, float(np.abs(a1 - inter32["a1"][j]).max()), float(np.abs(inter32["a1"][j] - inter64["a1"][j]).max()),
                       float(np.abs(a2 - inter32["a2"][j]).max()), float(np.abs(inter32["a2"][j] - inter64["a2"][j]).max()), float(err[i])))
    verdict = ("oracle side moved (chunk evaluation differs from the single-thread one)" if d_or > 1e-7 else
               "HIP side, not reproduced on re-run (transient)" if reruns_differ else
               "HIP side, reproduced bit for bit on re-run (deterministic: arithmetic of these inputs, not a race)")
    log("     verdict: %s" % verdict)
    rec["verdict"] = verdict
    if dump_dir:
        os.makedirs(dump_dir, exist_ok=True)
        path = os.path.join(dump_dir, "excursion_%s_%d.npz" % (platform, c0))
        np.savez_compressed(path, **rec)
        log("     evidence: %s" % path)
    return {"chunk_start": int(c0), "candidates_beyond_tol": int(len(bad)), "worst": float(err.max()), "verdict": verdict}


NEAR_TIE_EPS = (0.0, 3e-6, 1e-5)     # tools/gt_ties.py: multiplication order alone | the worst |dp| ever measured | the stated tolerance


def _pack(a4):
    return np.concatenate([np.asarray(a).reshape(-1) for a in a4])


def concordance(eng, w, platform, n, seed, batch=4096, chunk=32768, log=print, keep_taps=False, dump_dir=None, deadline=None, tightest=0):
    """Decode the HIP probabilities and the float32 oracle's of the same `n` synthetic candidates, count the differing calls, and say of
    every one whether float32 can decide it at all (tools/gt_ties.py).  `deadline` (time.perf_counter() value): no chunk STARTS after
    it -- the result then says `truncated` and how many candidates were compared.  `tightest` > 0: also keep that many candidates with
    the smallest winner / runner-up margins (inputs and all three evaluations), for tests/golden/gt_ties.npz."""
    import gt_ties
    from oracle import c_oracle
    dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None))
    flips, rows_total, worst, excursions, done = [], 0, 0.0, [], 0
    near = {eps: 0 for eps in NEAR_TIE_EPS}
    kept = []                                   # (margin, record) of the tightest candidates seen so far
    t0 = time.time()
    for c0 in range(0, n, chunk):
        if deadline is not None and c0 and time.perf_counter() > deadline:
            break
        m = min(chunk, n - c0)
        raw, infos = synth.synthetic_candidates(m, platform, seed=seed + c0, start=100000 + 7 * c0)
        x = synth.to_model_input(raw)
        got = [np.empty((m, k), np.float32) for k in (21, 3, 33, 33)]
        taps = {} if keep_taps else None
        for i in range(0, m, batch):
            nb = min(batch, m - i)
            for g, o in zip(got, eng.predict(x[i:i + batch])):
                g[i:i + o.shape[0]] = o
            if keep_taps:
                n_pad = (nb + 31) // 32 * 32
                taps[i // batch] = (eng.debug_read(0, 1, (33, n_pad, 256)), eng.debug_read(0, 2, (33, n_pad, 256)))
        want = c_oracle.forward(w, x)
        chunk_worst = max(float(np.abs(g - t).max()) for g, t in zip(got, want))
        if chunk_worst > PROB_TOL:
            excursions.append(dissect(eng, w, platform, c0, x, got, want, batch, taps, dump_dir, log))
        worst = max(worst, chunk_worst)
        counts, masks, margins = gt_ties.near_tie_counts(want, infos, NEAR_TIE_EPS)
        for eps in NEAR_TIE_EPS:
            near[eps] += counts[eps]
        rows_g = dec.decode_batch(x, infos, got)
        rows_w = dec.decode_batch(x, infos, want)
        assert len(rows_g) == len(rows_w)
        rows_total += len(rows_w)
        by_pos = None

        def record(idx, kind):
            o64 = c_oracle.forward(w, x[idx:idx + 1], dtype=np.float64)
            o64r = [o.astype(np.float32) for o in o64]
            one = lambda Y: dec.decode_batch(x[idx:idx + 1], infos[idx:idx + 1], Y)  # noqa: E731
            row_h, row_w, row_d = one([g[idx:idx + 1] for g in got]), one([t[idx:idx + 1] for t in want]), one(o64r)
            rec = gt_ties.analyse_flip(dec, x[idx], infos[idx], [g[idx] for g in got], [t[idx] for t in want], [o[0] for o in o64])
            rec.update({"kind": kind, "platform": platform, "chunk_start": int(c0), "index": int(idx), "first_margin_o32": float(margins[idx]),
                        "hip": row_h[0] if row_h else None, "oracle32": row_w[0] if row_w else None, "oracle64_rounded": row_d[0] if row_d else None,
                        "max_abs_dp": max(float(np.abs(g[idx] - t[idx]).max()) for g, t in zip(got, want)),
                        "near_tie_at": [eps for eps in NEAR_TIE_EPS if masks[eps][idx]], "excused": gt_ties.flip_is_excused(rec),
                        "arrays": {"x": x[idx].copy(), "info": list(infos[idx]), "hip": _pack([g[idx] for g in got]), "o32": _pack([t[idx] for t in want]),
                                   "o64": _pack([o[0] for o in o64])}})
            return rec

        for j, (a, b) in enumerate(zip(rows_g, rows_w)):
            if key(a) != key(b):
                if by_pos is None:
                    by_pos = {int(inf[1]): i for i, inf in enumerate(infos)}
                flips.append(record(by_pos[int(a.split("\t")[1])], "flip"))
        if tightest:
            ok = np.flatnonzero(masks[NEAR_TIE_EPS[-1]])
            flipped = {f["index"] for f in flips if f["chunk_start"] == c0}
            for idx in [i for i in ok[np.argsort(margins[ok])] if int(i) not in flipped][:tightest]:
                kept.append((float(margins[idx]), record(int(idx), "tight")))
            kept = sorted(kept, key=lambda t: t[0])[:tightest]
        done = c0 + m
        log("%s: %d / %d candidates, %d rows, %d flips, near-ties %s, max |dp| %.2e, %.0f s  [%s]" % (
            platform, done, n, rows_total, len(flips), [near[e] for e in NEAR_TIE_EPS], worst, time.time() - t0, gpu_state()))
    out = {"platform": platform, "candidates": done, "vcf_rows": rows_total, "gt_flips": len(flips), "max_abs_dp": worst, "flips": flips,
           "excursions": excursions, "near_ties": {"eps_%g" % eps: near[eps] for eps in NEAR_TIE_EPS},
           "flips_excused": sum(1 for f in flips if f["excused"]), "flips_not_excused": sum(1 for f in flips if not f["excused"]),
           "flips_among_near_ties_at_1e-5": sum(1 for f in flips if NEAR_TIE_EPS[-1] in f["near_tie_at"] or f["outcomes_tried"] != [0, 0]),
           "tightest": [r for _, r in kept]}
    if done < n:
        out["truncated"] = "time budget reached after %d of %d candidates" % (done, n)
    return out


def strip_arrays(rec):
    return {k: v for k, v in rec.items() if k != "arrays"}


def save_ties(path, results):
    """tests/golden/gt_ties.npz: every flip and the tightest margins of a run -- inputs, the three evaluations, the three calls and the analysis."""
    import json as _json
    recs = [r for res in results for r in res["flips"] + res["tightest"]]
    meta = [strip_arrays(r) for r in recs]
    np.savez_compressed(path, meta=np.array(_json.dumps(meta)),
                        x=np.stack([r["arrays"]["x"] for r in recs]).astype(np.float32) if recs else np.zeros((0, 33, 8, 4), np.float32),
                        info=np.array([_json.dumps(r["arrays"]["info"]) for r in recs]),
                        hip=np.stack([r["arrays"]["hip"] for r in recs]).astype(np.float32) if recs else np.zeros((0, 90), np.float32),
                        o32=np.stack([r["arrays"]["o32"] for r in recs]).astype(np.float32) if recs else np.zeros((0, 90), np.float32),
                        o64=np.stack([r["arrays"]["o64"] for r in recs]).astype(np.float64) if recs else np.zeros((0, 90), np.float64))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--platforms", default="ont,pacbio_ccs,illumina")
    ap.add_argument("--json", default=None)
    ap.add_argument("--head-gain", type=float, default=4.0)
    ap.add_argument("--taps", action="store_true", help="keep every batch's LSTM1 / LSTM2 outputs until its chunk is checked (first diverging layer on an excursion)")
    ap.add_argument("--dump-dir", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--tightest", type=int, default=0, help="keep this many smallest-margin candidates per platform (with --ties)")
    ap.add_argument("--ties", default=None, help="write every flip and the tightest margins as an .npz (the recipe of tests/golden/gt_ties.npz)")
    a = ap.parse_args()
    box = box_info()
    print("box: %s %s" % (box["host"], " ".join(box["unique_ids"])), flush=True)
    print(box.get("rocm_smi", ""), flush=True)
    print(box.get("ras", ""), flush=True)
    w = weights.synthetic_weights(seed=20250928, head_gain=a.head_gain)
    eng = _capi.Engine(device=0, max_batch=4096, n_slots=1)
    eng.load_weights(w)
    print("kernel selection at batch 4096, one slot: workgroups %s; env %s" % (eng.kernel_workgroups(4096),
          {k: v for k, v in os.environ.items() if k.startswith("CLAIR_AMD_")}), flush=True)
    res = [concordance(eng, w, p, a.n, 777, keep_taps=a.taps, dump_dir=a.dump_dir, log=lambda *s: print(*s, flush=True), tightest=a.tightest)
           for p in a.platforms.split(",")]
    eng.close()
    if a.ties:
        save_ties(a.ties, res)
    for r in res:
        print(json.dumps({k: v for k, v in r.items() if k not in ("flips", "tightest")}))
        for f in r["flips"]:
            print("  FLIP", json.dumps(strip_arrays(f)))
        for f in r["tightest"]:
            print("  TIGHT", json.dumps(strip_arrays(f)))
        r["flips"], r["tightest"] = [strip_arrays(f) for f in r["flips"]], [strip_arrays(f) for f in r["tightest"]]
    print("box: %s %s  excursions: %d" % (box["host"], " ".join(box["unique_ids"]), sum(len(r["excursions"]) for r in res)))
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"box": box, "results": res}, f, indent=1)
