#!/usr/bin/env python3
"""Headline benchmark: candidate sites/sec of the call_var forward pass on N MI355X of one node.

A "step" is one pass of the hot path (BiLSTM x2 -> slice dense -> dense tail -> softmax heads) over one
batch of synthetic pileup tensors already resident in HBM.  Default workload is BASELINE.json
configs[1]: ONT-profile candidates, batch 1024, one GPU (`--platform pacbio_ccs --batch 4096` and
`--platform illumina --batch 8192` are configs[2] and [4]).

Ranks.  One process per GPU.  Under `python -m torch.distributed.run ... bench.py --gpus N` the launcher's
RANK / LOCAL_RANK / WORLD_SIZE are used; a plain `python bench.py --gpus N` spawns the N ranks itself.
Candidate sites shard across ranks with no data-path collective (clair_amd/shard.py); the ranks exchange, over
RCCL / xGMI through the C ABI's communicator (include/clair_amd.h: clair_comm_*), the weight blob from rank 0
(once), the barrier around the timed region and the per-rank timers.  `--scaling weak` (default): every rank
runs --steps batches.  `--scaling strong --candidates M`: a fixed set of M candidates (default 5 000 000, the
whole-genome configs[3]) is dealt in contiguous blocks of whole batches; --steps is then derived.

After the contract's timed region (`value`: inputs resident in HBM) the same K steps are timed twice more, each exactly like `value`
(W warm-up steps, barrier + device sync on both sides, max over ranks):
  value_boundary        -- through the reference's own boundary, Clair.predict(batchX) on HOST arrays (clair/model.py:946-966, driven by
                           clair/call_var.py:1331-1352): pageable float32 NumPy batches in, four fresh NumPy arrays per batch out
                           (clair_submit / clair_wait, `boundary.slots` batches in flight);
  value_boundary_int16  -- the same with the raw int16 counts the pileup stage produces (clair_submit_counts);
and one leg over the whole candidate set BASELINE.json's config names (`--full-candidates`, 200 704 = 196 x 1024 for configs[1]):
  value_full_config / value_boundary_full_config -- so that a driver run with a handful of --steps still shows the sustained rate.
  value_sustained       -- the resident loop again for at least `--sustained-seconds` (default 2.5 s = ~20 000 steps at batch 1024), timed
                           exactly like `value`: the board is at its power cap for all but the first milliseconds of it, which is the
                           regime a chr20 / whole-genome candidate set runs in (the reference's own figure of merit is whole-run wall
                           clock, clair/call_var.py:1317, 1365).  `config.rates` says which of the three resident rates is which.
`gpu_state` holds the shader clock and socket power sampled from sysfs (tools/gpu_state_sampler.py, a process of its own, every
`gpu_state.period_ms` milliseconds) during each of those legs, on every rank (`per_rank[r].gpu_state`; the top-level one is rank 0's).
`gt_concordance_200k` (N=1 only, like cpu_baseline): tools/gt_concordance.py's count of VCF rows whose CHROM/POS/REF/ALT/GT differ between
the decode of the HIP probabilities and the decode of the float32 oracle's, over `--gt-candidates` (200 000) candidates of each platform
profile; every differing row analysed by tools/gt_ties.py -- `flips_not_excused` (must be 0: a flip is excused only when float32 cannot decide
the pair and float64 decides it the HIP way) and `near_ties`, the rows that are inherently ambiguous at eps = 0 / 3e-6 / 1e-5 on the
probabilities.  The leg has a time budget enforced inside each platform's loop (`--gt-seconds`; a platform cut short says `truncated`).
`--scaling strong --shard-of R/W` (one GPU): exactly the block rank R of a W-rank job over `--candidates` would run; per_rank[0] reads as that
rank's entry of the W-rank line (configs[3] and configs[4] on a one-GPU box).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- the dominant kernel = the one with the most chip time (stand-alone duration x share of the 256 CUs its grid
                  occupies) in THIS run's own per-kernel table: SURVEY.md 8(d) algorithmic FLOP per launch / its mean
                  HIP-event duration IN the multi-stream configuration of the timed loop / the 2.5 PFLOP/s dense f16 MFMA
                  peak (ONE number: `frac` == `kernels[dominant].frac`; `frac_rocprof` / `kernel_ms_rocprof` are the same from the committed
                  rocprofv3 trace of this configuration -- no marker packets in the queues -- when it was taken on these kernel sources).
                  `executed_frac` counts the three fp16 MFMAs the 2-way split issues per product; `alone_*` is the
                  same kernel with nothing else on the chip; `kernels` carries the same fractions for every kernel of the
                  pass; `traffic` comes from profiles/pmc_traffic.json and is null unless that table was measured on exactly
                  the kernel sources this run executes (clair_amd/build.py: csrc_digest).
  roofline_path.fabric_tb_s -- measured L2 <-> fabric bytes per candidate x the sustained rate, and its fraction of the 6.29 TB/s a
                  streaming copy achieves: the closest roof of the whole line (profiles/r06_fabric_sensitivity.txt: what it costs).
  cpu_baseline -- the blocked CPU port of the same forward pass (oracle/clair_cpu_port.c when present, else
                  oracle/clair_oracle.c) timed on the host cores of this box on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from clair_amd import _capi, shard, synth, weights  # noqa: E402

FLOP_PER_CANDIDATE = 40386432          # 2 x 20 193 216 MAC
BYTES_PER_CANDIDATE = 4584             # 4 224 B in + 360 B out
KERNEL_FLOP = {                         # algorithmic FLOP per candidate, per kernel
    "proj1": 0,                              # fused into lstm1
    "lstm1": 2 * 33 * 2 * (32 + 128) * 512,   # input projection + recurrence
    "proj2": 2 * 33 * 2 * 256 * 512,
    "lstm2": 2 * 33 * 2 * 128 * 512,
    "l3": 0,                                 # fused into l4
    "l4": 2 * 256 * 33 * 30 + 2 * 7680 * 192,
    "tail": 2 * (4 * 192 * 96 + 96 * 90),
}
# Bytes each kernel of THIS design moves through HBM per candidate (the intermediates it reads and
# writes once) -- reported next to the measured PMC traffic; SURVEY 8(d)'s algorithmic bytes are 4 584 B for the whole path.
DESIGN_BYTES = {
    "proj1": 0,
    "lstm1": 33 * 32 * 4 + 33 * 256 * 4,
    "proj2": 33 * 256 * 4 + 33 * 1024 * 4,
    "lstm2": 33 * 1024 * 4 + 33 * 256 * 4,
    "l3": 0,
    "l4": 33 * 256 * 4 + 8 * 192 * 4,
    "tail": 8 * 192 * 4 + 90 * 4,
}
# HBM bytes per launch come from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled as
# MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_summary.py traffic --json writes profiles/pmc_traffic.json).  Not collected
# live: a counter pass serialises kernels and cannot share a process with the timed run.  Every entry is stamped with the digest of
# the kernel sources it was measured on; an entry measured on other sources is NOT reported (traffic: null).
PMC_TABLE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def load_pmc(batch, fused):
    """({kernel: bytes per launch}, provenance string) for this batch size, or (None, why not)."""
    from clair_amd import build
    try:
        doc = json.load(open(PMC_TABLE))
    except (OSError, ValueError) as e:
        return None, "no PMC table (%s)" % e
    entry = doc.get("entries", {}).get(str(batch))
    if entry is None:
        return None, "profiles/pmc_traffic.json has no entry for batch %d" % batch
    here = build.csrc_digest()
    if entry.get("csrc_digest") != here:
        return None, "profiles/pmc_traffic.json batch %d was measured on kernel sources %s (git %s); this run executes %s: not reported" % (
            batch, entry.get("csrc_digest"), entry.get("git"), here)
    table = dict(entry.get("kernels", {}))
    if fused:
        if "layer2_fused" not in entry.get("fused", {}):
            return None, "profiles/pmc_traffic.json batch %d has no entry for the fused layer-2 launch" % batch
        table.pop("proj2", None)
        table["lstm2"] = entry["fused"]["layer2_fused"]
    return table, "%s (kernel sources %s)" % (entry.get("source"), here)


def load_rocprof_ms(batch):
    """{kernel: mean ms in the timed leg} from the committed `rocprofv3 --kernel-trace --stats` summary of this configuration, when
    profiles/pmc_traffic.json carries one stamped with the digest of the kernel sources this run executes; else None."""
    from clair_amd import build
    try:
        entry = json.load(open(PMC_TABLE)).get("entries", {}).get(str(batch))
    except (OSError, ValueError):
        return None
    if not entry or entry.get("csrc_digest") != build.csrc_digest():
        return None
    return entry.get("kernel_ms_rocprof")


PEAK_FP32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, spec
PEAK_F16_MFMA_TFLOPS = 2500.0           # MI355X_MICROARCH.md: f16/bf16 MFMA dense (AMD's 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
ACHIEVABLE_FABRIC_TBS = 6.29            # MI355X_MICROARCH.md: what a streaming copy achieves through L2 <-> fabric <-> HBM
# energy roofline of the 2-way fp16 split: MFMAs executed per candidate (three per fp32-grade product block of 32 x 32 x 16;
# L3's K padded 33 -> 48), the joules one costs, the board's power cap and idle draw
MFMA_PER_CANDIDATE = (64 * 4 * 33 * (120 + 96) + (17301504 + 2 * 1474560 + 2 * 82368) * 1024 * 3 // 32768 + 253440 * 2 * 1024 * 3 * 48 // 33 // 32768) / 1024.0
NJ_PER_MFMA, BOARD_CAP_W, BOARD_IDLE_W = 23.7, 1385.0, 247.0
WARM_STEPS = int(os.environ.get("BENCH_WARM_STEPS", "256"))   # untimed device warm-up before the contract's W warm-up steps
SPLIT_TERMS = 3                         # fp16 MFMAs executed per algorithmic fp32 product (2-way split, common.hip.h)
PLATFORM = {"ont": "ONT 122HD34", "pacbio_ccs": "PacBio CCS 15", "illumina": "Illumina 12345"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=196)      # 196 x 1024 ~= 200k chr20 candidate sites
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=4, help="forward passes in flight on the resident path (one compute lane each; the runtime has four hardware queues)")
    ap.add_argument("--platform", default="ont", choices=sorted(PLATFORM))
    ap.add_argument("--unique-batches", type=int, default=8, help="distinct synthetic batches kept resident")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--candidates", type=int, default=5000000, help="--scaling strong: size of the fixed candidate set")
    ap.add_argument("--shard-of", default=None, metavar="R/W", help="--scaling strong on ONE GPU: run exactly the block rank R of a W-rank job would be dealt "
                    "(clair_amd/shard.py: shard_batches(--candidates, --batch, R, W), its ragged last batch included); per_rank[0] then reads as that "
                    "rank's entry of the W-rank line would.  How configs[3] (5 M ONT candidates / 8 GPUs) and configs[4] (>= 1 M Illumina candidates, batch 8192 / 8 GPUs) are run on a one-GPU box")
    ap.add_argument("--full-candidates", type=int, default=200704, help="size of the untimed-by-contract leg over the whole candidate set of the config (0: skip)")
    ap.add_argument("--boundary-slots", type=int, default=6, help="batches in flight at the host-array boundary (0: skip the boundary legs)")
    ap.add_argument("--sustained-seconds", type=float, default=2.5, help="length of the value_sustained leg (0: skip)")
    ap.add_argument("--gt-candidates", type=int, default=200000, help="candidates PER PLATFORM PROFILE of the GT concordance count against the oracle (N=1 only; 0: skip)")
    ap.add_argument("--gt-seconds", type=float, default=240.0, help="time budget of that count, enforced inside each platform's loop: platform k of 3 starts no chunk after "
                    "k/3 of it and says `truncated` (200 000 candidates take 50-60 s per platform on the GPU boxes seen so far)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle spot check and the 1 024-candidate decode (timing-only builds whose results are garbage: tools/gpu/nozx_variants.sh)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


def spawn_and_relay(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU), relay rank 0's JSON line; every rank's
    stderr comes through line by line with a "[rank r]" prefix."""
    import threading
    procs = shard.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, stderr_pipe=True)

    def relay(r, pipe):
        for line in iter(pipe.readline, b""):
            sys.stderr.write("[rank %d] %s" % (r, line.decode(errors="replace")))
            sys.stderr.flush()

    threads = [threading.Thread(target=relay, args=(r, p.stderr), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    out = procs[0].stdout.read().decode()
    rcs = [p.wait() for p in procs]
    for t in threads:
        t.join(timeout=5)
    sys.stdout.write(out)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % ", ".join("rank %d rc=%d" % b for b in bad))
        return 1
    return 0


def cpu_baseline(w, x, seconds):
    """Time the CPU port of the same forward pass on the host cores, on a bounded sample of the same workload:
    all cores for about `seconds`, plus the reference's default of 4 threads (README.md:178) on a smaller sample."""
    from oracle import c_oracle
    port = getattr(c_oracle, "port_forward", None)
    fwd = port if port is not None else c_oracle.forward
    what = "oracle/clair_cpu_port.c (blocked: 48 candidates per GEMM block)" if port is not None else "oracle/clair_oracle.c"
    host_threads = c_oracle.max_threads()
    cores = c_oracle.usable_threads()          # capped by the cgroup CPU quota: more OpenMP threads than that only get throttled

    def timed(threads, budget, n):
        fwd(w, x[:min(96, x.shape[0])], threads=threads)          # warm-up (library load, thread pool)
        done, t0 = 0, time.perf_counter()
        while True:
            fwd(w, x[:n], threads=threads)
            done += n
            dt = time.perf_counter() - t0
            if dt >= budget:
                return done / dt, done, dt

    per_call = 96 * cores                     # two 48-candidate blocks per thread and call
    if x.shape[0] < per_call:
        x = np.concatenate([x] * ((per_call + x.shape[0] - 1) // x.shape[0]))[:per_call]
    rate_all, n_all, dt_all = timed(cores, seconds, x.shape[0])
    rate_4, n_4, dt_4 = timed(4, min(seconds, 5.0), min(x.shape[0], 1536))
    return {"value": round(rate_all, 1), "unit": "candidates/s", "cores": cores, "kind": "port",
            "sample": "%d candidates of the same synthetic batches, %s, OpenMP over %d threads (host: %d hardware threads, cgroup CPU "
                      "quota: %d), %.1f s" % (n_all, what, cores, host_threads, cores, dt_all),
            "value_4_threads": round(rate_4, 1),
            "sample_4_threads": "%d candidates, 4 OpenMP threads (the reference's default --threads), %.1f s" % (n_4, dt_4)}


def gt_concordance_at_scale(device, w, n_per_platform, budget_s):
    """tools/gt_concordance.py's count, in the bench line: per platform profile, the VCF rows (CHROM/POS/REF/ALT/GT) that differ between the
    decode of the HIP probabilities and the decode of the float32 oracle's on the same `n_per_platform` synthetic candidates; every flip
    analysed by tools/gt_ties.py (is the pair one float32 cannot decide, and does float64 decide it the HIP way: `flips_not_excused` must
    be 0), beside the number of candidates that are inherently ambiguous (`near_ties`: winner and runner-up closer than a perturbation eps
    of the probabilities moves them, eps = 0 / 3e-6 / 1e-5).  The oracle (test infrastructure, here as the checker) runs at a few thousand
    candidates per second on the host cores, so the count has a time budget ENFORCED INSIDE each platform's loop: platform k of 3 starts
    no 32 768-candidate chunk after k/3 of the budget, says `truncated` and reports the candidates it did compare."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gt_concordance", os.path.join(ROOT, "tools", "gt_concordance.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    eng = _capi.Engine(device=device, max_batch=4096, n_slots=1)
    eng.load_weights(w)
    out, t0 = {"candidates_per_platform": n_per_platform, "budget_s": budget_s, "platforms": {},
               "tool": "tools/gt_concordance.py: concordance(seed 777, one slot, batch 4096); flips analysed by tools/gt_ties.py"}, time.perf_counter()
    names = ("ont", "pacbio_ccs", "illumina")
    try:
        for k, platform in enumerate(names):
            r = tool.concordance(eng, w, platform, n_per_platform, 777, log=lambda *a: None, deadline=t0 + budget_s * (k + 1) / len(names))
            out["platforms"][platform] = {"candidates": r["candidates"], "vcf_rows": r["vcf_rows"], "gt_flips": r["gt_flips"], "max_abs_dp": r["max_abs_dp"],
                                          "excursions_beyond_1e-5": len(r["excursions"]),
                                          "flips_resolved_by_float64_the_hip_way": sum(1 for f in r["flips"] if f["float64_sides_with_hip"] or f["float64_tie"]),
                                          "flips_not_excused": r["flips_not_excused"], "near_ties": r["near_ties"],
                                          "flips_among_near_ties_at_1e-5": r["flips_among_near_ties_at_1e-5"],
                                          "flips": [{"position": f["hip"].split("\t")[1] if f["hip"] else None, "hip": f["hip_outcome"], "oracle32": f["oracle32_outcome"],
                                                     "oracle64": f["oracle64_outcome"], "margin_o32": f.get("margin_o32"), "margin_o64_towards_hip": f.get("margin_o64_towards_hip"),
                                                     "eps_o32_vs_o64": f["eps_o32_vs_o64"], "sensitivity": f.get("sensitivity"), "excused": f["excused"]} for f in r["flips"]]}
            if "truncated" in r:
                out["platforms"][platform]["truncated"] = r["truncated"]
    finally:
        eng.close()
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


class GpuStateSampler(object):
    """sclk / socket power of this rank's GPU while a leg runs: tools/gpu_state_sampler.py as a child process writing time-stamped
    samples (CLOCK_MONOTONIC, shared by all processes of the box); mean over the samples inside [t0, t1], or the nearest one."""

    def __init__(self, device):
        import subprocess
        import tempfile
        fd, self.path = tempfile.mkstemp(prefix="clair_gpu_state_", suffix=".txt")
        os.close(fd)
        self.proc = None
        # every read of pp_dpm_sclk / power1_* is a query to the SMU: at 1 kHz the sampler itself could move the clocks it reports,
        # inside the contract's timed region.  10 ms: the 2 ms burst leg gets the nearest sample, every other leg dozens.
        self.period_ms = float(os.environ.get("BENCH_SAMPLER_MS", "10"))
        card = "auto"
        bdf = shard.local_pci_bus_id(device)   # the sysfs directory of THIS HIP device (clair_device_pci_bus_id)
        if bdf and os.path.isdir("/sys/bus/pci/devices/%s" % bdf):
            card = "/sys/bus/pci/devices/%s" % bdf
        try:
            self.proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gpu_state_sampler.py"), card, self.path, "%g" % self.period_ms], stdin=subprocess.PIPE)
        except OSError:
            pass

    def close(self):
        self.samples, self.source = [], None
        if self.proc is None:
            return
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=5)
        except Exception:
            self.proc.kill()
        try:
            for line in open(self.path):
                if line.startswith("#"):
                    self.source = line[1:].strip()
                    continue
                t, mhz, w = line.split()
                self.samples.append((int(t), float(mhz), float(w)))
            os.remove(self.path)
        except (OSError, ValueError):
            pass

    def during(self, t0, t1):
        inside = [s_ for s_ in self.samples if t0 <= s_[0] <= t1]
        how = "%d samples inside the leg" % len(inside)
        if not inside and self.samples:
            mid = (t0 + t1) // 2
            inside = [min(self.samples, key=lambda s_: abs(s_[0] - mid))]
            how = "nearest sample, %.1f ms from the middle of the leg" % (abs(inside[0][0] - mid) / 1e6)
        mhz = [s_[1] for s_ in inside if s_[1] > 0]
        w = [s_[2] for s_ in inside if s_[2] > 0]
        return {"sclk_mhz": round(sum(mhz) / len(mhz)) if mhz else None, "power_w": round(sum(w) / len(w), 1) if w else None, "samples": how}


def boundary_legs(args, group, eng_resident, device, w, x, xd, od, batch, nuniq, steps, timed, full_steps):
    """value_boundary: the timed loop again through clair_submit / clair_wait on host arrays.  A second handle with `--boundary-slots`
    slots over three compute lanes (a handle with more than four slots: the fourth hardware queue is the incoming copy stream's); every batch is a pageable NumPy array (a different one per step, `nuniq` of them) and every
    result four fresh NumPy arrays, as clair/model.py:946-966 returns them.  The last batch's outputs are compared bit for bit with the
    resident path's."""
    world = group.world
    slots = args.boundary_slots
    eng = _capi.Engine(device=device, max_batch=batch, n_slots=slots)
    eng.load_weights(w)
    xs = [np.ascontiguousarray(x[i * batch:(i + 1) * batch]) for i in range(nuniq)]
    cs = []
    for a in xs:                       # the raw counts these tensors stand for (clair/utils.py:96-98 undone): exactly representable
        c = a.copy()
        c[..., 1:] += c[..., 0:1]
        cs.append(c.astype(np.int16))
    last = [None, None]                # (step index, outputs) of the batch that came back last
    pending = {}                       # slot -> step index

    def loop(k, counts):
        for i in range(k):
            s = i % slots
            if s in pending:
                last[:] = [pending.pop(s), eng.wait(s)]
            if counts:
                eng.submit_counts(s, cs[i % nuniq])
            else:
                eng.submit(s, xs[i % nuniq])
            pending[s] = i

    def drain():
        for s in sorted(pending, key=pending.get):
            last[:] = [pending.pop(s), eng.wait(s)]

    phases = []                        # forward passes launched, in order (tools/rocpd_summary.py --phases-from)
    out = {"slots": slots, "lanes": slots if slots <= 4 else 3, "steps": steps,
           "interface": "pageable NumPy batches [n,33,8,4] in, four fresh NumPy arrays per batch out (clair_submit / clair_submit_counts + clair_wait); "
                        "staging copy and enqueue on the engine's staging threads, H2D on one copy stream, results written to page-locked host "
                        "memory by a kernel on the lane"}
    # device warm-up outside the contract's W steps, as before `value`: the handle's first-use allocations, and the clock the chip dropped to
    # while this handle was being set up
    loop(max(0, WARM_STEPS - args.warmup), False)
    drain()
    phases.append(["boundary handle: device warm-up", max(0, WARM_STEPS - args.warmup)])
    for name, counts, k in (("float32", False, steps), ("int16", True, steps), ("float32_full", False, full_steps), ("int16_full", True, full_steps)):
        if k <= 0:
            continue
        loop(args.warmup, counts)
        drain()
        secs = group.max_float(timed("value_boundary" + ("" if name == "float32" else "_" + name), lambda: loop(k, counts), drain))
        out[name] = {"value": round(world * k * batch / secs, 1), "steps": k, "ms_per_step": round(secs / k * 1e3, 4),
                     "h2d_bytes_per_candidate": 2112 if counts else 4224}
        phases += [["boundary %s: warm-up" % name, args.warmup], ["boundary %s: timed (value_boundary%s)" % (name, "" if name == "float32" else "_" + name), k]]
    # the last batch of the last leg against the resident path on the same candidates
    i = (last[0] or 0) % nuniq
    eng_resident.run_resident(0, xd, od, i * batch, batch)
    eng_resident.sync()
    ref = _capi.split_outputs(eng_resident.dataset_download(od, i * batch, batch))
    out["bit_identical_to_resident"] = bool(last[1] is not None and all(np.array_equal(a, b) for a, b in zip(last[1], ref)))
    phases.append(["boundary: the resident launch its last batch is compared with", 1])
    out["launch_phases"] = phases
    eng.close()
    return out


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_and_relay(args)
    # stdout carries exactly one JSON line: native libraries that print there (RCCL's start-up banner does) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # RCCL through the C ABI when WORLD_SIZE > 1.  BENCH_SHARE_DEVICE=1 (a TEST of the multi-rank flow on a box with one GPU, never a
    # measurement): every rank runs on device 0 and the ranks talk over the socket transport -- no rank holds an RCCL communicator, so the
    # line says n_gpus 0 and the exit code is non-zero, as for any run without RCCL.
    share = os.environ.get("BENCH_SHARE_DEVICE") == "1"
    # BENCH_SHARE_TRANSPORT=rccl (with a stand-in librccl, CLAIR_AMD_RCCL_LIBRARY: real RCCL refuses two ranks on one GPU): the RCCL
    # bring-up itself under test -- tests/test_comm_gpu.py hangs it and expects the line all the same
    group = shard.NodeGroup(transport=os.environ.get("BENCH_SHARE_TRANSPORT", "tcp"), device=0) if share else shard.NodeGroup()
    try:
        rc = run_ranked(args, group, json_fd)
    except BaseException:
        group.close(barrier=False)     # unwinding: the peers may be gone, do not wait for them
        raise
    group.close()
    if group.rccl_abandoned:           # a helper thread is still inside a hung RCCL bring-up: the line is out, do not wait for librccl's tear-down
        group.exit_process(rc)
    return rc


def run_ranked(args, group, json_fd):
    t_start = time.perf_counter()
    rank, world, local_rank = group.rank, group.world, group.device          # the HIP device of this rank (= LOCAL_RANK unless BENCH_SHARE_DEVICE)
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d\n" % (args.gpus, world, world))

    batch, streams = args.batch, max(1, args.streams)
    w = weights.synthetic_weights(seed=20250928, head_gain=4.0) if rank == 0 else None
    if world > 1:
        w = group.broadcast_weights(w, root=0)          # 9.5 MB over RCCL, once
    eng = _capi.Engine(device=local_rank, max_batch=batch, n_slots=streams)
    eng.load_weights(w)

    steps = args.steps
    mine_candidates = args.steps * batch
    stands_for = None
    if args.scaling == "strong":
        as_rank, as_world = rank, world
        if args.shard_of:
            if world != 1:
                raise SystemExit("--shard-of runs one rank's block on one GPU: use it with --gpus 1")
            try:
                as_rank, as_world = (int(v) for v in args.shard_of.split("/"))
            except ValueError:
                raise SystemExit("--shard-of takes R/W (two integers), not %r" % args.shard_of)
            if not 0 <= as_rank < as_world:
                raise SystemExit("--shard-of R/W needs 0 <= R < W")
            stands_for = {"rank": as_rank, "world": as_world, "of_candidates": args.candidates}
        first_candidate, mine_candidates = shard.shard_batches(args.candidates, batch, as_rank, as_world)
        if stands_for:
            stands_for["first_candidate"] = first_candidate
        steps = (mine_candidates + batch - 1) // batch
    steps_max = int(round(group.max_float(steps)))
    nuniq = max(1, min(args.unique_batches, steps_max + args.warmup))
    x, infos = synth.synthetic_input(nuniq * batch, args.platform, seed=20250928 + rank)
    xd, od = eng.dataset_alloc(nuniq * batch)
    eng.dataset_upload(xd, 0, x)
    last_n = mine_candidates - (steps - 1) * batch if steps else 0       # strong scaling: the set's ragged last batch is run ragged

    def run(k, ragged_last=False):
        for i in range(k):
            eng.run_resident(i % streams, xd, od, (i % nuniq) * batch, last_n if ragged_last and i == k - 1 else batch)

    # device warm-up outside the contract's W warm-up steps: first-touch of the workspaces, clock ramp, code upload
    device_warm = max(0, WARM_STEPS - args.warmup)
    run(device_warm)
    eng.sync()
    run(args.warmup)
    eng.sync()
    phases = [["device warm-up + --warmup", device_warm + args.warmup]]    # forward passes launched, in order (tools/rocpd_summary.py --phases-from)
    sampler = GpuStateSampler(local_rank)       # every rank samples ITS GPU (an 8-rank line carries eight clocks and powers)
    legs = {}                                   # name -> (CLOCK_MONOTONIC ns at both ends) for gpu_state

    def timed(name, body, finish):
        """The contract's bracket: barrier + device sync, the body, device sync + barrier; -> this rank's seconds."""
        group.barrier()
        eng.sync()
        m0, t_0 = time.monotonic_ns(), time.perf_counter()
        body()
        finish()
        secs = time.perf_counter() - t_0
        legs[name] = (m0, time.monotonic_ns())
        group.barrier()
        return secs

    # Timed region: the plain hot path, no instrumentation.
    eng.timing_enable(False)
    mine_s = timed("value", lambda: run(steps, ragged_last=True), eng.sync)
    phases.append(["timed (value), %d lanes in flight" % streams, steps])
    elapsed = group.max_float(mine_s)          # the slowest rank defines the job time
    per_rank_s = group.gather_floats(mine_s)
    per_rank_steps = [int(round(v)) for v in group.gather_floats(steps)]
    per_rank_candidates = [int(round(v)) for v in group.gather_floats(mine_candidates)]
    rccl_ranks = int(round(sum(group.gather_floats(1.0 if group.transport == "rccl" else 0.0))))

    # The whole candidate set of the config, untimed by the contract: what the rate is when the run is not a handful of steps.
    full = None
    if args.full_candidates > 0 and args.scaling == "weak":
        full_steps = (args.full_candidates + batch - 1) // batch
        full_s = group.max_float(timed("value_full_config", lambda: run(full_steps), eng.sync))
        phases.append(["whole candidate set (value_full_config)", full_steps])
        full = {"steps": full_steps, "candidates_per_rank": full_steps * batch, "seconds": round(full_s, 6), "value": round(world * full_steps * batch / full_s, 1),
                "ms_per_step": round(full_s / full_steps * 1e3, 4)}

    # The sustained rate: the same loop for seconds, not milliseconds -- the board at its power cap from start to end.
    sustained = None
    if args.sustained_seconds > 0 and args.scaling == "weak":
        guess = (full["value"] if full else sum(per_rank_candidates) / elapsed) / world        # candidates/s of one rank
        sus_steps = max(steps, int(np.ceil(args.sustained_seconds * guess / batch)))
        sus_steps = int(round(group.max_float(sus_steps)))                                      # the same count on every rank
        sus_s = group.max_float(timed("value_sustained", lambda: run(sus_steps), eng.sync))
        phases.append(["sustained (value_sustained)", sus_steps])
        sustained = {"steps": sus_steps, "candidates_per_rank": sus_steps * batch, "seconds": round(sus_s, 6), "value": round(world * sus_steps * batch / sus_s, 1),
                     "ms_per_step": round(sus_s / sus_steps * 1e3, 4)}

    # The same steps through the reference's own boundary: host arrays in, host arrays out (clair_submit / clair_wait), timed like `value`.
    boundary = None
    if args.boundary_slots > 0:
        boundary = boundary_legs(args, group, eng, local_rank, w, x, xd, od, batch, nuniq, steps, timed, full["steps"] if full else 0)
        phases += boundary.pop("launch_phases")

    # Per-kernel tables, outside the timed region.  (a) the same loop, same streams, every kernel bracketed by HIP events (ten marker
    # packets per pass); (b) the same on ONE stream, so that a kernel's HIP-event duration is its own ("alone").
    fused = eng.kernel_workgroups(batch)["proj2"] == 0      # layer 2 as one launch: its events carry id "lstm2"
    eng.timing_enable(True)
    eng.timing_reset()
    run(steps)
    eng.sync()
    times = eng.kernel_times()
    phases.append(["HIP events on every kernel, %d lanes in flight" % streams, steps])
    iso_steps = min(steps, 32)
    times_iso = None
    for _ in range(2):                     # twice, the smaller mean per kernel: one hiccup in 32 launches (a 4 ms stall was seen once) must not pick the "dominant" kernel
        eng.timing_reset()
        for i in range(iso_steps):
            eng.run_resident(0, xd, od, (i % nuniq) * batch, batch)
        eng.sync()
        t_iso = eng.kernel_times()
        times_iso = t_iso if times_iso is None else {k: (min(t_iso[k], times_iso[k], key=lambda v: v[0] / v[1] if v[1] else float("inf"))) for k in t_iso}
    phases.append(["alone on one stream (alone_ms), twice", 2 * iso_steps])
    wgs = eng.kernel_workgroups(batch)
    active = [k for k in _capi.KERNEL_NAMES if times_iso[k][1]]
    cu_share = {k: min(1.0, wgs[k] / 256.0) if wgs[k] else 1.0 for k in active}
    alone_ms = {k: times_iso[k][0] / times_iso[k][1] for k in active}
    chip_time = {k: alone_ms[k] * cu_share[k] for k in active}
    # (c) the dominant kernel = most chip time in table (b), MEASURED in this run; the loop again with a HIP-event pair around that
    # kernel only (two marker packets per pass): its duration in the configuration the timed region ran in = roofline.frac
    dominant = max(chip_time, key=chip_time.get)
    eng.timing_enable(only=[dominant])
    eng.timing_reset()
    t1 = time.perf_counter()
    run(steps)
    eng.sync()
    elapsed_dom = time.perf_counter() - t1
    phases.append(["HIP events on the dominant kernel only (roofline.kernel_ms)", steps])
    dom_ms, dom_cnt = eng.kernel_times()[dominant]
    eng.timing_enable(False)

    # parity spot check of one resident batch against the oracle (outside the timed region)
    parity = concord = None
    if rank == 0 and not args.no_parity:
        from clair_amd import call_var as cvar
        from oracle import c_oracle
        ns = min(1024, batch * nuniq)
        eng.run_resident(0, xd, od, 0, min(ns, batch))
        eng.sync()
        phases.append(["parity check against the oracle", 1])
        ns = min(ns, batch)
        got = _capi.split_outputs(eng.dataset_download(od, 0, ns))
        want = c_oracle.forward(w, x[:ns])
        parity = max(float(np.abs(g - t_).max()) for g, t_ in zip(got, want))
        # VCF GT concordance: decode both probability sets with the same decoder, compare CHROM/POS/REF/ALT/GT
        dec = cvar.VariantDecoder(cvar.OutputConfig(True, False, False, False, False, None))
        key = lambda r: (r.split("\t")[:5], r.split("\t")[-1].split(":")[0])  # noqa: E731
        rows_g = dec.decode_batch(x[:ns], infos[:ns], got)
        rows_w = dec.decode_batch(x[:ns], infos[:ns], want)
        same = len(rows_g) == len(rows_w) and all(key(a) == key(b) for a, b in zip(rows_g, rows_w))
        flips = sum(key(a) != key(b) for a, b in zip(rows_g, rows_w)) if len(rows_g) == len(rows_w) else None
        concord = {"candidates": ns, "vcf_rows": len(rows_w), "gt_identical": bool(same), "gt_flips": flips}

    # every rank's GPU state per leg and CPU placement, to rank 0 (bookkeeping over the bootstrap sockets; JSON text, the wire format has no floats)
    sampler.close()
    my_state = {name: sampler.during(*span) for name, span in legs.items()}
    my_state["source"], my_state["period_ms"] = sampler.source, sampler.period_ms
    rank_notes = [json.loads(t_) for t_ in group.gather_objects(json.dumps({"gpu_state": my_state, "affinity": getattr(group, "affinity", None)}))]

    gt200k = None
    if rank == 0 and world == 1 and args.gt_candidates > 0:
        gt200k = gt_concordance_at_scale(local_rank, w, args.gt_candidates, args.gt_seconds)
        phases.append(["GT concordance at scale (a handle of its own: one slot, batch 4096)", None])

    rc = 0
    if rank == 0:
        total = sum(per_rank_candidates)          # real candidates: a shard's ragged last batch counts what it holds
        value = total / elapsed
        kflop = dict(KERNEL_FLOP)
        if fused:
            kflop["lstm2"] = KERNEL_FLOP["proj2"] + KERNEL_FLOP["lstm2"]
        design = dict(DESIGN_BYTES)
        if fused:
            design["lstm2"] = 33 * 256 * 4 + 33 * 1024 * 4 + 33 * 256 * 4
        pmc, pmc_note = load_pmc(batch, fused)
        rocprof_ms = load_rocprof_ms(batch)
        kern = {k: {"ms_mean": (round(ms / cnt, 5) if cnt else None), "launches": cnt} for k, (ms, cnt) in times.items()}
        kern_iso = {k: round(alone_ms[k], 5) if k in alone_ms else None for k in times_iso}

        def frac(flop, ms):
            return flop / (ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS if ms else None

        per_kernel = {}
        for k in active:
            in_ms = times[k][0] / times[k][1] if times[k][1] else None
            fl = kflop[k] * batch
            per_kernel[k] = {"algorithmic_flop_per_launch": fl, "in_flight_ms": round(in_ms, 5) if in_ms else None, "alone_ms": round(alone_ms[k], 5),
                             "frac": round(frac(fl, in_ms), 4) if in_ms else None, "alone_frac": round(frac(fl, alone_ms[k]), 4),
                             "rocprof_ms": rocprof_ms.get(k) if rocprof_ms else None,
                             "frac_rocprof": round(frac(fl, rocprof_ms[k]), 4) if rocprof_ms and rocprof_ms.get(k) else None,
                             "workgroups": wgs[k], "cu_share": round(cu_share[k], 4),
                             "chip_time_share_alone": round(chip_time[k] / max(sum(chip_time.values()), 1e-12), 3),
                             "traffic": round(pmc[k]) if pmc and k in pmc else None, "design_bytes_per_launch": design[k] * batch}
        dom = dominant
        # ONE duration for the dominant kernel: its mean over the timed loop repeated with HIP events on every kernel (table `kernels`,
        # the same number as kernels[dom].in_flight_ms).  The loop with events around this kernel only is kept as `kernel_ms_with_events_on_it_only`.
        dom_ms_mean = times[dom][0] / max(times[dom][1], 1)
        dom_ms_only = dom_ms / max(dom_cnt, 1)
        flop = kflop[dom] * batch
        tf = flop / (dom_ms_mean * 1e-3) / 1e12
        tf_alone = flop / (alone_ms[dom] * 1e-3) / 1e12
        traffic = pmc.get(dom) if pmc else None
        names = {"lstm1": "lstm32_kernel<true>", "proj2": "gemm_split_kernel", "lstm2": "lstm2_fused_kernel (proj2 + lstm2 in one launch)" if fused else "lstm32_kernel<false> / lstm32_pair_kernel",
                 "l4": "l3l4_kernel", "tail": "tail_kernel"}
        roof = {
            "bound": "mfma", "kernel": "%s (%s)" % (dom, names.get(dom, dom)), "achieved": round(tf, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_F16_MFMA_TFLOPS, 4), "traffic": round(traffic) if traffic else None,
            "traffic_source": pmc_note,
            "definition": "dominant kernel = most chip time (stand-alone HIP-event duration x share of the 256 CUs its grid occupies) in this run's own "
                          "table (`kernels`); frac = SURVEY.md 8(d) algorithmic FLOP of that kernel per launch (%d per candidate x batch) / its mean "
                          "HIP-event duration with %d batches in flight (the timed loop repeated with HIP events on every kernel: the same number as "
                          "kernels[dominant].in_flight_ms and .frac) / dense f16 MFMA peak; the rocprofv3 mean of the same kernel in the same loop is "
                          "`kernel_ms_rocprof` when profiles/pmc_traffic.json was stamped on these kernel sources" % (kflop[dom], streams),
            "kernel_ms": round(dom_ms_mean, 5), "launches": times[dom][1], "algorithmic_flop_per_launch": flop,
            "kernel_ms_with_events_on_it_only": round(dom_ms_only, 5),
            "kernel_ms_rocprof": rocprof_ms.get(dom) if rocprof_ms else None,
            # the same fraction from the un-instrumented duration: the kernel's rocprofv3 mean in the timed region of the committed trace of this
            # configuration (HIP events cost marker packets in the queues: 0.143-0.166 ms with events against 0.133 ms traced, batch 1024)
            "frac_rocprof": round(flop / (rocprof_ms[dom] * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4) if rocprof_ms and rocprof_ms.get(dom) else None,
            "executed_frac": round(tf * SPLIT_TERMS / PEAK_F16_MFMA_TFLOPS, 4),
            "executed_note": "matmuls run as a 2-way fp16 split: 3 v_mfma_f32_32x32x16_f16 per algorithmic fp32 product block",
            "alone_kernel_ms": round(alone_ms[dom], 5), "alone_frac": round(tf_alone / PEAK_F16_MFMA_TFLOPS, 4),
            "workgroups": wgs[dom], "cu_share": round(cu_share[dom], 4),
            "frac_of_cus_held": round(tf / PEAK_F16_MFMA_TFLOPS / max(cu_share[dom], 1e-9), 4),
            "in_flight_note": "`frac` divides by the peak of the WHOLE chip and by a duration measured while %d forward passes share it: a launch that holds "
                              "%d of the 256 CUs and waits for them behind the other lanes' kernels reads low by construction (3 lanes: shorter launches, lower "
                              "throughput; profiles/r04_lanes_sweep.txt).  `frac_of_cus_held` = frac / cu_share; `alone_frac` = the same launch with the chip to "
                              "itself; `roofline_path` = the whole pass against the same peak, which is the figure that moves with `value`" % (streams, min(wgs[dom], 256)),
            "hbm_gbs_measured_traffic": round(traffic / (dom_ms_mean * 1e-3) / 1e9, 1) if traffic else None,
            "hbm_frac_measured_traffic": round(traffic / (dom_ms_mean * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if traffic else None,
            "design_bytes_per_launch": design[dom] * batch,
            "kernels": per_kernel,
            "value_with_events_on_this_kernel": round(steps * batch / elapsed_dom, 1),
        }
        path_tf = value / world * FLOP_PER_CANDIDATE / 1e12
        n_gpus = rccl_ranks if world > 1 else 1
        if world > 1 and rccl_ranks != world:
            sys.stderr.write("bench.py: %d of %d ranks hold an RCCL communicator; reporting n_gpus=%d\n" % (rccl_ranks, world, rccl_ranks))
            rc = 1
        gpu_state = rank_notes[0]["gpu_state"]
        out = {
            "metric": "candidate sites/sec (whole node)",
            "value": round(value, 1),
            "value_boundary": round(boundary["float32"]["value"], 1) if boundary else None,
            "value_boundary_int16": round(boundary["int16"]["value"], 1) if boundary else None,
            "value_full_config": round(full["value"], 1) if full else None,
            "value_sustained": round(sustained["value"], 1) if sustained else None,
            "value_boundary_full_config": round(boundary["float32_full"]["value"], 1) if boundary and "float32_full" in boundary else None,
            "unit": "candidates/s",
            "n_gpus": n_gpus,
            "steps": steps_max,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(steps_max, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32 (matmuls as 2-way fp16 split on MFMA with fp32 accumulate; gates/activations fp32)",
            "data": "synthetic",
            "config": {"workload": "%s weights-shape model (random init), synthetic %s-profile pileup tensors, "
                                   "batch=%d, %d batches in flight per GPU, inputs resident in HBM"
                                   % (PLATFORM[args.platform], args.platform, batch, streams),
                       "batch": batch, "streams": streams, "candidates_total": total,
                       "rates": {"value": "the contract's K timed steps (%d x %d candidates = %.1f ms): a burst right after the bracket's idle gap, the board not yet "
                                          "at its power cap" % (steps_max, batch, elapsed * 1e3),
                                 "value_full_config": "the config's whole candidate set (%d steps), timed the same way right after" % full["steps"] if full else None,
                                 "value_sustained": "%d steps = %.2f s timed the same way: the board at its cap throughout (gpu_state.value_sustained); the rate "
                                                    "a whole-genome run sees" % (sustained["steps"], sustained["seconds"]) if sustained else None},
                       "device_warm_steps": device_warm,
                       "device_warm_note": "untimed steps before the contract's --warmup (clock ramp, first touch); BENCH_WARM_STEPS=0 removes them",
                       "collective": "none on the data path; RCCL (clair_comm_*) for the weight broadcast, barrier and timers" if world > 1 else "none (1 rank)",
                       "stands_for": stands_for,          # --shard-of: this one-GPU run is rank R's block of a W-rank strong-scaling job
                       "transport": group.transport, "ranks_with_rccl_communicator": rccl_ranks if world > 1 else None,
                       "rccl_failure": getattr(group, "rccl_failure", None)},
            "per_rank": [{"rank": r if not stands_for else stands_for["rank"], "steps": s_, "candidates": c_, "seconds": round(t_, 6), "candidates_per_s": round(c_ / t_, 1) if t_ > 0 else None,
                          "affinity": rank_notes[r]["affinity"], "gpu_state": rank_notes[r]["gpu_state"] if world > 1 else "see gpu_state"}
                         for r, (s_, c_, t_) in enumerate(zip(per_rank_steps, per_rank_candidates, per_rank_s))],
            "boundary": boundary,
            "full_config": full,
            "sustained": sustained,
            "gpu_state": gpu_state,
            "launch_phases": phases,
            "roofline": roof,
            "roofline_path": {"achieved": round(path_tf, 2), "unit": "TFLOP/s per GPU (algorithmic, 40 386 432 FLOP / candidate)",
                              "frac_of_f16_mfma": round(path_tf / PEAK_F16_MFMA_TFLOPS, 4),
                              "executed_frac_of_f16_mfma": round(path_tf * SPLIT_TERMS / PEAK_F16_MFMA_TFLOPS, 4),
                              "frac_of_fp32_mfma": round(path_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                              "algorithmic_hbm_gbs": round(value / world * BYTES_PER_CANDIDATE / 1e9, 2),
                              "algorithmic_hbm_frac": round(value / world * BYTES_PER_CANDIDATE / 1e9 / PEAK_HBM_GBS, 6),
                              "measured_traffic_bytes_per_candidate": round(sum(pmc[k] for k in active if k in pmc) / batch) if pmc else None,
                              # the closest roof of the whole line: every intermediate crosses the L2 <-> fabric boundary (PMC FETCH_SIZE + WRITE_SIZE per
                              # candidate x the sustained rate) against what a streaming copy achieves (profiles/r06_fabric_sensitivity.txt: what it costs)
                              "fabric_tb_s": round(sum(pmc[k] for k in active if k in pmc) / batch * (sustained["value"] if sustained else value) / world / 1e12, 3) if pmc else None,
                              "fabric_frac_of_achievable": round(sum(pmc[k] for k in active if k in pmc) / batch * (sustained["value"] if sustained else value) / world / 1e12
                                                                 / ACHIEVABLE_FABRIC_TBS, 4) if pmc else None,
                              "fabric_achievable_tb_s": ACHIEVABLE_FABRIC_TBS,
                              # the board is at its power cap whatever runs: what the formulation's MFMAs alone would allow
                              "energy_bound": {"mfma_per_candidate": MFMA_PER_CANDIDATE, "nj_per_mfma": NJ_PER_MFMA, "cap_w": BOARD_CAP_W, "idle_w": BOARD_IDLE_W,
                                               "candidates_per_s_per_gpu": round((BOARD_CAP_W - BOARD_IDLE_W) / (MFMA_PER_CANDIDATE * NJ_PER_MFMA * 1e-9), 1),
                                               "frac": round(value / world / ((BOARD_CAP_W - BOARD_IDLE_W) / (MFMA_PER_CANDIDATE * NJ_PER_MFMA * 1e-9)), 4),
                                               "source": "constants measured once, not in this run: profiles/r04_lstm_energy_variants.txt (nJ per v_mfma_f32_32x32x16_f16), "
                                                         "profiles/r04_energy_by_kernel_final.txt (socket power at the cap and idle)"}},
            "kernels_in_flight_ms": kern,
            "kernels_alone_ms": kern_iso,
            "parity_max_abs_err": parity,
            "gt_concordance": concord,
            "gt_concordance_200k": gt200k,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, x, args.cpu_seconds)
        out["bench_wall_s"] = round(time.perf_counter() - t_start, 1)       # of which gt_concordance_200k.seconds and ~17 s of cpu_baseline are host work on the oracle / port
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    eng.dataset_free(xd, od)
    eng.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
