#!/usr/bin/env python3
"""Headline benchmark: candidate sites/sec of the call_var forward pass on N MI355X.

A "step" is one pass of the hot path (BiLSTM x2 -> slice dense -> dense tail -> softmax
heads) over one batch of synthetic pileup tensors already resident in HBM.  Default workload is
BASELINE.json configs[1]: ONT-profile candidates, batch 1024, one GPU.  Candidate sites shard
across ranks with no data-path collective (scaling: weak, per-rank work fixed); for N>1 the
driver launches one rank per GPU through torch.distributed.run and RCCL is used only for the
barrier / max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     -- the dominant kernel, timed with HIP events on its own stream inside the timed
                  region: algorithmic FLOP per launch / mean launch duration vs the 157.3 TFLOP/s
                  fp32-input MFMA peak (MI355X_MICROARCH.md)
  cpu_baseline -- the C port of the same forward pass (oracle/clair_oracle.c, OpenMP) timed on the
                  host cores of this box on a bounded sample (N=1 only)
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
KERNEL_FLOP = {                         # algorithmic FLOP per candidate, per kernel
    "proj1": 0,                              # fused into lstm1
    "lstm1": 2 * 33 * 2 * (32 + 128) * 512,   # input projection + recurrence
    "proj2": 2 * 33 * 2 * 256 * 512,
    "lstm2": 2 * 33 * 2 * 128 * 512,
    "l3": 0,                                 # fused into l4
    "l4": 2 * 256 * 33 * 30 + 2 * 7680 * 192,
    "tail": 2 * (4 * 192 * 96 + 96 * 90),
}
# Algorithmic HBM bytes per candidate and kernel: what the kernel must read + write once.
KERNEL_BYTES = {
    "proj1": 0,
    "lstm1": 33 * 32 * 4 + 33 * 256 * 4,                  # x in; layer output out as two fp16 planes (4 B / unit)
    "proj2": 33 * 256 * 4 + 33 * 1024 * 4,                # fp16 planes in; fp32 x-projection (fragment-major) out
    "lstm2": 33 * 1024 * 4 + 33 * 256 * 4,                # x-projection in; fp32 layer output out
    "l3": 0,
    "l4": 33 * 256 * 4 + 16 * 192 * 4,                    # layer output in; split-K partials out
    "tail": 16 * 192 * 4 + 90 * 4,
}
# HBM bytes per launch at batch 1024 from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE
# doubled as MI355X_MICROARCH.md prescribes for gfx950): profiles/r01_pmc_hbm_traffic.txt.  Not collected live: a
# counter pass serialises kernels and cannot share a process with the timed run.
PMC_TRAFFIC_BYTES_B1024 = {"lstm1": 46.0e6, "proj2": 181.6e6, "lstm2": 175.2e6, "l4": 54.6e6, "tail": 16.1e6}
PEAK_FP32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, spec
PEAK_F16_MFMA_TFLOPS = 2500.0           # MI355X_MICROARCH.md: f16/bf16 MFMA dense (AMD's 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
WARM_STEPS = int(os.environ.get("BENCH_WARM_STEPS", "256"))   # untimed device warm-up before the contract's W warm-up steps
SPLIT_TERMS = 3                         # fp16 MFMAs executed per algorithmic fp32 product (2-way split, common.hip.h)
PLATFORM = {"ont": "ONT 122HD34", "pacbio_ccs": "PacBio CCS 15", "illumina": "Illumina 12345"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=196)      # 196 x 1024 ~= 200k chr20 candidate sites
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=3, help="pipeline slots (HIP streams) with batches in flight")
    ap.add_argument("--platform", default="ont", choices=sorted(PLATFORM))
    ap.add_argument("--unique-batches", type=int, default=8, help="distinct synthetic batches kept resident")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def cpu_baseline(w, x, seconds):
    """Time the C port of the same forward pass on the host cores, on a bounded sample of the same workload:
    all cores for about `seconds`, plus the reference's default of 4 threads (README.md:178) on a smaller sample."""
    from oracle import c_oracle
    cores = c_oracle.max_threads()

    def timed(threads, budget):
        c_oracle.forward(w, x[:min(64, x.shape[0])], threads=threads)          # warm-up (library load, thread pool)
        n = x.shape[0] if threads == 0 else min(x.shape[0], 1024)   # all cores: the whole resident set per call (32 candidates per thread)
        done, t0 = 0, time.perf_counter()
        while True:
            c_oracle.forward(w, x[:n], threads=threads)
            done += n
            dt = time.perf_counter() - t0
            if dt >= budget:
                return done / dt, done, dt

    rate_all, n_all, dt_all = timed(0, seconds)
    rate_4, n_4, dt_4 = timed(4, min(seconds, 6.0))
    return {"value": round(rate_all, 1), "unit": "candidates/s", "cores": cores, "kind": "port",
            "sample": "%d candidates of the same synthetic batches, oracle/clair_oracle.c with OpenMP over %d threads, %.1f s"
                      % (n_all, cores, dt_all),
            "value_4_threads": round(rate_4, 1),
            "sample_4_threads": "%d candidates, 4 OpenMP threads (the reference's default --threads), %.1f s" % (n_4, dt_4)}


def main():
    args = parse_args()
    group = shard.NodeGroup()          # torch.distributed (RCCL) only when WORLD_SIZE > 1
    rank, world, local_rank = group.rank, group.world, group.local_rank
    barrier = group.barrier

    batch, streams = args.batch, max(1, args.streams)
    w = weights.synthetic_weights(seed=20250928, head_gain=4.0)
    eng = _capi.Engine(device=local_rank, max_batch=batch, n_slots=streams)
    eng.load_weights(w)

    nuniq = max(1, min(args.unique_batches, args.steps + args.warmup))
    x, infos = synth.synthetic_input(nuniq * batch, args.platform, seed=20250928 + rank)
    xd, od = eng.dataset_alloc(nuniq * batch)
    eng.dataset_upload(xd, 0, x)

    def run(steps):
        for i in range(steps):
            eng.run_resident(i % streams, xd, od, (i % nuniq) * batch, batch)

    # device warm-up outside the contract's W warm-up steps: first-touch of the workspaces, clock ramp, code upload
    run(max(0, WARM_STEPS - args.warmup))
    eng.sync()
    run(args.warmup)
    eng.sync()
    # Timed region: the plain hot path, no instrumentation (the HIP events of the passes below add marker packets
    # to every stream).
    eng.timing_enable(False)
    barrier()
    eng.sync()
    t0 = time.perf_counter()
    run(args.steps)
    eng.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    # Same loop again with a HIP-event pair around every kernel, on the kernel's own stream: per-kernel durations
    # with the other streams' kernels sharing the chip ("overlapped").
    eng.timing_enable(True)
    eng.timing_reset()
    t1 = time.perf_counter()
    run(args.steps)
    eng.sync()
    elapsed_events = time.perf_counter() - t1
    times = eng.kernel_times()
    # Un-overlapped pass for the per-kernel roofline: the same steps on ONE stream, so a kernel's HIP-event
    # duration is its own.
    iso_steps = min(args.steps, 32)
    eng.timing_reset()
    for i in range(iso_steps):
        eng.run_resident(0, xd, od, (i % nuniq) * batch, batch)
    eng.sync()
    times_iso = eng.kernel_times()
    eng.timing_enable(False)

    elapsed = group.max_float(elapsed)          # the slowest rank defines the step time

    # parity spot check of one resident batch against the oracle (outside the timed region)
    parity = concord = None
    if rank == 0:
        from clair_amd import call_var as cvar
        from oracle import c_oracle
        ns = min(1024, batch * nuniq)
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

    if rank == 0:
        total = args.steps * batch * world
        value = total / elapsed
        kern = {k: {"ms_mean": (ms / cnt if cnt else None), "launches": cnt} for k, (ms, cnt) in times.items()}
        kern_iso = {k: round(ms / cnt, 5) if cnt else None for k, (ms, cnt) in times_iso.items()}
        # dominant = most chip time: duration x share of the 256 CUs its grid occupies (the recurrent kernels
        # launch 2 workgroups per 32-candidate tile, one per CU: 64 CUs at batch 1024; with several batches in
        # flight the projection GEMM is launched on half the chip, see clair_engine_create)
        wgs = eng.kernel_workgroups(batch)
        cu_share = {k: min(1.0, wgs[k] / 256.0) if wgs[k] else 1.0 for k in times_iso}
        dom = max(times_iso, key=lambda k: times_iso[k][0] * cu_share[k])
        dom_ms = times_iso[dom][0] / max(times_iso[dom][1], 1)
        ovl_ms = times[dom][0] / max(times[dom][1], 1)
        flop, byts = KERNEL_FLOP[dom] * batch, KERNEL_BYTES[dom] * batch
        tf = flop / (dom_ms * 1e-3) / 1e12
        gbs = byts / (dom_ms * 1e-3) / 1e9
        # which ceiling binds this kernel: the larger of its two minimum times (matmuls run as 3 fp16 MFMAs per
        # algorithmic product, so the matrix ceiling for algorithmic FLOP is the f16 dense peak / 3)
        t_mfma = flop * SPLIT_TERMS / (PEAK_F16_MFMA_TFLOPS * 1e12)
        t_hbm = byts / (PEAK_HBM_GBS * 1e9)
        if t_hbm >= t_mfma:
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None}
        else:
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(tf, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / PEAK_F16_MFMA_TFLOPS, 4), "traffic": None}
        roof["traffic"] = round(PMC_TRAFFIC_BYTES_B1024[dom] * batch / 1024) if dom in PMC_TRAFFIC_BYTES_B1024 else None
        roof.update({
            "traffic_source": "profiles/r01_pmc_hbm_traffic.txt (rocprofv3 PMC passes at batch 1024, scaled by batch/1024)",
            "kernel_ms": round(dom_ms, 4), "algorithmic_flop_per_launch": flop, "algorithmic_bytes_per_launch": byts,
            "algorithmic_tflops": round(tf, 2), "algorithmic_gbs": round(gbs, 1),
            "mfma_frac_executed": round(tf * SPLIT_TERMS / PEAK_F16_MFMA_TFLOPS, 4),
            "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
            "note": "matmuls run as 2-way fp16 split: 3 v_mfma_f32_32x32x16_f16 per algorithmic fp32 product block",
            "measured": "HIP events on the kernel's stream, %d launches on one stream right after the timed region "
                        "(no other stream active): the 'alone' column of profiles/r01_bench_default_streams3_kernel_stats.txt "
                        "(rocprofv3 --kernel-trace --stats of this command); overlapped_kernel_ms = the same kernel with the "
                        "other slots' kernels sharing the chip, the 'timed' / 'instrum.' columns" % iso_steps,
            "overlapped_kernel_ms": round(ovl_ms, 4),
            "frac_overlapped": round(roof["frac"] * dom_ms / ovl_ms, 4) if ovl_ms else None,
            "workgroups": wgs[dom], "cu_share": round(cu_share[dom], 4),
            "frac_of_cu_share": round(roof["frac"] / cu_share[dom], 4),
            "cu_share_note": "the kernel is launched on this share of the 256 CUs (one persistent workgroup per CU) so that "
                             "the other batches in flight run beside it; achieved/peak above are against the WHOLE chip"})
        path_tf = value / world * FLOP_PER_CANDIDATE / 1e12
        out = {
            "metric": "candidate sites/sec (whole node)",
            "value": round(value, 1),
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (matmuls as 2-way fp16 split on MFMA with fp32 accumulate; gates/activations fp32)",
            "data": "synthetic",
            "config": {"workload": "%s weights-shape model (random init), synthetic %s-profile pileup tensors, "
                                   "batch=%d, %d batches in flight per GPU, inputs resident in HBM"
                                   % (PLATFORM[args.platform], args.platform, batch, streams),
                       "batch": batch, "streams": streams, "candidates_per_gpu": args.steps * batch},
            "roofline": roof,
            "roofline_path": {"achieved": round(path_tf, 2), "unit": "TFLOP/s (algorithmic fp32-equivalent, 40 386 432 FLOP / candidate)",
                              "peak_fp32_mfma": PEAK_FP32_MFMA_TFLOPS, "frac_of_fp32_mfma": round(path_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                              "peak_f16_split": round(PEAK_F16_MFMA_TFLOPS / SPLIT_TERMS, 1),
                              "frac_of_f16_split": round(path_tf * SPLIT_TERMS / PEAK_F16_MFMA_TFLOPS, 4),
                              "hbm_gbs": round(value / world * sum(KERNEL_BYTES.values()) / 1e9, 1),
                              "hbm_frac": round(value / world * sum(KERNEL_BYTES.values()) / 1e9 / PEAK_HBM_GBS, 4)},
            "value_with_kernel_events": round(args.steps * batch / elapsed_events, 1),   # this rank, instrumented repeat of the timed loop
            "kernels": kern,
            "kernels_single_stream_ms": kern_iso,
            "parity_max_abs_err": parity,
            "gt_concordance": concord,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, x, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    eng.dataset_free(xd, od)
    eng.close()
    group.close()


if __name__ == "__main__":
    main()
