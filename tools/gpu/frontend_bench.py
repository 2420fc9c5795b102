#!/usr/bin/env python3
"""The device front end against the sequential host stages on the same alignments (run on the GPU box):

    python tools/gpu/frontend_bench.py [ref_len] [noisy_every] [depth] [read_len_min read_len_max]

Synthetic contig at 50x with 2-9 kb reads (tools/fast_reads.py, the inputs of tools/e2e_bam_bench.py; one candidate site per
~2 x noisy_every bases).  Prints the time of every
step of both paths -- host: candidate search (clair_host_evc_*), pileup (clair_host_pileup_*); device: packing (clair_host_sampack_*),
copy + tally (clair_frontend_add_reads), candidate filter, second pass + assembly -- checks that the two produce the same candidates and
the same windows bit for bit, and prints the per-kernel rates.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fast_reads  # noqa: E402
from clair_amd import _capi, _hostapi  # noqa: E402


def main():
    ref_len = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    noisy_every = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    read_len = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (2000, 9000)
    t0 = time.time()
    case = fast_reads.make(ref_len=ref_len, depth=depth, noisy_every=noisy_every, seed=5, read_len=read_len)
    case["ref0"] = 0
    sam = case["sam"]
    print("inputs: %.1f MB of SAM text, %d alignments over %d bases (%.0f s to generate)" % (len(sam) / 1e6, sam.count(b"\n"), ref_len, time.time() - t0))

    # ---- sequential host stages (one thread each, as callVarBam runs them) ----
    t0 = time.time()
    finder = _hostapi.CandidateFinder(case["ctg"], case["ref"], case["ref0"], min_coverage=4, threshold=0.125)
    finder.feed(sam)
    finder.finish()
    want_pos = finder.take_positions()
    t_evc = time.time() - t0
    t0 = time.time()
    b = _hostapi.PileupBuilder(case["ctg"], case["ref"], case["ref0"], want_pos)
    b.feed(sam)
    b.finish()
    hc, hs, hcounts = b.take_columns()
    t_pile = time.time() - t0
    slots_left = b.stats()["slots_left"] if "slots_left" in b.stats() else None
    print("host  : candidate search %.3f s (%.0f MB/s), pileup %.3f s (%.0f windows/s)  -> %d candidates, %d windows%s"
          % (t_evc, len(sam) / 1e6 / t_evc, t_pile, len(hc) / t_pile, len(want_pos), len(hc), "" if slots_left is None else ", %d slots left" % slots_left))

    # ---- device front end ----
    _capi.Frontend(0, "ACGT" * 64, 0, -64, 320).close()          # context creation is not the front end's time
    for rep in range(2):
        t0 = time.time()
        f = _capi.Frontend(0, case["ref"], case["ref0"], case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64)
        t_create = time.time() - t0
        p = _hostapi.SamPacker(case["ctg"])
        t_pack = t_add = 0.0
        step = 64 << 20
        tail = b""
        for at in range(0, len(sam), step):
            chunk = tail + sam[at:at + step]
            t0 = time.time()
            tail = p.feed(chunk, final=at + step >= len(sam))
            t_pack += time.time() - t0
            t0 = time.time()
            f.add_slab(p)
            t_add += time.time() - t0
        st = f.stats()
        t0 = time.time()
        n_cand = f.find_candidates(min_coverage=4, threshold=0.125)
        t_cand = time.time() - t0
        t0 = time.time()
        n_win = f.build_windows(min_coverage=0, drop_non_iupac_centre=False)
        t_win = time.time() - t0
        t0 = time.time()
        binds = f.budget_binds()
        t_budget = time.time() - t0
        total = t_pack + t_add + t_cand + t_win + t_budget
        print("device: tables %.3f s | pack %.3f s (%.0f MB/s) | copy + tally %.3f s | candidates %.3f s | windows %.3f s | budget replay %.3f s  = %.3f s"
              " -> %d candidates, %d windows, %d elements, anomalies %d, budget binds: %s"
              % (t_create, t_pack, len(sam) / 1e6 / t_pack, t_add, t_cand, t_win, t_budget, total, n_cand, n_win, st["elements"], f.stats()["anomalies"], binds))
        if rep == 0:
            got_pos = f.candidates()
            centres, seqs = f.window_info(0, n_win)
            counts = f.window_counts(0, n_win)
            same = (np.array_equal(got_pos, want_pos) and np.array_equal(centres, hc) and np.array_equal(seqs, hs) and np.array_equal(counts.astype(np.int32), hcounts))
            print("device == host: %s (candidates, centres, reference windows, %d counts)" % (same, counts.size))
            if not same and not binds:
                return 1
        f.close()
    print("speed-up of the two stages: %.1fx (host %.3f s, device %.3f s with packing)" % ((t_evc + t_pile) / total, t_evc + t_pile, total))

    # ---- the same with the text parsed on the device, out of a page-locked buffer (what callVarBam does with the pipe) ----
    engine = _capi.Engine(0, 64, 1)
    step = 256 << 20
    pinned = engine.pinned_buffer(min(len(sam), step) + 16)
    for rep in range(2):
        f = _capi.Frontend(0, case["ref"], case["ref0"], case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64)
        f.text_options(case["ctg"])
        t_copy = t_text = 0.0
        at = 0
        while at < len(sam):
            cut = len(sam) if at + step >= len(sam) else sam.rindex(b"\n", at, at + step) + 1
            t0 = time.time()
            pinned[:cut - at] = np.frombuffer(sam, np.uint8, cut - at, at)       # stands in for the pipe read
            t_copy += time.time() - t0
            t0 = time.time()
            f.add_text(pinned.ctypes.data, cut - at)
            t_text += time.time() - t0
            at = cut
        t0 = time.time()
        n_cand = f.find_candidates(min_coverage=4, threshold=0.125)
        n_win = f.build_windows(min_coverage=0, drop_non_iupac_centre=False)
        binds = f.budget_binds()
        t_rest = time.time() - t0
        print("device, text parsed on the device: copy + parse + tally %.3f s (%.0f MB/s of text) | candidates + windows + budget replay %.3f s  = %.3f s -> %d candidates, "
              "%d windows, anomalies %d, budget binds: %s   (filling the page-locked buffer, the pipe read's stand-in: %.3f s)"
              % (t_text, len(sam) / 1e6 / t_text, t_rest, t_text + t_rest, n_cand, n_win, f.stats()["anomalies"], binds, t_copy))
        if rep == 0:
            centres, seqs = f.window_info(0, n_win)
            same = (np.array_equal(f.candidates(), want_pos) and np.array_equal(centres, hc) and np.array_equal(seqs, hs)
                    and np.array_equal(f.window_counts(0, n_win).astype(np.int32), hcounts))
            print("device (text) == host: %s" % same)
            if not same and not binds:
                return 1
        f.close()
    print("speed-up of the two stages with the text parsed on the device: %.1fx (host %.3f s, device %.3f s)" % ((t_evc + t_pile) / (t_text + t_rest), t_evc + t_pile, t_text + t_rest))
    engine.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
