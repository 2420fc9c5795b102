#!/usr/bin/env python3
"""Long differential fuzz of the device front end (run on the GPU box):

    python tools/gpu/frontend_fuzz.py [n_cases] [first_seed]

Every case is a fresh random set of alignments with random options of both pileup stages (tests/frontend_cases.py: fuzz_case -- read
lengths, substitution / insertion / deletion rates, N operations, start-position bursts, --dcov, mapping-quality floors, depth floors,
allele-frequency thresholds, regions, bed intervals, left-edge windows on or off).  Expected values: the sequential host stages (clair_host_evc_*, clair_host_pileup_*:
pinned byte for byte against records minted from the reference's own scripts).  Each case goes through the device front end twice --
text packed on the host, text parsed on the device -- in a random number of slabs / chunks; candidates, window centres, reference
windows and every count are compared, and the budget replay and the CLAIR_FE_* reports must stay silent.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import frontend_cases as fc  # noqa: E402
from clair_amd import _capi, _hostapi  # noqa: E402


def cuts(sam, k):
    at = 0
    for i in range(k):
        cut = len(sam) if i == k - 1 else sam.index(b"\n", len(sam) * (i + 1) // k) + 1
        yield sam[at:cut], i == k - 1
        at = cut


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    t0 = time.time()
    windows = counts = cands = reads = bad = with_region = with_bed = 0
    for seed in range(first, first + n_cases):
        case, pile_kw, evc_kw, region = fc.fuzz_case(seed)
        rng = dict(ctg_start=region[0], ctg_end=region[1]) if region else {}
        want_pos = fc.host_candidates(case, **rng, **evc_kw)
        left_edge = seed % 3 != 0
        hc, hs, hcounts = fc.host_windows(case, candidates=want_pos, pile_region=region, dcov=pile_kw["dcov"], min_mq=pile_kw["min_mq"],
                                          min_coverage=pile_kw["min_coverage"], consider_left_edge=left_edge)
        pack_kw = dict(dcov=pile_kw["dcov"], pile_min_mq=pile_kw["min_mq"], evc_min_mq=evc_kw["min_mq"], pile_region=region)
        span = (case["ref0"] - 64, case["ref0"] + len(case["ref"]) + 64)
        for path in ("host-packed", "device-parsed"):
            f = _capi.Frontend(0, case["ref"], case["ref0"], *span)
            k = 1 + (seed + (path == "device-parsed")) % 5
            if path == "host-packed":
                p = _hostapi.SamPacker(case["ctg"], **pack_kw)
                for piece, last in cuts(case["sam"], k):
                    p.feed(piece, final=last)
                    f.add_slab(p)
                anomalies = p.stats()["anomalies"]
            else:
                f.text_options(case["ctg"], **pack_kw)
                for piece, _ in cuts(case["sam"], k):
                    f.add_text(piece)
                anomalies = 0
            n = f.find_candidates(min_coverage=evc_kw["min_coverage"], threshold=evc_kw["threshold"], ctg_start=rng.get("ctg_start"), ctg_end=rng.get("ctg_end"),
                                  bed=evc_kw["bed"])
            ok = n == len(want_pos) and np.array_equal(f.candidates(), want_pos)
            nw = f.build_windows(min_coverage=pile_kw["min_coverage"], drop_non_iupac_centre=False, consider_left_edge=left_edge)
            centres, seqs = f.window_info(0, nw)
            got = f.window_counts(0, nw).astype(np.int32)
            ok = ok and np.array_equal(hc, centres) and np.array_equal(hs, seqs) and np.array_equal(hcounts, got)
            ok = ok and (anomalies | f.stats()["anomalies"]) == 0 and not f.budget_binds()
            reads += f.stats()["reads"]
            f.close()
            if not ok:
                bad += 1
                print("MISMATCH: seed %d, %s, options %r %r %r" % (seed, path, pile_kw, evc_kw, region))
        windows += len(hc)
        counts += hcounts.size
        cands += len(want_pos)
        with_region += region is not None
        with_bed += evc_kw["bed"] is not None
    print("%d cases (%d with a region, %d with bed intervals) x 2 packing paths in %.0f s: %d alignments put on the device, %d candidate sites, %d windows, "
          "%d counts compared -> %d mismatches" % (n_cases, with_region, with_bed, time.time() - t0, reads, cands, windows, 2 * counts, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
