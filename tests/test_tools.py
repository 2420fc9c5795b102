"""CPU checks of the GPU-box tooling whose code paths only run when something goes wrong there."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


class _FakeEngine:
    """Stands in for clair_amd._capi.Engine: the oracle's probabilities, with a transient fault on chosen calls."""

    def __init__(self, w, fault_calls, delta=5e-5):
        from oracle import c_oracle
        self.w, self.oracle, self.calls, self.fault_calls, self.delta = w, c_oracle, 0, set(fault_calls), delta
        self.last = None

    def predict(self, x):
        outs, inter = self.oracle.forward(self.w, x, keep_intermediates=True)
        if self.calls in self.fault_calls:
            outs[0][3, 5] += self.delta
            outs[0][3, 6] -= self.delta
        self.calls += 1
        self.last = inter
        return outs

    def debug_read(self, slot, which, shape):
        a = self.last["a1" if which == 1 else "a2"].transpose(1, 0, 2)
        out = np.zeros(shape, np.float32)
        out[:, :a.shape[1]] = a
        return out

    def kernel_workgroups(self, n):
        return {}


def test_concordance_dissects_an_excursion(synth_weights, tmp_path):
    """tools/gt_concordance.py on a chunk beyond the tolerance: names the candidates, re-runs the (fake) HIP side, re-evaluates the
    oracle single-threaded, classifies the event and writes the evidence file."""
    import gt_concordance
    lines = []
    eng = _FakeEngine(synth_weights, fault_calls=[1])       # second batch of the first chunk, once
    r = gt_concordance.concordance(eng, synth_weights, "ont", 160, 777, batch=32, chunk=96, log=lines.append, keep_taps=True, dump_dir=str(tmp_path))
    assert len(r["excursions"]) == 1
    ex = r["excursions"][0]
    assert ex["chunk_start"] == 0 and ex["candidates_beyond_tol"] == 1 and 4e-5 < ex["worst"] < 6e-5
    assert "transient" in ex["verdict"]
    text = "\n".join(lines)
    assert "(1, 0, 3)" in text          # batch 1, tile 0, lane 3
    dump = np.load(os.path.join(str(tmp_path), "excursion_ont_0.npz"))
    assert dump["index"][0] == 35 and dump["x"].shape[1:] == (33, 8, 4) and "a2_first_0" in dump.files
    # a fault that reproduces on every pass is classified as deterministic
    eng = _FakeEngine(synth_weights, fault_calls=range(1, 100, 1))
    eng.fault_calls.discard(0)
    lines.clear()
    r = gt_concordance.concordance(eng, synth_weights, "ont", 32, 777, batch=32, chunk=32, log=lines.append, dump_dir=None)
    assert r["excursions"] == []        # call 0 is clean
    box = gt_concordance.box_info(full=False)
    assert "host" in box and isinstance(box["unique_ids"], list)
    assert isinstance(gt_concordance.gpu_state(), str)


def test_fast_reads_are_well_formed_alignments():
    """tools/fast_reads.py (inputs of the front-end benchmarks): CIGARs consume exactly their SEQ, stay inside the contig, start positions
    ascend, and the host stages accept them and find candidates at about the advertised density."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fast_reads
    from clair_amd import _hostapi
    case = fast_reads.make(ref_len=60000, depth=20, read_len=(500, 3000), seed=2, noisy_every=20)
    last = 0
    for line in case["sam"].splitlines():
        col = line.split(b"\t")
        ops = re.findall(rb"(\d+)([MID])", col[5])
        assert b"".join(n + o for n, o in ops) == col[5]
        q = sum(int(n) for n, o in ops if o in b"MI")
        r = sum(int(n) for n, o in ops if o in b"MD")
        pos = int(col[3])
        assert q == len(col[9]) == len(col[10]) and pos >= last and pos + r - 1 <= case["ref_len"] and ops[0][1] == b"M" and ops[-1][1] == b"M"
        last = pos
    f = _hostapi.CandidateFinder(case["ctg"], case["ref"], 0, min_coverage=4, threshold=0.125)
    assert f.feed(case["sam"]) == b""
    f.finish()
    n = len(f.take_positions())
    assert 60000 / 120 < n < 60000 / 15
