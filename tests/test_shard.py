"""Multi-GPU path on CPU: candidate sharding + the (trivial) collectives, world_size 2 over gloo.

Each rank runs the forward pass of ITS shard (the oracle stands in for the GPU engine here: no GPU in
this container), the shards are gathered in rank order and must equal the single-process result
bit for bit -- candidates are independent, so sharding must not change anything."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from clair_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,batch,world", [(0, 4, 2), (1, 4, 2), (10, 4, 2), (16, 4, 2), (1000, 100, 8), (5, 1024, 8), (200000, 1024, 8)])
def test_shard_batches_partition(n, batch, world):
    covered = 0
    for r in range(world):
        first, cnt = shard.shard_batches(n, batch, r, world)
        assert cnt >= 0
        if cnt:
            assert first == covered          # contiguous, in rank order
            assert first % batch == 0        # whole batches
            covered += cnt
    assert covered == n
    sizes = [(shard.shard_batches(n, batch, r, world)[1] + batch - 1) // batch for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, %(root)r)
    from clair_amd import shard, synth, weights
    from oracle import c_oracle
    g = shard.NodeGroup(backend="gloo")
    w = weights.synthetic_weights(seed=5)
    x, _ = synth.synthetic_input(37, "ont", seed=9)
    first, cnt = shard.shard_batches(len(x), 8, g.rank, g.world)
    outs = c_oracle.forward(w, x[first:first + cnt], threads=1)
    packed = np.concatenate(outs, axis=1)
    g.barrier()
    full = g.gather_arrays(packed)
    total = g.sum_int(cnt)
    slowest = g.max_float(1.0 + g.rank)
    if g.rank == 0:
        np.save(%(out)r, full)
        print("RESULT", total, slowest, full.shape[0])
    g.close()
""")


def test_two_rank_gloo_shards_match_single_process(tmp_path):
    out = str(tmp_path / "gathered.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": out})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT")][0].split()
    assert int(line[1]) == 37 and float(line[2]) == 2.0 and int(line[3]) == 37
    from clair_amd import synth, weights
    from oracle import c_oracle
    w = weights.synthetic_weights(seed=5)
    x, _ = synth.synthetic_input(37, "ont", seed=9)
    want = np.concatenate(c_oracle.forward(w, x, threads=1), axis=1)
    got = np.load(out)
    assert got.shape == want.shape and np.array_equal(got, want)
