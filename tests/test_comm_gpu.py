"""RCCL communicator of the C ABI on the GPU box (one GPU there: a one-rank communicator exercises the whole binding --
librccl.so resolved at run time, ncclGetUniqueId / ncclCommInitRank, the staged collectives).  The N > 1 logic is covered on
CPU by tests/test_shard.py; 8-GPU runs are the driver's."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_collectives():
    from clair_amd import _capi
    lib = _capi.load()
    uid = (ctypes.c_uint8 * 128)()
    assert lib.clair_comm_unique_id(uid) == 0, lib.clair_comm_last_error(None)
    assert any(bytes(uid))
    h = ctypes.c_void_p()
    assert lib.clair_comm_create(0, 0, 1, uid, ctypes.byref(h)) == 0, lib.clair_comm_last_error(None)
    try:
        assert lib.clair_comm_barrier(h) == 0
        v = np.array([1.5, -2.0, 3.25], dtype=np.float64)
        for op in (0, 1, 2):
            a = v.copy()
            assert lib.clair_comm_allreduce_f64(h, ctypes.c_void_p(a.ctypes.data), a.size, op) == 0
            assert np.array_equal(a, v)
        blob = np.arange(2377818, dtype=np.float32)          # the weight blob's size
        b = blob.copy()
        assert lib.clair_comm_broadcast(h, ctypes.c_void_p(b.ctypes.data), b.nbytes, 0) == 0
        assert np.array_equal(b, blob)
        send = np.random.default_rng(1).random((1000, 90)).astype(np.float32)
        recv = np.zeros((1,) + send.shape, dtype=np.float32)
        assert lib.clair_comm_allgather(h, ctypes.c_void_p(send.ctypes.data), ctypes.c_void_p(recv.ctypes.data), send.nbytes) == 0
        assert np.array_equal(recv[0], send)
        assert lib.clair_comm_allreduce_f64(h, ctypes.c_void_p(v.ctypes.data), v.size, 7) != 0
        assert b"unknown reduction" in lib.clair_comm_last_error(h)
    finally:
        lib.clair_comm_destroy(h)


def test_bench_under_torch_distributed_run_one_rank_and_own_spawner(tmp_path):
    """The driver's launch line with --nproc-per-node 1, and `python bench.py --gpus 1` without a launcher: one JSON line each,
    n_gpus 1, per-rank table present."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_WARM_STEPS="16")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tail = ["bench.py", "--gpus", "1", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--gt-candidates", "0", "--sustained-seconds", "0.2"]
    for cmd in ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", str(port)] + tail, [sys.executable] + tail):
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["steps"] == 12 and len(d["per_rank"]) == 1 and d["per_rank"][0]["steps"] == 12
        assert d["roofline"]["bound"] == "mfma" and d["roofline"]["kernel"].split()[0] in ("proj2", "l4", "lstm1", "lstm2")
