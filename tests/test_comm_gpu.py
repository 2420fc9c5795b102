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


def test_bench_two_ranks_share_the_one_gpu_over_the_socket_transport():
    """The N > 1 flow of bench.py with REAL engines (round 5): two ranks, both on device 0 (BENCH_SHARE_DEVICE=1), the barrier, the max over
    ranks, the weight blob and the per-rank records over the socket transport; every rank bound to its share of the cores next to the GPU
    (CLAIR_AMD_BIND=1: the same NUMA node, split in two).  A test of the plumbing, not a measurement: no rank holds an RCCL communicator, so
    the line says n_gpus 0 and rank 0 exits non-zero -- a multi-GPU number is only ever reported over RCCL."""
    import json
    import os
    import sys
    from clair_amd import shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_WARM_STEPS="16", BENCH_SHARE_DEVICE="1", CLAIR_AMD_BIND="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "4", "--no-cpu-baseline", "--gt-candidates", "0",
           "--sustained-seconds", "0.3", "--full-candidates", "8192"]
    procs = shard.spawn_ranks(cmd, 2, env=env, stderr_pipe=True)
    out, errs, rcs = _collect(procs, 600)
    assert rcs == [1, 0], errs
    assert "0 of 2 ranks hold an RCCL communicator" in errs[0]
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 0 and d["config"]["transport"] == "tcp" and len(d["per_rank"]) == 2
    assert d["config"]["candidates_total"] == 2 * 40 * 1024 and d["value"] > 1e6 and d["value_sustained"] > 1e6 and d["parity_max_abs_err"] < 1e-5
    aff = [r["affinity"] for r in d["per_rank"]]
    assert aff[0]["pci"] == aff[1]["pci"] and aff[0]["pci"] and aff[0]["numa_node"] == aff[1]["numa_node"]
    if aff[0]["cpus_bound"]:          # the kernel knows the GPU's node: the two ranks took disjoint halves of its cores
        a, b = (set(shard.parse_cpulist(x["cpus_bound"])) for x in aff)
        assert a and b and not (a & b)
    for r in d["per_rank"]:           # each rank sampled the GPU's clock and power during its legs
        assert set(r["gpu_state"]["value_sustained"]) == {"sclk_mhz", "power_w", "samples"}
    assert d["boundary"]["bit_identical_to_resident"] is True


def _collect(procs, timeout):
    """stdout of rank 0 and every rank's stderr, read CONCURRENTLY (a rank that fills one pipe while the test waits on another would
    block for ever: HIP / RCCL / AMD_LOG_LEVEL output easily exceeds the 64 KB a pipe holds), then the exit codes."""
    import threading
    bufs = {}

    def drain(key, pipe):
        bufs[key] = pipe.read().decode(errors="replace")

    threads = [threading.Thread(target=drain, args=(("err", r), p.stderr), daemon=True) for r, p in enumerate(procs) if p.stderr is not None]
    threads.append(threading.Thread(target=drain, args=(("out", 0), procs[0].stdout), daemon=True))
    for t in threads:
        t.start()
    rcs = [p.wait(timeout=timeout) for p in procs]
    for t in threads:
        t.join(timeout=30)
    return bufs.get(("out", 0), ""), [bufs.get(("err", r), "") for r in range(len(procs))], rcs


def _fake_rccl(tmp_path):
    import os
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl.c")
    so = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", src, "-o", so])
    return so


def test_comm_create_timed_gives_up_on_an_rccl_that_never_returns(tmp_path):
    """clair_comm_create_timed against a librccl whose ncclCommInitRank sleeps for ever (tests/fake_rccl.c, CLAIR_AMD_RCCL_LIBRARY):
    CLAIR_COMM_TIMED_OUT at the deadline, no communicator, an error text that says so -- and when RCCL returns AFTER the deadline the
    abandoned helper thread aborts the communicator it got (ncclCommAbort, marked by the stand-in)."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mark = str(tmp_path / "abort_mark")
    code = textwrap.dedent("""
        import ctypes, sys, time
        sys.path.insert(0, %r)
        from clair_amd import _capi
        lib = _capi.load()
        uid = (ctypes.c_uint8 * 128)()
        assert lib.clair_comm_unique_id(uid) == 0 and bytes(uid)[:4] == b"\\x07" * 4        # the stand-in's id: it IS the library in use
        h = ctypes.c_void_p()
        t0 = time.time()
        rc = lib.clair_comm_create_timed(0, 0, 2, uid, 1500, ctypes.byref(h))
        dt = time.time() - t0
        print("RC", rc, bool(h.value), round(dt, 2), lib.clair_comm_last_error(None).decode())
        time.sleep(float(sys.argv[1]))
    """ % root)
    env = dict(os.environ, CLAIR_AMD_RCCL_LIBRARY=_fake_rccl(tmp_path), CLAIR_FAKE_RCCL_HANG="all", CLAIR_FAKE_RCCL_ABORT_MARK=mark)
    r = subprocess.run([sys.executable, "-c", code, "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    f = [ln for ln in r.stdout.splitlines() if ln.startswith("RC ")][0].split(" ", 4)
    assert f[1] == "2" and f[2] == "False" and 1.4 <= float(f[3]) < 5.0 and "did not return within 1.5 s" in f[4] and "rank 0 of 2" in f[4]
    assert not os.path.exists(mark)
    # RCCL returns 3 s after the call, 1.5 s after the deadline: nobody waits for it any more, the helper thread aborts what it got
    r = subprocess.run([sys.executable, "-c", code, "4"], env=dict(env, CLAIR_FAKE_RCCL_LATE="3"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("RC ")][0].split()[1] == "2"
    assert open(mark).read() == "aborted\n"


def test_bench_prints_its_two_rank_line_although_the_rccl_bring_up_hangs(tmp_path):
    """VERDICT r05 item 4 on the GPU box, with real engines: two ranks (both on device 0) ask for the RCCL transport, the library's
    ncclCommInitRank never returns on either rank.  Each rank gives up at CLAIR_AMD_RCCL_INIT_TIMEOUT, tells the other over the bootstrap
    sockets, both go on over the socket transport; rank 0 prints the ONE JSON line with `rccl_failure`: "... init timed out on rank ...",
    n_gpus 0, and the ranks leave (os._exit: a helper thread is still inside the library) with the exit codes of any run without RCCL."""
    import json
    import os
    import sys
    import time
    from clair_amd import shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_WARM_STEPS="16", BENCH_SHARE_DEVICE="1", BENCH_SHARE_TRANSPORT="rccl", CLAIR_AMD_RCCL_LIBRARY=_fake_rccl(tmp_path),
               CLAIR_FAKE_RCCL_HANG="all", CLAIR_AMD_RCCL_INIT_TIMEOUT="3")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4", "--no-cpu-baseline", "--gt-candidates", "0",
           "--sustained-seconds", "0.2", "--full-candidates", "4096", "--boundary-slots", "0"]
    t0 = time.time()
    procs = shard.spawn_ranks(cmd, 2, env=env, stderr_pipe=True)
    out, errs, rcs = _collect(procs, 600)
    assert rcs == [1, 0], errs
    assert time.time() - t0 < 300
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 0 and d["config"]["transport"] == "tcp" and d["config"]["ranks_with_rccl_communicator"] == 0
    why = d["config"]["rccl_failure"]
    assert "clair_comm_create failed on 2 of 2 ranks" in why and "init timed out on rank 0 after 3 s" in why and "init timed out on rank 1 after 3 s" in why
    assert "RCCL start-up failed" in errs[0] and "0 of 2 ranks hold an RCCL communicator" in errs[0]
    assert len(d["per_rank"]) == 2 and d["config"]["candidates_total"] == 2 * 20 * 1024 and d["value"] > 1e6 and d["parity_max_abs_err"] < 1e-5
