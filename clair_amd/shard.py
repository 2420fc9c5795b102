"""Candidate sharding across the GPUs of one node (one process per GPU).

The reference scales out by printing one ``callVarBam`` shell command per 10 Mbp chunk and
letting GNU parallel run them (/root/reference/clair/callVarBamParallel.py:90-119,
README.md:297); chunk VCFs are concatenated in order afterwards (README.md:303).  Candidates are
classified independently (docs/POST_PROCESSING.md:17), so here rank r simply owns a contiguous
block of whole batches of the candidate stream -- per-rank output fragments concatenate in
input order -- and there is no data-path collective.

What the ranks do exchange -- the weight blob from rank 0 (9.5 MB, once), per-rank output rows,
counters and timers -- goes through ``NodeGroup``:

* transport ``"rccl"``: the C ABI's communicator (include/clair_amd.h: clair_comm_*, a direct
  binding of librccl.so; xGMI between the GPUs of the node).  One GPU per rank is required.
* transport ``"tcp"``: the same operations over the bootstrap sockets on 127.0.0.1 (pure
  Python).  It is what the CPU tests run, and it carries the RCCL unique id during start-up.

Rendezvous (one node): rank 0 listens on an ephemeral port of 127.0.0.1 and publishes it in a
file every rank can name: ``$CLAIR_AMD_RDZV`` when set (bench.py's own spawner sets it), else
``/tmp/clair_amd_rdzv_<MASTER_PORT>_<parent pid>`` -- under ``torch.distributed.run`` all ranks
share the launcher as parent and MASTER_PORT itself is taken by the launcher's store.
No torch anywhere in this module.
"""
import os
import pickle
import socket
import struct
import tempfile
import time

import numpy as np


def shard_batches(n_candidates, batch, rank, world):
    """Contiguous block of whole batches for `rank`: returns (first_candidate, n_candidates_of_rank).

    Batches are dealt so that ranks differ by at most one batch; the ragged last batch stays
    with the last rank that has any work, which keeps every rank's range contiguous."""
    if n_candidates <= 0:
        return 0, 0
    nb = (n_candidates + batch - 1) // batch
    base, extra = divmod(nb, world)
    my_batches = base + (1 if rank < extra else 0)
    first_batch = rank * base + min(rank, extra)
    first = first_batch * batch
    last = min(n_candidates, (first_batch + my_batches) * batch)
    return first, max(0, last - first)


def rendezvous_path():
    explicit = os.environ.get("CLAIR_AMD_RDZV")
    if explicit:
        return explicit
    return os.path.join(tempfile.gettempdir(), "clair_amd_rdzv_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))


def _send_msg(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("peer closed the bootstrap connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


class _TcpStar(object):
    """Rank 0 <-> every other rank over 127.0.0.1: the bootstrap channel and the CPU transport.
    One primitive -- every rank contributes an object, every rank gets the list in rank order."""

    def __init__(self, rank, world, path, timeout):
        self.rank, self.world = rank, world
        self.peers = []          # rank 0: sockets indexed by rank - 1
        self.root = None         # other ranks: socket to rank 0
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(world)
            srv.settimeout(timeout)
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "w") as f:
                f.write("%d\n" % srv.getsockname()[1])
            os.replace(tmp, path)          # atomic: a reader sees no file or the whole port
            self._path = path
            socks = {}
            try:
                while len(socks) < world - 1:
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(timeout)
                    r = _recv_msg(conn)
                    if not isinstance(r, int) or not 0 < r < world or r in socks:
                        raise RuntimeError("bootstrap: unexpected rank announcement %r" % (r,))
                    socks[r] = conn
            except socket.timeout:
                raise RuntimeError("bootstrap: only %d of %d ranks joined within %.0f s" % (len(socks) + 1, world, timeout))
            finally:
                srv.close()
                try:
                    os.unlink(path)
                except OSError:
                    pass
            self.peers = [socks[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            port = None
            while port is None:
                try:
                    with open(path) as f:
                        port = int(f.read().strip())
                except (OSError, ValueError):
                    if time.time() > deadline:
                        raise RuntimeError("bootstrap: rank 0 never published %s" % path)
                    time.sleep(0.01)
            s = socket.create_connection(("127.0.0.1", port), timeout=timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send_msg(s, rank)
            self.root = s

    def allgather(self, obj):
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [_recv_msg(p) for p in self.peers]
            for p in self.peers:
                _send_msg(p, items)
            return items
        _send_msg(self.root, obj)
        return _recv_msg(self.root)

    def close(self):
        for s in self.peers + ([self.root] if self.root else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.root = [], None


class NodeGroup(object):
    """The ranks of one node, from the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE, as
    set by torch.distributed.run or by bench.py's own spawner).  With WORLD_SIZE == 1 nothing is
    opened.  transport: "rccl" | "tcp" | None (= "rccl" when the HIP library sees a device, else "tcp")."""

    def __init__(self, transport=None, timeout=180.0, rank=None, world=None, local_rank=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank))) if local_rank is None else int(local_rank)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside world of %d" % (self.rank, self.world))
        self.transport = "none"
        self._lib = None
        self._comm = None
        self._star = _TcpStar(self.rank, self.world, rendezvous_path(), timeout)
        if self.world == 1:
            return
        if transport is None:
            from clair_amd import _capi
            transport = "rccl" if _capi.load().clair_device_count() > 0 else "tcp"
        if transport not in ("rccl", "tcp"):
            raise ValueError("unknown transport %r" % (transport,))
        self.transport = transport
        if transport == "rccl":
            import ctypes
            from clair_amd import _capi
            lib = _capi.load()
            uid = (ctypes.c_uint8 * 128)()
            if self.rank == 0 and lib.clair_comm_unique_id(uid) != 0:
                err = lib.clair_comm_last_error(None).decode()
                self._star.allgather(("error", err))
                raise _capi.EngineError("clair_comm_unique_id failed: %s" % err)
            got = self._star.allgather(("id", bytes(uid)) if self.rank == 0 else None)[0]
            if got[0] != "id":
                raise _capi.EngineError("rank 0 could not create the RCCL unique id: %s" % got[1])
            uid = (ctypes.c_uint8 * 128).from_buffer_copy(got[1])
            h = ctypes.c_void_p()
            if lib.clair_comm_create(self.local_rank, self.rank, self.world, uid, ctypes.byref(h)) != 0:
                raise _capi.EngineError("clair_comm_create failed on rank %d: %s" % (self.rank, lib.clair_comm_last_error(None).decode()))
            self._lib, self._comm = lib, h

    # -- helpers ---------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            from clair_amd import _capi
            raise _capi.EngineError("%s failed on rank %d: %s" % (what, self.rank, self._lib.clair_comm_last_error(self._comm).decode()))

    def _reduce_f64(self, values, op):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        if self.world == 1:
            return a
        if self._comm is not None:
            import ctypes
            self._check(self._lib.clair_comm_allreduce_f64(self._comm, ctypes.c_void_p(a.ctypes.data), a.size, {"sum": 0, "max": 1, "min": 2}[op]),
                        "clair_comm_allreduce_f64")
            return a
        parts = np.stack(self._star.allgather(a))
        return {"sum": parts.sum, "max": parts.max, "min": parts.min}[op](axis=0)

    # -- operations -------------------------------------------------------------------------------
    def barrier(self):
        if self.world == 1:
            return
        if self._comm is not None:
            self._check(self._lib.clair_comm_barrier(self._comm), "clair_comm_barrier")
        else:
            self._star.allgather(None)

    def max_float(self, value):
        return float(self._reduce_f64([float(value)], "max")[0])

    def sum_int(self, value):
        v = int(value)
        if abs(v) >= 2 ** 53:
            raise OverflowError("sum_int carries its operands as float64")
        return int(round(float(self._reduce_f64([float(v)], "sum")[0])))

    def gather_floats(self, value):
        """One float per rank, in rank order, on every rank."""
        v = np.zeros(self.world, dtype=np.float64)
        v[self.rank] = float(value)
        return [float(t) for t in self._reduce_f64(v, "sum")]

    def broadcast_array(self, array, root=0):
        """`array` (same shape and dtype on every rank; contents matter on `root` only) filled from root's copy."""
        a = np.ascontiguousarray(array)
        if self.world == 1:
            return a
        if self._comm is not None:
            import ctypes
            a = a.copy()
            self._check(self._lib.clair_comm_broadcast(self._comm, ctypes.c_void_p(a.ctypes.data), a.nbytes, int(root)), "clair_comm_broadcast")
            return a
        return np.ascontiguousarray(self._star.allgather(a if self.rank == root else None)[root])

    def broadcast_weights(self, w, root=0):
        """The weight dictionary of clair_amd.weights (9.5 MB as one float32 blob) from `root` to every rank."""
        from clair_amd.weights import TENSOR_TABLE
        sizes = [int(np.prod(s)) for s in TENSOR_TABLE.values()]
        blob = np.zeros(sum(sizes), dtype=np.float32)
        if self.rank == root:
            blob = np.concatenate([np.ascontiguousarray(w[k], dtype=np.float32).ravel() for k in TENSOR_TABLE])
        blob = self.broadcast_array(blob, root)
        out, at = {}, 0
        for (k, shape), n in zip(TENSOR_TABLE.items(), sizes):
            out[k] = blob[at:at + n].reshape(shape).copy()
            at += n
        return out

    def gather_arrays(self, array):
        """All ranks' float32 arrays (first dims may differ) concatenated in rank order, on every rank."""
        a = np.ascontiguousarray(array, dtype=np.float32)
        if self.world == 1:
            return a
        counts = [int(round(c)) for c in self.gather_floats(a.shape[0])]
        width = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
        if self._comm is not None:
            import ctypes
            pad = np.zeros((max(max(counts), 1), width), dtype=np.float32)
            pad[:a.shape[0]] = a.reshape(a.shape[0], width)
            recv = np.empty((self.world,) + pad.shape, dtype=np.float32)
            self._check(self._lib.clair_comm_allgather(self._comm, ctypes.c_void_p(pad.ctypes.data), ctypes.c_void_p(recv.ctypes.data), pad.nbytes),
                        "clair_comm_allgather")
            parts = [recv[r, :c] for r, c in enumerate(counts)]
        else:
            parts = [p.reshape(p.shape[0], width) for p in self._star.allgather(a)]
        out = np.concatenate(parts, axis=0)
        return out.reshape((out.shape[0],) + a.shape[1:])

    def close(self):
        if self.world > 1:
            try:
                self.barrier()
            except Exception:
                pass
        if self._comm is not None:
            self._lib.clair_comm_destroy(self._comm)
            self._comm = None
        self._star.close()


def spawn_ranks(argv, world, env=None, rdzv_dir=None):
    """Start `world` copies of `argv` (one per GPU: RANK = LOCAL_RANK = 0..world-1) the way torch.distributed.run would,
    with a private rendezvous file.  Returns the list of Popen objects (stdout of rank 0 is a pipe, the others inherit)."""
    import subprocess
    base = dict(os.environ if env is None else env)
    d = rdzv_dir or tempfile.mkdtemp(prefix="clair_amd_rdzv_")
    procs = []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                 CLAIR_AMD_RDZV=os.path.join(d, "port"))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else None))
    return procs
