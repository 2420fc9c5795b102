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
  Python).  It is what the CPU tests run, it carries the RCCL unique id during start-up, and it is what every rank falls back to
  (loudly: stderr, ``rccl_failure``) when RCCL's own start-up fails on any of them.

Rendezvous (one node): rank 0 listens on an ephemeral port of 127.0.0.1 and publishes it in a
file every rank can name: ``$CLAIR_AMD_RDZV`` when set (bench.py's own spawner sets it), else
``/tmp/clair_amd_rdzv_<MASTER_PORT>_<parent pid>`` -- under ``torch.distributed.run`` all ranks
share the launcher as parent and MASTER_PORT itself is taken by the launcher's store.
No torch anywhere in this module.
"""
import os
import socket
import struct
import sys
import tempfile
import threading
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


# -- CPU placement of a rank ------------------------------------------------------------------------------------------------------------
# The reference pins its pipeline stages to cores with taskset (clair/callVarBam.py:103-115).  Here a rank (bench.py --gpus N, a
# callVarBamParallel --run worker) feeds ONE GPU through page-locked staging buffers; on a two-socket node the copies of a rank that runs
# on the far socket cross the inter-socket link twice (pageable -> staging, staging -> PCIe root of the GPU).  So each rank binds itself to
# the host cores next to ITS GPU: the device's PCI address (clair_device_pci_bus_id) names /sys/bus/pci/devices/<bdf>/{numa_node,
# local_cpulist}; ranks whose GPUs share a node split that node's cores among themselves (contiguous slices, in rank order).  Nothing is
# bound at world size 1 (the single-GPU runs behave as before) unless CLAIR_AMD_BIND=1; CLAIR_AMD_BIND=0 turns it off everywhere.
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format); anything malformed -> []."""
    cpus = []
    try:
        for part in text.strip().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            lo, hi = int(a), int(b or a)
            if hi < lo or hi - lo > 1 << 16:
                return []
            cpus.extend(range(lo, hi + 1))
    except ValueError:
        return []
    return sorted(set(cpus))


def gpu_locality(bdf, sysfs_root="/sys"):
    """What the kernel says about the PCI device `bdf` ("0000:c1:00.0"): {"pci", "numa_node" (None when unknown / -1), "cpus"}."""
    out = {"pci": bdf or None, "numa_node": None, "cpus": []}
    if not bdf:
        return out
    base = os.path.join(sysfs_root, "bus", "pci", "devices", bdf)
    try:
        node = int(open(os.path.join(base, "numa_node")).read().strip())
        out["numa_node"] = node if node >= 0 else None
    except (OSError, ValueError):
        pass
    try:
        out["cpus"] = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
    except OSError:
        pass
    if not out["cpus"] and out["numa_node"] is not None:      # some kernels leave local_cpulist empty for devices behind a switch
        try:
            out["cpus"] = parse_cpulist(open(os.path.join(sysfs_root, "devices", "system", "node", "node%d" % out["numa_node"], "cpulist")).read())
        except OSError:
            pass
    return out


def physical_cores(cpus, sysfs_root="/sys"):
    """`cpus` grouped by physical core, in the order of each core's lowest hardware thread: the kernel lists a node's cores as e.g.
    "0-15,64-79" where 64-79 are the SMT siblings of 0-15 (devices/system/cpu/cpuN/topology/thread_siblings_list).  Without that file a
    hardware thread is its own core."""
    cpus = sorted(set(int(c) for c in cpus))
    have, seen, groups = set(cpus), set(), []
    for c in cpus:
        if c in seen:
            continue
        try:
            sib = parse_cpulist(open(os.path.join(sysfs_root, "devices", "system", "cpu", "cpu%d" % c, "topology", "thread_siblings_list")).read())
        except OSError:
            sib = []
        grp = sorted(t for t in set(sib) | {c} if t in have and t not in seen)
        seen.update(grp)
        groups.append(grp)
    return groups


def plan_affinity(localities, allowed, sysfs_root="/sys"):
    """Per rank, the cores it should run on: its GPU's local cores that this process may use (`allowed`: the cgroup's / taskset's set),
    split evenly among the ranks that share the same set -- in units of PHYSICAL cores (contiguous runs of cores in rank order, every
    core with all its hardware threads: two ranks never share a core's SMT siblings; a slice is never empty while there are cores).
    None for a rank whose GPU has no known local cores inside `allowed`: that rank stays where the launcher put it."""
    allowed = set(allowed)
    usable = [tuple(c for c in loc.get("cpus", []) if c in allowed) for loc in localities]
    plan = [None] * len(localities)
    for cpus in set(u for u in usable if u):
        sharers = [r for r, u in enumerate(usable) if u == cpus]
        k = len(sharers)
        cores = physical_cores(cpus, sysfs_root)
        for j, r in enumerate(sharers):
            if len(cores) >= k:
                lo, hi = j * len(cores) // k, (j + 1) * len(cores) // k
                plan[r] = sorted(t for core in cores[lo:hi] for t in core)
            else:                        # more ranks than cores on this node: share all of them
                plan[r] = list(cpus)
    return plan


def set_affinity_of_process(cpus):
    """sched_setaffinity for EVERY thread of this process (/proc/self/task), not just the caller: by the time a rank binds itself the HIP
    runtime, the library's staging threads and NumPy's pool may exist, and a thread keeps the mask it was created with.  Threads that
    vanish meanwhile are skipped; the calling thread's result decides (raises OSError when refused)."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = []
    me = threading.get_native_id() if hasattr(threading, "get_native_id") else 0
    for tid in tids:
        if tid == me:
            continue
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:
            pass
    os.sched_setaffinity(0, cpus)
    return len(tids)


def local_pci_bus_id(device):
    """PCI address of HIP device `device` through the C ABI, or "" (no library, no device)."""
    try:
        import ctypes
        from clair_amd import _capi
        lib = _capi.load()
        buf = ctypes.create_string_buffer(64)
        if lib.clair_device_pci_bus_id(int(device), buf, 64) == 0:
            return buf.value.decode()
    except Exception:      # noqa: BLE001 -- placement is best effort; the engine itself reports a missing library / device
        pass
    return ""


def bind_to_gpu(local_rank, peers_allgather=None, rank=0, sysfs_root="/sys", bdf=None, apply=True):
    """Bind this process to the cores next to GPU `local_rank` (see above).  `peers_allgather`: a callable that exchanges one object with
    every rank (NodeGroup's star) so that ranks sharing a NUMA node split its cores; without it the rank takes the whole node.
    Returns the record bench.py prints per rank: {"pci", "numa_node", "cpus_local", "cpus_bound" (None: not bound), "note"}."""
    loc = gpu_locality(local_pci_bus_id(local_rank) if bdf is None else bdf, sysfs_root)
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = []
    everyone = peers_allgather([loc["pci"] or "", loc["numa_node"] if loc["numa_node"] is not None else -1, np.array(loc["cpus"], dtype=np.int64)]) if peers_allgather else None
    if everyone is not None:
        locs = [{"pci": e[0], "numa_node": e[1], "cpus": [int(c) for c in e[2]]} for e in everyone]
        mine = plan_affinity(locs, allowed, sysfs_root)[rank]
    else:
        mine = plan_affinity([loc], allowed, sysfs_root)[0]
    rec = {"pci": loc["pci"], "numa_node": loc["numa_node"], "cpus_local": len(loc["cpus"]), "cpus_allowed": len(allowed), "cpus_bound": None, "note": None}
    if not mine:
        rec["note"] = "no local cores known for this GPU inside the allowed set: left where the launcher put it"
        return rec
    if apply:
        try:
            set_affinity_of_process(mine)
        except (AttributeError, OSError) as e:
            rec["note"] = "sched_setaffinity refused: %s" % e
            return rec
    rec["cpus_bound"] = _compress_cpulist(mine)
    return rec


def bind_worker(device, all_devices, sysfs_root="/sys", apply=True):
    """The same for processes that never talk to each other (callVarBamParallel --run: one worker per GPU): every worker looks up the
    PCI address of EVERY worker's device itself, so all of them arrive at the same split.  Bound only when more than one GPU is in use
    (or CLAIR_AMD_BIND=1)."""
    devices = sorted(set(int(d_) for d_ in all_devices))
    if not want_binding(len(devices)) or int(device) not in devices:
        return None
    locs = [gpu_locality(local_pci_bus_id(d_), sysfs_root) for d_ in devices]
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    me = devices.index(int(device))
    mine = plan_affinity(locs, allowed, sysfs_root)[me]
    rec = {"pci": locs[me]["pci"], "numa_node": locs[me]["numa_node"], "cpus_local": len(locs[me]["cpus"]), "cpus_allowed": len(allowed), "cpus_bound": None, "note": None}
    if not mine:
        rec["note"] = "no local cores known for this GPU inside the allowed set"
        return rec
    if apply:
        try:
            set_affinity_of_process(mine)
        except (AttributeError, OSError) as e:
            rec["note"] = "sched_setaffinity refused: %s" % e
            return rec
    rec["cpus_bound"] = _compress_cpulist(mine)
    return rec


def _compress_cpulist(cpus):
    out, i = [], 0
    cpus = sorted(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append("%d" % cpus[i] if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def rccl_init_timeout(socket_timeout=180.0):
    """Seconds the collective part of the RCCL bring-up (ncclCommInitRank + the first all-reduce) may take on a rank before that rank
    gives it up: CLAIR_AMD_RCCL_INIT_TIMEOUT, default 60 s, and never more than a third of the bootstrap sockets' own timeout (the
    ranks that did come up wait on a socket for the one that is still counting)."""
    try:
        v = float(os.environ.get("CLAIR_AMD_RCCL_INIT_TIMEOUT", "60"))
    except ValueError:
        v = 60.0
    return max(0.001, min(v, socket_timeout / 3.0))


def want_binding(world):
    v = os.environ.get("CLAIR_AMD_BIND", "")
    return v == "1" or (v != "0" and world > 1)


# -- wire format of the bootstrap / CPU transport ------------------------------------------------------------------------------------
# Fixed framing, no pickle: a peer on 127.0.0.1 can make a rank parse bytes, never run code.  frame = b"CLSH" | kind u8 | 3 pad |
# payload length u64 | payload.  kinds: 0 None, 1 int64, 2 bytes, 3 utf-8 string, 4 ndarray (dtype code u8, ndim u8, 6 pad, dims
# int64 x ndim, raw C-order bytes), 5 list / 6 tuple (count u32, then that many frames).  Before any frame a connecting rank
# presents the 32-character token rank 0 wrote next to its port in the (0600, owner-checked) rendezvous file.
_MAGIC = b"CLSH"
_DTYPES = {0: np.dtype("<f4"), 1: np.dtype("<f8"), 2: np.dtype("<i8"), 3: np.dtype("u1"), 4: np.dtype("<i4")}
_DTYPE_CODES = {v: k for k, v in _DTYPES.items()}
_MAX_PAYLOAD = 1 << 36
_TOKEN_BYTES = 32


def _encode(obj):
    if obj is None:
        kind, payload = 0, b""
    elif isinstance(obj, (bool, int, np.integer)):
        kind, payload = 1, struct.pack("<q", int(obj))
    elif isinstance(obj, (bytes, bytearray)):
        kind, payload = 2, bytes(obj)
    elif isinstance(obj, str):
        kind, payload = 3, obj.encode("utf-8")
    elif isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
        if np.dtype(dt) not in _DTYPE_CODES:
            raise TypeError("shard transport carries float32/float64/int64/int32/uint8 arrays, not %s" % a.dtype)
        a = a.astype(dt, copy=False)
        kind = 4
        payload = struct.pack("<BB6x", _DTYPE_CODES[np.dtype(dt)], a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape) + a.tobytes()
    elif isinstance(obj, (list, tuple)):
        kind = 5 if isinstance(obj, list) else 6
        payload = struct.pack("<I", len(obj)) + b"".join(_encode(o) for o in obj)
    else:
        raise TypeError("shard transport cannot carry %r" % type(obj))
    return _MAGIC + struct.pack("<B3xQ", kind, len(payload)) + payload


def _decode(buf, at=0):
    """One frame of `buf` starting at `at` -> (object, next offset).  Raises ValueError on anything malformed."""
    if len(buf) - at < 16 or bytes(buf[at:at + 4]) != _MAGIC:
        raise ValueError("bootstrap: bad frame header")
    kind, n = struct.unpack_from("<B3xQ", buf, at + 4)
    at += 16
    if n > len(buf) - at:
        raise ValueError("bootstrap: truncated frame")
    body, nxt = memoryview(buf)[at:at + n], at + n
    if kind == 0 and n == 0:
        return None, nxt
    if kind == 1 and n == 8:
        return struct.unpack("<q", body)[0], nxt
    if kind == 2:
        return bytes(body), nxt
    if kind == 3:
        return bytes(body).decode("utf-8"), nxt
    if kind == 4 and n >= 8:
        code, ndim = struct.unpack_from("<BB6x", body, 0)
        if code not in _DTYPES or ndim > 8 or n < 8 + 8 * ndim:
            raise ValueError("bootstrap: bad array header")
        dims = struct.unpack_from("<%dq" % ndim, body, 8)
        count = 1
        for d_ in dims:
            if d_ < 0:
                raise ValueError("bootstrap: negative dimension")
            count *= d_
        if count * _DTYPES[code].itemsize != n - 8 - 8 * ndim:
            raise ValueError("bootstrap: array size does not match its shape")
        return np.frombuffer(bytes(body[8 + 8 * ndim:]), dtype=_DTYPES[code]).reshape(dims).copy(), nxt
    if kind in (5, 6) and n >= 4:
        (count,) = struct.unpack_from("<I", body, 0)
        items, pos = [], at + 4
        for _ in range(count):
            item, pos = _decode(buf, pos)
            if pos > nxt:
                raise ValueError("bootstrap: nested frame overruns its container")
            items.append(item)
        if pos != nxt:
            raise ValueError("bootstrap: container length mismatch")
        return (items if kind == 5 else tuple(items)), nxt
    raise ValueError("bootstrap: unknown frame kind %d" % kind)


def _send_msg(sock, obj):
    sock.sendall(_encode(obj))


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("peer closed the bootstrap connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    head = _recv_exact(sock, 16)
    if head[:4] != _MAGIC:
        raise ValueError("bootstrap: bad frame header")
    (n,) = struct.unpack_from("<Q", head, 8)
    if n > _MAX_PAYLOAD:
        raise ValueError("bootstrap: frame of %d bytes refused" % n)
    obj, _ = _decode(head + _recv_exact(sock, n))
    return obj


def _publish(path, port, token):
    """Rank 0's port and token, visible atomically under `path`, readable by this user only (mkstemp: 0600, fresh name, no symlink
    is followed; os.replace swaps the directory entry)."""
    fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", dir=os.path.dirname(path) or ".")
    try:
        os.write(fd, ("%d %s\n" % (port, token)).encode())
    finally:
        os.close(fd)
    os.replace(tmp, path)


def _read_published(path):
    """(port, token) or None while rank 0 has not published yet.  A file that is a symlink, belongs to someone else or is readable
    by others is refused: its content is not rank 0's."""
    try:
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    except FileNotFoundError:
        return None
    except OSError as e:
        raise RuntimeError("bootstrap: refusing rendezvous file %s: %s" % (path, e))
    try:
        st = os.fstat(fd)
        if st.st_uid != os.geteuid() or (st.st_mode & 0o077):
            raise RuntimeError("bootstrap: refusing rendezvous file %s (owner %d, mode %o): not written by this job" % (path, st.st_uid, st.st_mode & 0o777))
        text = os.read(fd, 256).decode("ascii", "replace").split()
    finally:
        os.close(fd)
    if len(text) != 2 or not text[0].isdigit() or len(text[1]) != _TOKEN_BYTES:
        return None
    return int(text[0]), text[1]


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
            import hmac
            import secrets
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(world)
            token = secrets.token_hex(_TOKEN_BYTES // 2)
            _publish(path, srv.getsockname()[1], token)
            self._path = path
            socks = {}
            deadline = time.time() + timeout
            try:
                while len(socks) < world - 1:
                    srv.settimeout(max(0.01, deadline - time.time()))
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(min(timeout, 10.0))
                    try:          # nothing of a peer is parsed before it has shown the token
                        if not hmac.compare_digest(_recv_exact(conn, _TOKEN_BYTES), token.encode()):
                            raise ValueError("wrong token")
                        r = _recv_msg(conn)
                        if not isinstance(r, int) or not 0 < r < world or r in socks:
                            raise ValueError("unexpected rank announcement %r" % (r,))
                    except (ValueError, ConnectionError, socket.timeout, OSError):
                        conn.close()      # not one of ours: keep waiting for the real ranks
                        continue
                    conn.settimeout(timeout)
                    socks[r] = conn
            except socket.timeout:
                missing = [r for r in range(1, world) if r not in socks]
                raise RuntimeError("bootstrap: rank(s) %s of %d never joined within %.0f s" % (", ".join(map(str, missing)), world, timeout))
            finally:
                srv.close()
                try:
                    os.unlink(path)
                except OSError:
                    pass
            self.peers = [socks[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            found = None
            while found is None:
                found = _read_published(path)
                if found is None:
                    if time.time() > deadline:
                        raise RuntimeError("bootstrap: rank 0 never published %s (rank %d waited %.0f s)" % (path, rank, timeout))
                    time.sleep(0.01)
            port, token = found
            s = socket.create_connection(("127.0.0.1", port), timeout=timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(token.encode())
            _send_msg(s, rank)
            self.root = s

    def allgather(self, obj):
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            items = [obj]
            for r, p in enumerate(self.peers, start=1):
                try:
                    items.append(_recv_msg(p))
                except socket.timeout:
                    raise RuntimeError("rank %d did not answer within the timeout" % r)
            for p in self.peers:
                _send_msg(p, items)
            return items
        _send_msg(self.root, obj)
        try:
            return _recv_msg(self.root)
        except socket.timeout:
            raise RuntimeError("rank 0 did not answer within the timeout (a peer rank is missing or stuck)")

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
    opened.  transport: "rccl" | "tcp" | None (= "rccl" when the HIP library sees a device, else "tcp").
    bind: pin this process to the host cores of its GPU's NUMA node (None: when WORLD_SIZE > 1 and the transport is not the CPU one, or
    CLAIR_AMD_BIND=1; `sysfs_root` / `bdf` exist for the CPU tests); the record is `self.affinity`."""

    def __init__(self, transport=None, timeout=180.0, rank=None, world=None, local_rank=None, bind=None, sysfs_root="/sys", bdf=None, device=None,
                 init_timeout=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank))) if local_rank is None else int(local_rank)
        self.device = self.local_rank if device is None else int(device)      # the HIP device of this rank
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside world of %d" % (self.rank, self.world))
        self.transport = "none"
        self.rccl_failure = None          # why an asked-for RCCL communicator is not there (then transport == "tcp")
        self.rccl_abandoned = False       # this rank's RCCL bring-up ran into its deadline: a helper thread may still sit inside librccl
        self._lib = None
        self._comm = None
        self.timeout = timeout
        self._star = _TcpStar(self.rank, self.world, rendezvous_path(), timeout)
        # every rank next to its own GPU (see bind_to_gpu), before any buffer of the engine is allocated and first touched
        self.affinity = None
        if bind is None:          # over the CPU transport (tests without a GPU) only when asked for by name
            bind = want_binding(self.world) and (transport != "tcp" or os.environ.get("CLAIR_AMD_BIND") == "1")
        if bind:
            self.affinity = bind_to_gpu(self.device, self._star.allgather if self.world > 1 else None, self.rank, sysfs_root=sysfs_root, bdf=bdf)
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
            # Pre-flight over the sockets, BEFORE anything collective: a rank whose device ordinal is out of range or that cannot
            # load librccl would otherwise raise alone and leave the others inside ncclCommInitRank for ever.
            mine = "" if lib.clair_comm_preflight(self.device) == 0 else lib.clair_comm_last_error(None).decode()
            status = self._star.allgather(mine)
            bad = ["rank %d: %s" % (r, m) for r, m in enumerate(status) if m]
            if bad:
                self._star.close()
                raise _capi.EngineError("RCCL start-up refused on %d of %d ranks -- %s" % (len(bad), self.world, "; ".join(bad)))
            # From here on a failure is RCCL's own start-up (no usable bootstrap interface, a library / driver mismatch, a bring-up that
            # HANGS): every rank learns of it over the sockets and ALL of them go on with the "tcp" transport -- the data path has no
            # collective, what is carried is the timing barrier, one max and the 9.5 MB of weights.  Said on stderr and in bench.py's
            # line (`config.transport`, `config.rccl_failure`).  The collective part -- ncclCommInitRank and the first all-reduce on the
            # new communicator -- runs under a deadline of its own (clair_comm_create_timed, `init_timeout`): a rank whose RCCL never
            # returns reports "timed out" like any other failure, well inside the sockets' `timeout`.
            uid = (ctypes.c_uint8 * 128)()
            first = None
            if self.rank == 0:
                ok = lib.clair_comm_unique_id(uid) == 0
                first = ("id", bytes(uid)) if ok else ("error", lib.clair_comm_last_error(None).decode())
            got = self._star.allgather(first)[0]
            if got[0] != "id":
                self._fall_back("rank 0 could not create the RCCL unique id: %s" % got[1])
                return
            uid = (ctypes.c_uint8 * 128).from_buffer_copy(got[1])
            h = ctypes.c_void_p()
            init_timeout = rccl_init_timeout(timeout) if init_timeout is None else float(init_timeout)
            timed = getattr(lib, "clair_comm_create_timed", None)
            if timed is not None:
                rc = timed(self.device, self.rank, self.world, uid, max(1, int(init_timeout * 1000)), ctypes.byref(h))
            else:                     # an older build of the library (CLAIR_AMD_LIB): no deadline
                rc = lib.clair_comm_create(self.device, self.rank, self.world, uid, ctypes.byref(h))
            mine = "" if rc == 0 else lib.clair_comm_last_error(None).decode()
            if rc == 2:               # CLAIR_COMM_TIMED_OUT: a helper thread of this process is still inside RCCL (see close_process)
                self.rccl_abandoned = True
                mine = "init timed out on rank %d after %g s (%s)" % (self.rank, init_timeout, mine)
            status = self._star.allgather(mine)
            bad = ["rank %d: %s" % (r, m) for r, m in enumerate(status) if m]
            if bad:
                if not mine:          # a peer never came up: abort, do not destroy (ncclCommDestroy may wait for it)
                    getattr(lib, "clair_comm_abort", lib.clair_comm_destroy)(h)
                self._fall_back("clair_comm_create failed on %d of %d ranks -- %s" % (len(bad), self.world, "; ".join(bad)))
                return
            self._lib, self._comm = lib, h

    def _fall_back(self, why):
        self.transport, self.rccl_failure = "tcp", why
        if self.rank == 0:
            print("[clair_amd.shard] RCCL start-up failed (%s): barrier, reductions and the weight broadcast of this run go over the bootstrap "
                  "sockets on 127.0.0.1 instead" % why, file=sys.stderr)

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

    def gather_objects(self, obj):
        """One small object per rank (None, int, str, bytes, arrays, lists / tuples of those), in rank order, on every rank -- always over
        the bootstrap sockets: bookkeeping (placement records, sampled clocks), never data."""
        return self._star.allgather(obj)

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

    def close(self, barrier=True):
        """Leave the group.  The closing barrier goes over the sockets (they time out; an RCCL collective after a peer has died
        does not) and is skipped when the caller is unwinding from an error (barrier=False)."""
        if self.world > 1 and barrier:
            try:
                self._star.allgather(None)
            except Exception:
                pass
        if self._comm is not None:
            self._lib.clair_comm_destroy(self._comm)
            self._comm = None
        self._star.close()

    def exit_process(self, rc):
        """Leave the PROCESS with `rc` once the caller has written its results.  Normally sys.exit; after an abandoned RCCL bring-up
        (`rccl_abandoned`) a helper thread still sits inside librccl, whose own threads and static destructors may wait for it for ever:
        then stdout / stderr are flushed and the process ends with os._exit -- the N-rank line is already out."""
        if self.rccl_abandoned:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(int(rc))
        sys.exit(int(rc))


def spawn_ranks(argv, world, env=None, rdzv_dir=None, stderr_pipe=False):
    """Start `world` copies of `argv` (one per GPU: RANK = LOCAL_RANK = 0..world-1) the way torch.distributed.run would,
    with a private rendezvous file (a fresh 0700 directory).  Returns the list of Popen objects (stdout of rank 0 is a pipe, the
    others inherit; stderr_pipe=True makes every rank's stderr a pipe for the caller to relay)."""
    import subprocess
    base = dict(os.environ if env is None else env)
    d = rdzv_dir or tempfile.mkdtemp(prefix="clair_amd_rdzv_")
    procs = []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                 CLAIR_AMD_RDZV=os.path.join(d, "port"))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else None, stderr=subprocess.PIPE if stderr_pipe else None))
    return procs
