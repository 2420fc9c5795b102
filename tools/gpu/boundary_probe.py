"""Where the host-array boundary (clair_submit / clair_wait) loses its time: VERDICT r03 item 1.

1. raw hipMemcpyAsync rates out of / into page-locked memory by transfer size and number of streams, with the chip idle and with
   the forward pass running beside them (resident loop on another thread);
2. the submit/wait loop of tools/pcie_rate.py by number of slots in flight, with the CPU time spent inside submit and wait."""
import ctypes
import sys
import threading
import time
import numpy as np
sys.path.insert(0, ".")
from clair_amd import _capi, synth, weights

hip = ctypes.CDLL("libamdhip64.so")
H2D, D2H = 1, 2


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> %d" % (what, rc))


def raw_copy(kind, nbytes, n_streams, reps):
    streams, hbuf, dbuf = [], [], []
    for _ in range(n_streams):
        s, h, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        chk(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1), "stream")
        chk(hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(nbytes), 0), "hostmalloc")
        chk(hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(nbytes)), "malloc")
        ctypes.memset(h, 1, nbytes)
        streams.append(s); hbuf.append(h); dbuf.append(d)
    def once(k):
        for i in range(k):
            j = i % n_streams
            if kind == H2D:
                chk(hip.hipMemcpyAsync(dbuf[j], hbuf[j], ctypes.c_size_t(nbytes), H2D, streams[j]), "h2d")
            else:
                chk(hip.hipMemcpyAsync(hbuf[j], dbuf[j], ctypes.c_size_t(nbytes), D2H, streams[j]), "d2h")
        for s in streams:
            chk(hip.hipStreamSynchronize(s), "sync")
    once(2 * n_streams)
    t0 = time.perf_counter()
    once(reps)
    dt = time.perf_counter() - t0
    for s, h, d in zip(streams, hbuf, dbuf):
        hip.hipStreamDestroy(s); hip.hipHostFree(h); hip.hipFree(d)
    return reps * nbytes / dt / 1e9, dt / reps * 1e6


def copy_table(tag):
    print("## raw hipMemcpyAsync, page-locked host memory, %s" % tag)
    for kind, name in ((H2D, "H2D"), (D2H, "D2H")):
        for nbytes in (368640, 1 << 20, 2162688, 4325376, 16 << 20, 64 << 20):
            row = []
            for ns in (1, 3):
                gbs, us = raw_copy(kind, nbytes, ns, max(12, min(300, int(2e9 / nbytes))))
                row.append("%d stream(s): %5.1f GB/s (%6.1f us/copy)" % (ns, gbs, us))
            print("%s %9d B   %s" % (name, nbytes, "   ".join(row)))
    sys.stdout.flush()


W = weights.synthetic_weights(seed=20250928, head_gain=4.0)
RAW = "--no-raw" not in sys.argv
if RAW:
    copy_table("chip idle")

# the same with the forward pass running on three slots beside the copies
eng = _capi.Engine(device=0, max_batch=1024, n_slots=3)
eng.load_weights(W)
N = 1024 * 24
xd, od = eng.dataset_alloc(N)
x = synth.synthetic_input(1024, "ont", seed=1)[0]
for b in range(24):
    eng.dataset_upload(xd, b * 1024, x)
stop = False
steps = [0]


def spin():
    while not stop:
        for b in range(24):
            eng.run_resident(b % 3, xd, od, b * 1024, 1024)
        eng.sync()
        steps[0] += 24


if RAW:
    th = threading.Thread(target=spin)
    th.start()
    time.sleep(0.5)
    s0, t0 = steps[0], time.perf_counter()
    copy_table("forward pass resident on 3 slots beside it")
    s1, t1 = steps[0], time.perf_counter()
    stop = True
    th.join()
    print("(resident loop meanwhile: %.2f M candidates/s; the two Python threads share the interpreter lock)" % ((s1 - s0) * 1024 / (t1 - t0) / 1e6))
eng.dataset_free(xd, od)
eng.close()

print("## submit/wait loop by slots in flight (batch 1024, host arrays in and out)")
def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


MODES = [{"int16": "int16 counts"}.get(m, m) for m in arg("--modes", "pageable,pinned,int16").split(",")]
for n_slots in [int(v) for v in arg("--slots", "3,4,6,9,12").split(",")]:
    eng = _capi.Engine(device=0, max_batch=1024, n_slots=n_slots)
    eng.load_weights(W)
    xs = [synth.synthetic_input(1024, "ont", seed=s)[0] for s in range(n_slots)]
    bufs = [eng.slot_input(s) for s in range(n_slots)]
    cs = []
    for xx in xs:
        c = xx.copy(); c[..., 1:] += c[..., 0:1]; cs.append(c.astype(np.int16))
    for mode in MODES:
        def loop(rounds):
            tsub = twait = 0.0
            t0 = time.perf_counter()
            for r in range(rounds):
                for s in range(n_slots):
                    if r:
                        a = time.perf_counter(); eng.wait(s); twait += time.perf_counter() - a
                    a = time.perf_counter()
                    if mode == "pageable":
                        eng.submit(s, xs[s])
                    elif mode == "int16 counts":
                        eng.submit_counts(s, cs[s])
                    else:
                        if r == 0:
                            np.copyto(bufs[s], xs[s])
                        eng.submit(s, bufs[s])
                    tsub += time.perf_counter() - a
            for s in range(n_slots):
                eng.wait(s)
            dt = time.perf_counter() - t0
            nb = rounds * n_slots
            return nb * 1024 / dt, tsub / nb * 1e6, twait / nb * 1e6
        loop(5)
        rate, us_sub, us_wait = loop(max(40, 600 // n_slots))
        print("%2d slots  %-13s %9.0f candidates/s   submit %6.1f us  wait %6.1f us per batch (CPU wall inside the calls)" % (n_slots, mode, rate, us_sub, us_wait))
        sys.stdout.flush()
    eng.close()
