"""Where the ~0.43 s of process start-up of call_var go (one GPU)."""
import time
t0 = time.perf_counter()
import sys
sys.path.insert(0, ".")
import numpy as np  # noqa: E402
t1 = time.perf_counter()
from clair_amd import call_var, _capi, _hostapi, weights, synth  # noqa: E402,F401
t2 = time.perf_counter()
w = weights.synthetic_weights(seed=1)
weights.save_weights("gpurun_out/su_model", w)
t3 = time.perf_counter()
from clair_amd.model import Clair  # noqa: E402
m = Clair(device=0, max_batch=1024, n_slots=2)
m.init()
t4 = time.perf_counter()
m.restore_parameters("gpurun_out/su_model")
t5 = time.perf_counter()
x, _ = synth.synthetic_input(1024, "ont", seed=1)
t6 = time.perf_counter()
m.predict(x)
t7 = time.perf_counter()
m.predict(x)
t8 = time.perf_counter()
print("import numpy %.3f | import clair_amd %.3f | (make+save weights %.3f) | engine create %.3f | restore_parameters %.3f | first predict %.3f | second %.4f"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t7 - t6, t8 - t7))
m.close()
