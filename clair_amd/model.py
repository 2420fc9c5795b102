"""Drop-in for the inference surface of ``clair.model.Clair`` on MI355X.

Mirrors the members call_var / evaluate use on the reference class
(/root/reference/clair/model.py):

  Clair(**kwargs)              :58-192   constructor (graph build + session there; engine handle here)
  .init()                      :807-813  no-op here (there: run the variable initialiser)
  .restore_parameters(prefix)  :1016-1020  load a weight container / checkpoint prefix
  .predict(batchX)             :946-966  -> [gt21 [n,21], genotype [n,3], len1 [n,33], len2 [n,33]]
                                          float32; also stored in ``self.prediction``
  .prediction                  :964
  .close() / __del__           :872-876, 1149-1152

The forward pass itself runs in hand-written HIP kernels behind the C ABI of
include/clair_amd.h (clair_amd/csrc); there is no CPU or framework fallback.
"""
import numpy as np

from clair_amd import _capi, param, weights


class Clair(object):
    """MI355X engine behind the reference's ``Clair`` interface (inference members only)."""

    def __init__(self, **kwargs):
        self.device = int(kwargs.pop("device", 0))
        self.max_batch = int(kwargs.pop("max_batch", max(param.predictBatchSize, 1024)))
        self.n_slots = int(kwargs.pop("n_slots", 2))
        # the reference reports unsupported kwargs instead of failing (clair/model.py:112-116)
        for key, value in kwargs.items():
            print("Info: the parameter %s, with value %s is not supported" % (key, value))
        self.input_shape = (2 * param.flankingBaseNum + 1, param.matrixRow, param.matrixNum)
        self.output_gt21_shape = 21
        self.output_genotype_shape = 3
        self.output_indel_length_shape_1 = 33
        self.output_indel_length_shape_2 = 33
        self.prediction = None
        self.layers = []
        self._engine = _capi.Engine(self.device, self.max_batch, self.n_slots)
        self._weights_loaded = False

    # -- reference interface ---------------------------------------------------------------
    def init(self):
        """clair/model.py:807-813 runs the TF initialiser; weights here come only from
        restore_parameters / set_parameters, so there is nothing to do."""
        return None

    def restore_parameters(self, file_name):
        """clair/model.py:1016-1020.  ``file_name`` is a checkpoint prefix (or an .npz container)."""
        self.set_parameters(weights.load_weights(file_name))

    def set_parameters(self, w):
        """Load weights from a dict of arrays keyed as clair_amd.weights.TENSOR_TABLE."""
        self._engine.load_weights(w)
        self._weights_loaded = True

    def predict(self, batchX):
        """clair/model.py:946-966: list of four float32 arrays; kept in ``self.prediction``."""
        x = np.asarray(batchX)
        n = x.shape[0]
        if n <= self.max_batch:
            prediction = self._engine.predict(x)
        else:  # larger than one engine batch: pipeline max_batch pieces over the slots
            prediction = self._predict_pieces(x)
        self.prediction = prediction
        return prediction

    def close(self):
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_summary_file_writer(self, logs_path):
        """Dead path in the reference too (clair/model.py:1053-1062 returns None)."""
        return None

    # -- pipelined helpers (used by clair_amd.call_var) ----------------------------------------
    def submit(self, slot, batchX):
        self._engine.submit(slot, batchX)

    def submit_counts(self, slot, counts):
        """submit() for raw pileup counts [n,33,8,4] int16 (before clair/utils.py:96-98): the subtraction and the conversion
        run on the device, the host link carries half the bytes.  Same outputs as submit() on the float32 tensor."""
        self._engine.submit_counts(slot, counts)

    def submit_calls(self, slot, batch, centre, counts=False, with_probabilities=False):
        """Forward pass + decode on the device (include/clair_amd.h: clair_submit_ex): wait(slot) returns the call records of
        include/clair_call.h -- or (records, probabilities) -- instead of the probabilities."""
        self._engine.submit_calls(slot, batch, centre, counts=counts, with_probabilities=with_probabilities)

    def pinned_buffer(self, nbytes):
        """Page-locked host memory of the engine (include/clair_amd.h: clair_pinned_alloc) as a uint8 array."""
        return self._engine.pinned_buffer(nbytes)

    def wait(self, slot):
        return self._engine.wait(slot)

    def _predict_pieces(self, x):
        pieces, inflight = [], []
        mb, ns = self.max_batch, self.n_slots
        for k, start in enumerate(range(0, x.shape[0], mb)):
            slot = k % ns
            if len(inflight) == ns:
                pieces.append(self._engine.wait(inflight.pop(0)))
            self._engine.submit(slot, x[start:start + mb])
            inflight.append(slot)
        while inflight:
            pieces.append(self._engine.wait(inflight.pop(0)))
        return [np.concatenate([p[i] for p in pieces], axis=0) for i in range(4)]

    @property
    def engine(self):
        return self._engine
