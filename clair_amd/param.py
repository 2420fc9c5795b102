"""Tensor geometry and batch defaults (counterpart of /root/reference/shared/param.py:1-16).

Only the values the inference path reads are kept; training hyper-parameters are out of scope.
"""
REPO_NAME = "Clair"
NUM_THREADS = 12            # shared/param.py:3 (host-side threads; the GPU engine ignores it)
flankingBaseNum = 16        # shared/param.py:9
matrixRow = 8               # shared/param.py:10
matrixNum = 4               # shared/param.py:11
predictBatchSize = 1000     # shared/param.py:16
engineBatchSize = 4096      # candidates per forward pass when --batch_size is not given: results do not depend on it, and at the reference's 1 000 the
                            # host side of call_var (queues, ctypes calls, one NumPy view per batch) keeps the engine at half its rate
                            # (3.0-4.1 against 5.3-6.7 M candidates/s inside call_variants, profiles/r04_e2e_binary.txt)
no_of_positions = 2 * flankingBaseNum + 1
input_tensor_size = no_of_positions * matrixRow * matrixNum  # 1056


def pipeline_slots():
    """Batches the callers keep in flight at the engine's host boundary (clair_submit* / clair_wait).  Three compute lanes fill the
    chip; twice as many slots keep every lane fed while its other batch is on the host link.  CLAIR_AMD_SLOTS
    overrides (1 and 2 select the fused layer-2 launch of one- and two-lane handles; up to 4 slots get a lane each -- the configuration of a
    caller whose batches are already in HBM -- and 4 slots fed from the host are the one bad choice: four lanes and the copy stream are five
    streams on four hardware queues)."""
    import os
    try:
        return max(1, min(64, int(os.environ.get("CLAIR_AMD_SLOTS", "6"))))
    except ValueError:
        return 6
