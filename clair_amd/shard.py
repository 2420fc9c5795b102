"""Candidate sharding across the GPUs of one node (one process per GPU).

The reference scales out by printing one ``callVarBam`` shell command per 10 Mbp chunk and
letting GNU parallel run them (/root/reference/clair/callVarBamParallel.py:90-119,
README.md:297); chunk VCFs are concatenated in order afterwards (README.md:303).  Candidates are
classified independently (docs/POST_PROCESSING.md:17), so here rank r simply owns a contiguous
block of whole batches of the candidate stream -- per-rank output fragments concatenate in
input order -- and there is no data-path collective.  torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in CPU tests) carries only the barrier, the MAX-reduce of the
elapsed time and the gather of per-rank counters.  The forward pass itself never touches torch.
"""
import os


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


class NodeGroup(object):
    """Process group of the ranks of one node, created from the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  With WORLD_SIZE == 1 nothing
    is imported or initialised."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._dist = None
        self._torch = None
        self._device = "cpu"
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            kwargs = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self._device = "cuda"
                kwargs["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, **kwargs)

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max_float(self, value):
        if self._dist is None:
            return float(value)
        t = self._torch.tensor([float(value)], dtype=self._torch.float64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, value):
        if self._dist is None:
            return int(value)
        t = self._torch.tensor([int(value)], dtype=self._torch.int64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return int(t.item())

    def gather_arrays(self, array):
        """All ranks' float32 arrays (first dims may differ) concatenated in rank order, on every rank."""
        import numpy as np
        if self._dist is None:
            return np.asarray(array)
        torch, dist = self._torch, self._dist
        a = np.ascontiguousarray(array, dtype=np.float32)
        counts = [torch.zeros(1, dtype=torch.int64, device=self._device) for _ in range(self.world)]
        dist.all_gather(counts, torch.tensor([a.shape[0]], dtype=torch.int64, device=self._device))
        counts = [int(c.item()) for c in counts]
        width = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
        pad = torch.zeros((max(counts), width), dtype=torch.float32, device=self._device)
        if a.shape[0]:
            pad[:a.shape[0]] = torch.from_numpy(a.reshape(a.shape[0], width)).to(self._device)
        parts = [torch.zeros_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad)
        out = np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)
        return out.reshape((out.shape[0],) + a.shape[1:])

    def close(self):
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None
