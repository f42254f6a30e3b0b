"""Multi-GPU plumbing: one process per GPU, reads/batches sharded with no data-path collective,
one integer all-reduce of the per-position counters at the end.  The reference's
equivalent is "run several processes on different inputs, then add the BED files"
(docs/Usage.md:22-27, DeepMod_tools/sum_chr_mod.py:47-52)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin shard (the reference's queue gives the same 'any worker takes the next batch'
    distribution; round-robin makes it deterministic)."""
    return [it for i, it in enumerate(items) if i % world == rank]


def all_reduce_counts(counts: np.ndarray, dist) -> np.ndarray:
    """Sum host int32 counters over all ranks of an initialised torch.distributed group (gloo on
    CPU; with nccl the tensor is staged through the current GPU).  Integer sum: order independent."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(counts))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
