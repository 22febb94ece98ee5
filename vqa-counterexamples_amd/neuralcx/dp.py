"""Data parallelism for the NeuralCX path: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-GPU (its nn.DataParallel wrapper is unwrapped at counterexamples.py:221-225), so this
is net-new; the only behaviour to match is "same result as one big batch".  Scheme:
  * one global permutation per epoch (seed 42, counterexamples.py:119), sliced contiguously by rank;
  * every rank scales its loss by 1/B_global (ncx_dims.loss_scale), so a plain SUM all-reduce of the flat
    gradient buffer reproduces the global-batch gradient of counterexamples.py:334;
  * metrics (loss sum, recall hits, count) are reduced as 3 scalars only when they are printed.
"""
import os
import random
from typing import List, Optional

import torch
import torch.distributed as dist


def init_distributed(device_index: Optional[int] = None, backend: Optional[str] = None):
    """-> (rank, world, local_rank).  Reads the torchrun environment; world 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:             # NCX_DIST_BACKEND=gloo: rehearsal of several ranks on one card
            backend = os.environ.get("NCX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local if device_index is None else device_index)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def epoch_batches(n_examples: int, global_batch: int, epoch: int, seed: int = 42, shuffle: bool = True) -> List[List[int]]:
    """Global batches of example ids for one epoch; identical on every rank (batchify, counterexamples.py:509-516,
    keeps the last partial batch)."""
    ids = list(range(n_examples))
    if shuffle:
        random.Random(seed * 100003 + epoch).shuffle(ids)
    return [ids[i:i + global_batch] for i in range(0, n_examples, global_batch)]


def epoch_plan(n_examples: int, global_batch: int, epoch: int, rank: int, world: int, device, seed: int = 42,
               shuffle: bool = True):
    """The same batches as `epoch_batches` + `shard`, with ONE host-to-device copy per epoch: returns
    (ids int64 [n_examples] on `device`, [(lo, hi, global_size, first_id, active)] per global batch) where ids[lo:hi] is
    this rank's slice.  Per-step index tensors are then device slices: a pageable host-to-device copy per step would
    make the host wait for the stream to drain every step.

    A rank whose slice of a (short, last) global batch is empty -- batchify keeps the partial batch,
    counterexamples.py:513-515, so len(batch) < world happens -- still gets an entry: ONE padding triplet (the batch's
    first) with active = False.  It runs the whole step with loss weight 0 (zero gradients), so every rank enters every
    collective of every step and advances its step counter (Adam bias correction, dropout seeds) in lockstep."""
    batches = epoch_batches(n_examples, global_batch, epoch, seed, shuffle)
    flat = torch.tensor([i for b in batches for i in b], dtype=torch.int64).to(device)
    plan, off = [], 0
    for b in batches:
        n = len(b)
        lo, hi = n * rank // world, n * (rank + 1) // world
        if hi > lo:
            plan.append((off + lo, off + hi, n, b[lo], True))
        else:
            plan.append((off, off + 1, n, b[0], False))          # padding triplet, weight 0
        off += n
    return flat, plan


def shard(batch_ids: List[int], rank: int, world: int) -> List[int]:
    """Contiguous slice of a global batch owned by `rank` (sizes differ by at most one)."""
    n = len(batch_ids)
    lo = n * rank // world
    hi = n * (rank + 1) // world
    return batch_ids[lo:hi]


def allreduce_sum_(flat: torch.Tensor, group=None, bucket_elems: int = 0):
    """In-place SUM all-reduce of a flat buffer, optionally in buckets (async, then waited)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if bucket_elems <= 0 or flat.numel() <= bucket_elems:
        dist.all_reduce(flat, group=group)
        return flat
    works = [dist.all_reduce(flat[o:o + bucket_elems], group=group, async_op=True)
             for o in range(0, flat.numel(), bucket_elems)]
    for w in works:
        w.wait()
    return flat


def reduce_metrics(loss_sum: float, hits1: int, hits5: int, count: int, device, group=None):
    t = torch.tensor([loss_sum, hits1, hits5, count], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return float(t[0]), int(t[1]), int(t[2]), int(t[3])
