"""Multi-GPU plumbing for the ASR path: one process per GPU (`torch.distributed`, backend "nccl" =
RCCL over xGMI; "gloo" in the CPU tests), utterances (30 s windows) are the independent units and are sharded
across ranks, each rank holds a full weight replica that arrives by ONE broadcast of the weight arena at load
time; there is no collective inside an utterance and none at request time (results travel as small Python
objects to rank 0).  The reference has no distributed code at all: its only knob is CT2's
replica pool `device_index=[0..n-1]` (main.py:295), which this replaces.
"""
import numpy as np


def shard_range(n_items, world, rank):
    """Contiguous, balanced partition of range(n_items): the first n % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_arena(arena, nbytes, src=0, device=None, group=None):
    """Broadcast the flat weight arena (uint8) from rank `src`.  `arena` is a numpy array on `src`, None elsewhere.
    With `device` (e.g. "cuda:3") the broadcast runs on GPU memory over RCCL and the returned tensor can be handed to
    wis_model_create(arena_on_device=1) by pointer; without it, it is a host tensor (gloo)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device or "cpu")
    if rank == src:
        assert arena is not None and arena.nbytes == nbytes
        buf.copy_(torch.from_numpy(np.ascontiguousarray(arena).view(np.uint8).reshape(-1)))
    dist.broadcast(buf, src=src, group=group)
    return buf


def sharded_map(fn, items, group=None):
    """Run `fn(list_of_items) -> list_of_results` on this rank's shard; rank 0 returns the results of ALL items in input
    order (other ranks return None).  `items` must be the same list on every rank (or at least have the same length)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(len(items), world, rank)
    local = fn(items[lo:hi]) if hi > lo else []
    assert len(local) == hi - lo
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((lo, local), gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = [None] * len(items)
    for lo_r, res in gathered:
        out[lo_r:lo_r + len(res)] = res
    return out
