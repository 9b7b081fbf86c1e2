"""Multi-GPU data parallelism for the forward path: images shard by batch, weights are
replicated, and the ONLY collective is one all-gather of the class probabilities.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
on CPU for the tests).  The reference has no multi-device code at all;
this is the north_star's added capability.  [n_local, num_classes] f32 per rank is
1 MB at 256 images -- latency-bound, so one all_gather_into_tensor per step is the whole
communication schedule.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first n_total % world ranks take one extra image."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_probs(local_probs, n_total: int, group=None):
    """All-gather the per-rank [n_local, C] probability blocks into [n_total, C] in global image order.
    Ragged shards (n_total % world != 0) are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    C = local_probs.shape[1]
    n_max = (n_total + world - 1) // world
    lo, hi = shard_bounds(n_total, world, rank)
    assert local_probs.shape[0] == hi - lo, "local block does not match this rank's shard"
    if n_total % world == 0:
        out = torch.empty((n_total, C), dtype=local_probs.dtype, device=local_probs.device)
        dist.all_gather_into_tensor(out, local_probs.contiguous(), group=group)
        return out
    padded = torch.zeros((n_max, C), dtype=local_probs.dtype, device=local_probs.device)
    padded[: hi - lo] = local_probs
    buf = torch.empty((world * n_max, C), dtype=local_probs.dtype, device=local_probs.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_total, world, r)
        parts.append(buf[r * n_max: r * n_max + (b - a)])
    return torch.cat(parts, 0)


def predict_sharded(forward_local: Callable, images, group=None):
    """images: [n_total, S, S, 3] (same on every rank, or only this rank's rows are read).
    forward_local(block) -> [n_local, C] probabilities for this rank's block (the HIP engine on GPU)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = images.shape[0]
    lo, hi = shard_bounds(n_total, world, rank)
    return gather_probs(forward_local(images[lo:hi]), n_total, group)


def gather_probs_async(local_probs, n_total: int, group=None):
    """The same collective without making the caller's stream wait for it: returns (gathered, work).  The collective is ordered after
    everything already enqueued on the current stream (it reads `local_probs`), the caller's stream goes on -- e.g. with the next batch's
    forward into ANOTHER probability buffer -- and `work.wait()` orders the current stream behind the collective when the result (or the
    right to overwrite `local_probs`) is needed.  Ragged shards take the synchronous path (work = None)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if n_total % world:
        return gather_probs(local_probs, n_total, group), None
    out = torch.empty((n_total, local_probs.shape[1]), dtype=local_probs.dtype, device=local_probs.device)
    work = dist.all_gather_into_tensor(out, local_probs.contiguous(), group=group, async_op=True)
    return out, work

