"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm,
"gloo" for the CPU tests).

* Independent (scalar, point) batches (BASELINE configs 2, 3, 5) shard by contiguous index ranges
  with no data-path collective: `shard_range`.
* One large MSM (config 4) is linear in its terms: every rank sums its slice, the ranks exchange
  ONE projective point each (all-gather of 3*NB bytes; elliptic-curve addition is not an RCCL
  reduction operator, so all-reduce cannot be used) and add the partial sums locally with the
  complete addition.  Payload is 96 bytes per rank: latency-bound, xGMI bandwidth is irrelevant.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n units owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_points(local_xyz: np.ndarray, group=None) -> np.ndarray:
    """All-gather one projective point (uint8[3*NB]) per rank -> (world, 3*NB)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.from_numpy(np.ascontiguousarray(local_xyz, dtype=np.uint8)).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return np.stack([o.cpu().numpy() for o in out])


def allgather_into(all_parts, part, backend: str, group=None, synchronize=None):
    """All-gather one projective point per rank into `all_parts` ((world, 3*NB) uint8 tensor) from `part` ((3*NB,) uint8).

    backend "nccl" (RCCL): both tensors live in HBM, the collective is ordered on torch's current stream - no host
    round trip.  Any other backend (gloo: CPU tests, several ranks rehearsed on one card) moves the 96 bytes through
    host tensors; `synchronize` (e.g. Context.synchronize) is called first so that the partial sum is complete."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if backend == "nccl":
        dist.all_gather_into_tensor(all_parts, part, group=group)
        return all_parts
    if synchronize is not None:
        synchronize()
    h = part.cpu()
    parts = [torch.empty_like(h) for _ in range(world)]
    dist.all_gather(parts, h, group=group)
    all_parts.copy_(torch.stack(parts).to(all_parts.device))
    return all_parts


def fold_points_device(cv, d_points_xyz, count: int, d_scratch_xyz, d_out_xy, d_out_inf):
    """Sum `count` device-resident projective points (the all-gathered partial sums of a split MSM) by a tree of complete
    additions and normalise the result: ceil(log2 count) + 1 small launches, no scalar multiplication (an MSM with unit
    scalars would spend a whole scalar multiplication's latency, ~1 ms, on it).

    d_points_xyz, d_scratch_xyz: (>= count, 3*NB) uint8 device tensors (both are overwritten); d_out_xy (2*NB,), d_out_inf (1,)."""
    # Only library calls touch the buffers (all on the context's stream): a tree over the even part of every level, the
    # odd row of a level is left where it is - later levels write below it - and added at the end.
    cur, other = d_points_xyz, d_scratch_xyz
    leftovers = []
    while count > 1:
        half = count // 2
        if count & 1:
            leftovers.append(cur[count - 1:count])
        cv.add_device(cur[:half], cur[half:2 * half], other[:half], half)
        count = half
        cur, other = other, cur
    for row in leftovers:
        cv.add_device(cur[:1], row, other[:1], 1)
        cur, other = other, cur
    cv.batch_normalize_device(cur[:1], d_out_xy, d_out_inf, 1)


def msm_sharded(local_msm: Callable[[int, int], np.ndarray], add_points: Callable[[np.ndarray, np.ndarray], np.ndarray],
                n: int, group=None) -> np.ndarray:
    """sum_i k_i P_i over n terms split across the ranks of `group`.

    local_msm(lo, hi) -> projective X||Y||Z bytes of the partial sum over terms [lo, hi)
    add_points(p, q)  -> projective sum (complete addition, e.g. Curve.add)
    Every rank returns the same projective point (the fold order is fixed: rank 0, 1, 2, ...)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n, rank, world)
    part = local_msm(lo, hi)
    parts = allgather_points(part, group)
    acc = parts[0]
    for r in range(1, world):
        acc = add_points(acc, parts[r])
    return acc
