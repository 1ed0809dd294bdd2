"""Multi-GPU plumbing: one process per GPU, independent filter instances sharded in
contiguous blocks, and ONE collective -- the final all-gather of the estimated calibration
parameters (``state.dofs``), over RCCL (``backend="nccl"`` on ROCm) / xGMI.

The reference has no multi-process code at all (Simulator.run loops num_kf_runs
sequentially, Simulator.py:138-153); the batch axis replaces that loop.  There is no
exchange between instances during filtering, so no other collective exists on the path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, start+count) of `total` instances owned by `rank`; blocks
    differ by at most one instance."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def all_gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenates every rank's [n_r, ...] rows in rank order.  Shards may differ in length
    (ragged tail): lengths are exchanged first and shorter shards are padded for the gather."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax - local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    bufs = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(bufs, local.contiguous(), group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def gather_calibration(filt, group=None) -> torch.Tensor:
    """Calibration estimates of the whole job: [B_total, 6] on every rank."""
    return all_gather_rows(filt.calibration(), group=group)


def dof_mse(dofs: torch.Tensor, gt) -> torch.Tensor:
    """Filter.calculate_dof_metric (Filter.py:452-455) per instance: mean((dofs - gt)^2)."""
    d = dofs - torch.as_tensor(gt, dtype=dofs.dtype, device=dofs.device)
    return (d * d).mean(dim=1)
