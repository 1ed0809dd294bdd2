"""Monte-Carlo calibration run (BASELINE config #4): B independent filters with randomised
IMU / camera noise and perturbed initial extrinsic DOFs, sharded over the GPUs of one node
(one process per GPU), followed by ONE collective -- the RCCL all-gather of the estimated
calibration parameters -- and the DOF-MSE statistic of the reference
(Filter.calculate_dof_metric, Filter.py:452-455).

    python -m dvi_ekf_amd.montecarlo --batch 65536 --frames 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m dvi_ekf_amd.montecarlo --batch 65536 --frames 8
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np
import torch

from .dist import dof_mse, gather_calibration, shard_range
from .workloads import monte_carlo_pairs


def run_monte_carlo(batch_total: int, frames: int = 8, seed: int = 4, rank: int = 0, world: int = 1,
                    device=None, group=None) -> dict:
    """Runs this rank's shard and returns job-wide statistics (identical on every rank)."""
    device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    start, count = shard_range(batch_total, rank, world)
    wl = monte_carlo_pairs(count, horizon=frames, seed=seed, device=device, first_instance=start)
    f = wl.make_filter()
    # the whole run is ONE call: one IMU sample per camera frame, so the time loop of Filter.run runs inside the kernel
    # (dvi_eskf_run_epochs -> eskf_run_kernel); same bits as a step call per frame
    status, _ = f.run_epochs(np.ones(frames, dtype=np.int32), wl.dt, wl.om, wl.acc, wl.pos, wl.quat, wl.notch)
    worst = torch.zeros(count, dtype=torch.int32, device=device)
    for k in range(frames):
        worst |= status[k]
    dofs = gather_calibration(f, group=group)                    # [batch_total, 6] on every rank
    flags = worst.to(torch.float64).reshape(-1, 1)
    from .dist import all_gather_rows
    flags = all_gather_rows(flags, group=group).reshape(-1)
    gt = np.asarray(wl.cfg.gt_imu_dofs, dtype=float)
    mse = dof_mse(dofs, gt)
    finite = torch.isfinite(mse)
    good = finite & (flags == 0)
    return {
        "batch": int(batch_total), "frames": int(frames), "world": int(world),
        "instances_ok": int(good.sum()), "instances_flagged": int((flags != 0).sum()),
        "dof_mse_mean": float(mse[good].mean()), "dof_mse_std": float(mse[good].std()),
        "dof_mse_median": float(mse[good].median()),
        "reference_instance_dofs": [float(v) for v in dofs[0].tolist()],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--seed", type=int, default=4)
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    out = run_monte_carlo(a.batch, a.frames, a.seed, rank, world, torch.device("cuda", local))
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
