"""The N>1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Covers the
instance sharding and the single collective of the path (all-gather of the calibration
estimates), including a ragged split."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dvi_ekf_amd.dist import all_gather_rows, dof_mse, shard_range


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8192, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, count = shard_range(total, rank, world)
        full = torch.arange(total * 6, dtype=torch.float64).reshape(total, 6)
        local = full[start:start + count].clone()           # this rank's calibration estimates
        got = all_gather_rows(local)
        assert got.shape == (total, 6) and torch.equal(got, full)
        mse = dof_mse(got, torch.zeros(6, dtype=torch.float64))
        assert torch.allclose(mse, (full ** 2).mean(dim=1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [16, 17])
def test_all_gather_of_calibration_estimates_world2(total):
    mp.spawn(_worker, args=(2, _free_port(), total), nprocs=2, join=True)
