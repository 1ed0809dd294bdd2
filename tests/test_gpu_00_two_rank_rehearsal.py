"""The N > 1 path of bench.py on the one GPU of the test box: two ranks launched by torch.distributed.run
exactly as the driver launches them (one process per rank), sharing the GPU, collective over gloo -- and the
calibration estimates they gather must equal the unsharded single-process run bit for bit (strong scaling
shards ONE batch: rank r owns instances [r B/N, (r+1) B/N), and the counter-based noise generator gives a
global instance the numbers it has in the unsharded run).

This file sorts in front of the other GPU tests on purpose: the launches below are made from a process that
has not touched the GPU itself (tests/conftest.py counts devices without initialising the runtime)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(tmp_path, nranks, name, backend="gloo", force_dist=False, self_launch=False):
    out = tmp_path / f"{name}.npy"
    args = ["--gpus", str(nranks), "--steps", "24", "--warmup", "0", "--batch", "2048", "--strong", "--horizon", "8",
            "--no-cpu-baseline", "--no-secondary", "--dump-dofs", str(out)]
    env = dict(os.environ, DVI_BENCH_BACKEND=backend, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force_dist:
        env["DVI_BENCH_FORCE_DIST"] = "1"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if (nranks == 1 and not force_dist) or self_launch:
        # (self_launch: `python bench.py --gpus N` from a plain shell -- the script starts the launcher itself)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), np.load(out)


def test_two_ranks_gather_what_one_process_computes(tmp_path):
    one, dofs1 = _bench(tmp_path, 1, "one")
    two, dofs2 = _bench(tmp_path, 2, "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong" and one["valid"] and two["valid"]
    assert two["config"]["global_batch"] == one["config"]["global_batch"] == 2048 and two["config"]["batch_per_gpu"] == 1024
    assert dofs1.shape == dofs2.shape == (2048, 6)
    assert np.array_equal(dofs1, dofs2)                     # bit for bit
    assert np.isfinite(dofs1).all() and len(np.unique(dofs1[:, 0])) > 1000


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 --strong ...` with no launcher around it (the shape of the driver's
    N = 1 command with another N) starts one process per rank itself and prints the one JSON line."""
    one, dofs1 = _bench(tmp_path, 1, "one_s")
    two, dofs2 = _bench(tmp_path, 2, "two_s", self_launch=True)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["valid"]
    assert np.array_equal(dofs1, dofs2)


def test_rccl_code_path_with_one_rank(tmp_path):
    """The RCCL calls of the N > 1 path (communicator over the `nccl` backend, device barriers, all-gather of the
    calibration estimates, MAX over ranks) on the one GPU there is: a one-rank job launched by torch.distributed.run
    that is forced through the multi-rank code path must produce what the plain single-process run produces."""
    one, dofs1 = _bench(tmp_path, 1, "plain")
    rccl, dofs2 = _bench(tmp_path, 1, "rccl", backend="nccl", force_dist=True)
    assert rccl["valid"] and rccl["n_gpus"] == 1
    assert "RCCL" in rccl["config"]["final_collective"]
    assert np.array_equal(dofs1, dofs2)
