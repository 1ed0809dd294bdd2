"""eskf_step_kernel (one launch per predict(+update) step, the Jacobian records handed from the block's pre phase to its
covariance rounds on chip) against the two kernels it replaces, eskf_pre4_kernel + eskf_cov_rows_kernel.  Both paths
run the SAME device functions (pre4_body, cov_rows_body) on the same inputs in the same order; only the way the 75-word
record travels differs (LDS staging rows -> registers instead of memory).  The results must therefore be the same bits:
state, covariance, status words, through pending injections, zeroed frozen DOFs, non-finite measurements, ragged
batches, fp32, and for every block size of the one-launch kernel (16 / 32 / 48 / 64 instances per block).  The path is
selected per process (DVI_ESKF_FUSED, DVI_ESKF_FUSED_ROUNDS), so each side runs in a child (tests/_pre4_child.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, scenario, tag, **env_over):
    out = tmp_path / f"{scenario}_{tag}.npz"
    env = dict(os.environ)
    for k in ("DVI_ESKF_PRE_LEGACY", "DVI_ESKF_FUSED", "DVI_ESKF_FUSED_ROUNDS"):
        env.pop(k, None)
    env.update(env_over)
    env["DVI_ESKF_VERBOSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_pre4_child.py"), scenario, str(out), ROOT], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert ("eskf_step_kernel" in p.stderr) == (env_over.get("DVI_ESKF_FUSED") == "1"), p.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("scenario", ["unfrozen", "mixed_zeroed", "frozen_hold", "fp32", "pin"])
def test_one_launch_step_equals_the_two_kernels_bit_for_bit(tmp_path, scenario):
    ref = _run(tmp_path, scenario, "split", DVI_ESKF_FUSED="0")
    # (per-instance noise records -- "pin" -- are served by the 16-instance blocks only)
    for rounds in (("1",) if scenario == "pin" else ("4", "3", "2", "1")):
        got = _run(tmp_path, scenario, "fused" + rounds, DVI_ESKF_FUSED="1", DVI_ESKF_FUSED_ROUNDS=rounds)
        assert set(ref.files) == set(got.files)
        bad = [k for k in ref.files if not np.array_equal(ref[k], got[k], equal_nan=True)]
        assert not bad, (scenario, rounds, bad)
    assert (ref["st1"][[5, 7, 9]] & 2).all()        # the non-finite measurements were flagged (on both sides)


@pytest.mark.parametrize("batch,dtype,rounds", [(777, "f64", "1"), (777, "f64", "4"), (1500, "f64", "2"), (1000, "f64", "3"), (515, "f32", "4"),
                                                (2048, "f64", ""), (5, "f64", ""), (5, "f32", "4"), (300, "f64pin", "")])
def test_run_in_one_launch_equals_the_steps_one_by_one(batch, dtype, rounds, steps=6):
    """dvi_eskf_run_epochs with one predict per epoch is one launch of eskf_run_kernel (the time loop in the kernel):
    bit for bit the step-by-step run, also started from another handle's state, continued, and with shared streams.
    (Between the steps of a run the covariances of a block's last round stay in LDS and the order of the rounds
    alternates: every block size -- a quarter, a third, half, all of them resident -- and ragged batches.)"""
    env = dict(os.environ)
    for k in ("DVI_ESKF_PRE_LEGACY", "DVI_ESKF_FUSED", "DVI_ESKF_FUSED_ROUNDS", "DVI_ESKF_RUN", "DVI_ESKF_RUN_ROUNDS"):
        env.pop(k, None)
    if rounds:
        env["DVI_ESKF_RUN_ROUNDS"] = rounds
    env["DVI_ESKF_VERBOSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_run_child.py"), str(batch), dtype, ROOT, str(steps)], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "eskf_run_kernel" in p.stderr, p.stderr[-2000:]


@pytest.mark.parametrize("batch,rounds,steps", [(777, "4", 17), (1000, "3", 16), (130, "2", 9)])
def test_longer_runs_in_one_launch(batch, rounds, steps):
    """the same comparison over more steps than a block has rounds, even and odd step counts"""
    test_run_in_one_launch_equals_the_steps_one_by_one(batch, "f64", rounds, steps)
