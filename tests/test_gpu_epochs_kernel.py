"""eskf_epochs_kernel (the long-horizon kernel: a run of epochs with any number of predicts per update in ONE launch --
the time loop of Filter.run and the predicts of propagate_imu, Filter.py:144-217, in the kernel; recurrences and Jacobian
blocks produced by wavefronts that run ahead of the covariance wavefronts, covariances in registers for the whole run)
against the path it replaces, the same epochs enqueued one by one (eskf_nominal_kernel + eskf_jac_kernel +
eskf_cov_rows_kernel).  Both sides evaluate the same expressions in the same order, so the results must be the same bits:
state, covariance, buffers, status words -- for every block size of the new kernel, ragged batches and schedules, both
precisions, shared and per-instance streams, runs started from another handle and runs continued by a second call."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(batch, dtype, sched, nc="", frozen="000000", hold="1", timeout=600, pp=""):
    env = dict(os.environ)
    for k in ("DVI_ESKF_PRE_LEGACY", "DVI_ESKF_FUSED", "DVI_ESKF_FUSED_ROUNDS", "DVI_ESKF_RUN", "DVI_ESKF_RUN_ROUNDS", "DVI_ESKF_EPOCHS",
              "DVI_ESKF_EPOCHS_NC", "DVI_ESKF_EPOCHS_PP", "DVI_ESKF_EPOCH_STREAMS"):
        env.pop(k, None)
    if nc:
        env["DVI_ESKF_EPOCHS_NC"] = nc
    if pp:
        env["DVI_ESKF_EPOCHS_PP"] = pp
    env["DVI_ESKF_EPOCHS"] = "1"        # (also where the launcher would leave a batch of this size to the epoch-by-epoch path)
    env["DVI_ESKF_VERBOSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_epochs_child.py"), str(batch), dtype, ROOT, sched, frozen, hold], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "eskf_epochs_kernel" in p.stderr, p.stderr[-2000:]


@pytest.mark.parametrize("batch,dtype,nc", [(5, "f64", ""), (5, "f32", ""), (1000, "f64", ""), (1000, "f32", ""), (1000, "f64", "2"), (1000, "f64", "12"),
                                            (1000, "f32", "8"), (333, "f64", "4"), (700, "f32", "10"), (8192, "f64", ""), (8192, "f32", "")])
def test_run_of_epochs_in_one_launch_equals_the_epochs_one_by_one(batch, dtype, nc):
    """the ragged 33 / 33 / 34 schedule of BASELINE config #5 (three camera frames), every block size"""
    _child(batch, dtype, "33,33,34", nc)


@pytest.mark.parametrize("batch,nc", [(5, "8"), (1000, "8"), (700, "10"), (1500, "12"), (8192, "")])
def test_two_pairs_of_instances_per_covariance_wavefront(batch, nc):
    """fp32: a covariance wavefront keeps the rows of four instances (32 / 40 / 48 instances per block), ragged blocks included"""
    _child(batch, "f32", "33,33,34", nc, pp="2")
    if batch == 1000:
        _child(batch, "f32", "2,7,0,3,12,5", nc, pp="2")
        _child(515, "f32", "4,3,5", nc, frozen="101001", hold="0", pp="2")


@pytest.mark.parametrize("batch,dtype,sched", [(777, "f64", "10,10,10,10,10,10"), (130, "f32", "2,7,3,12,5"), (64, "f64", "2,2,2,2,2,2,2,2,2")])
def test_other_schedules(batch, dtype, sched):
    _child(batch, dtype, sched)


def test_frozen_dofs_zeroed_by_every_predict():
    """the v1-literal plumbing (Filter.py:243-245): a mixed frozen mask, hold_frozen = 0"""
    _child(515, "f64", "4,3,5", frozen="101001", hold="0")
    _child(515, "f64", "4,3,5", frozen="111111", hold="1")


def test_epochs_of_one_and_of_no_predict_in_the_schedule():
    """epochs with a single predict take the four-chain pre kernel on the epoch-by-epoch side (a different rounding
    sequence in the last bits): closeness instead of equality; an epoch without a predict is an update on its own"""
    # (frozen extrinsics: the unfrozen Monte-Carlo instances amplify a last-bit difference by orders of magnitude within a few
    #  updates, DESIGN.md section 5 -- a closeness bound says nothing there)
    _child(300, "f64", "3,1,0,4,1", frozen="111111")
