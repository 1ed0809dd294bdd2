"""eskf_pre4_kernel (the four-chain pre kernel of single-predict launches) and eskf_nominal_kernel + eskf_jac_kernel (the
pre work of an epoch: recurrences, then the Jacobian blocks of all predicts at once) against the two-role pre kernel
they replace.  Both evaluate the same formulas with the same helper functions; what differs is
where the compiler contracts a*b+c into an fma (the chains are inlined into different surroundings), i.e. the
last bit of some intermediate.  Stated tolerance, relative to the instance's magnitude: a single predict --
state, covariance and the dense Fx / Fi -- 1e-13; the same after an update has been injected on the fly; a
three-step sequence (the updates amplify a last-bit difference by the conditioning of S, as they do for any two
fp64 implementations: DESIGN.md section 5) 1e-8, with identical status words.  Covered: random configurations,
pending injections, zeroed frozen DOFs, non-finite measurements, fp32 and full Jacobian records.  The two kernels
are selected per process (DVI_ESKF_PRE_LEGACY), so each side runs in a child."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, scenario, legacy):
    out = tmp_path / f"{scenario}_{'legacy' if legacy else 'pre4'}.npz"
    env = dict(os.environ)
    env.pop("DVI_ESKF_PRE_LEGACY", None)
    env["DVI_ESKF_FUSED"] = "0"     # (this file is about the pre kernels: the one-launch step has tests/test_gpu_step_kernel.py)
    if legacy:
        env["DVI_ESKF_PRE_LEGACY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_pre4_child.py"), scenario, str(out), ROOT], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return np.load(out)


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    B = a.shape[0]
    a, b = a.reshape(B, -1), b.reshape(B, -1)
    ok = np.isfinite(b).all(axis=1)
    assert np.array_equal(ok, np.isfinite(a).all(axis=1))
    return float((np.abs(a[ok] - b[ok]).max(axis=1) / np.maximum(1.0, np.abs(b[ok]).max(axis=1))).max())


@pytest.mark.parametrize("scenario", ["unfrozen", "mixed_zeroed", "frozen_hold", "keep_jac", "fp32"])
def test_pre4_agrees_with_two_role_pre_kernel(tmp_path, scenario):
    a, b = _run(tmp_path, scenario, True), _run(tmp_path, scenario, False)
    assert set(a.files) == set(b.files)
    eps = 2.0 ** -23 / 2.0 ** -52 if scenario == "fp32" else 1.0
    worst = {}
    for key in a.files:
        if key.startswith("st") or key.endswith("_st"):
            assert np.array_equal(a[key], b[key]), key
            continue
        tol = 1e-13 if key.startswith("p_") else 1e-12 if key.startswith("pu_") else 1e-8
        worst[key] = _rel(a[key], b[key])
        assert worst[key] <= tol * eps, (key, worst[key])
    print(scenario, {k: f"{v:.1e}" for k, v in worst.items()})
    assert (a["st1"][[5, 7, 9]] & 2).all()          # the non-finite measurements were flagged (on both sides)

