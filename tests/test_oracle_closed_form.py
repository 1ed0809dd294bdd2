"""The hand-derived closed forms that the HIP kernels implement (SURVEY.md App.
A.3 / B, oracle/closed_form.py) agree with the literal restatement that
differentiates the reference's own expressions (oracle/eskf_ref.py), and the
product's host-side probe model agrees with the DH chain."""
import math

import numpy as np

from dvi_ekf_amd.probe import probe_pose
from oracle.closed_form import error_jacobians, probe_closed_form
from oracle.eskf_ref import RefFilter, RefParams, probe_chain, quat_nrm, quat_to_mat


def test_probe_closed_form_matches_dh_chain():
    rng = np.random.default_rng(0)
    L, th = 50.0, math.radians(30)
    for _ in range(100):
        dofs = np.concatenate([rng.normal(0, 0.7, 3), rng.normal(0, 10, 3)])
        q7 = rng.normal(0, 1.0)
        p, R, z6 = probe_chain(np.concatenate([dofs, [q7]]), L, th)
        k = probe_closed_form(dofs, q7, L, th)
        p2, R2, z2 = probe_pose([*dofs, q7], L, th)
        for a, b in ((p, k["p"]), (R, k["R"]), (z6, k["z6"]), (p, p2), (R, R2), (z6, z2)):
            assert np.abs(a - b).max() < 1e-13
    # the reference's ground-truth configuration: p_CB=(0,-50,20), R_BC=Rx(120deg)
    p, R, _ = probe_pose([0, 0, 0, 0, 0, 20, 0], L, th)
    assert np.allclose(p, [0, -50, 20], atol=1e-13)
    c, s = math.cos(math.radians(120)), math.sin(math.radians(120))
    assert np.allclose(R, [[1, 0, 0], [0, c, -s], [0, s, c]], atol=1e-13)


def test_error_jacobians_match_differentiated_reference_expressions():
    rng = np.random.default_rng(1)
    L, th = 50.0, math.radians(30)
    for _ in range(50):
        dofs = np.concatenate([rng.normal(0, 0.5, 3), rng.normal(0, 10, 3)])
        notch = rng.normal(0, 0.5, 3)
        x = np.zeros(26)
        x[6:10] = quat_nrm(rng.normal(size=4))
        x[10:16], x[16:19] = dofs, notch
        x[22:26] = quat_nrm(rng.normal(size=4))
        prm = RefParams(frozen=(0,) * 6)
        f = RefFilter(prm, x, np.eye(24), rng.normal(0, 0.05, 3), rng.normal(0, 1, 3))
        f.R_WB_old = quat_to_mat(quat_nrm(rng.normal(size=4)))
        dt = rng.uniform(1e-3, 1.0)
        f._predict_error(dt)
        Om = quat_nrm(np.array([*(0.5 * dt * f.om_old), 1.0]))
        Fx, Fi = error_jacobians(dt, dofs, notch, f.R_WB_old, f.om_old, f.acc_old, prm.sigma_om, L, th,
                                 quat_to_mat(Om).T)
        scale = max(1.0, np.abs(f.Fx).max())
        assert np.abs(Fx - f.Fx).max() < 1e-13 * scale
        assert np.abs(Fi - f.Fi).max() < 1e-13 * scale
