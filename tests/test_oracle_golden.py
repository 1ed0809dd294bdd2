"""Pins the oracle (NumPy twin and C restatement) against the reference's own
known-answer data: data/trajs/kf_best_mandala0_mono.txt and
imu_ref_mandala0_mono.txt, then cross-checks the two
restatements on a longer, unfrozen run and the covariance checkpoints of
SURVEY.md App. A.5."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle.eskf_ref import RefFilter, RefParams
from tests.helpers import IMU_REF, default_run, load_golden, log_row
from dvi_ekf_amd import rotmath as rm

PRINT_ROUNDING = 5.0e-10 + 1e-12   # the file stores 9 decimals


def _ref_filter(cfg, ri, Q, R, sa, so, hold=True):
    prm = RefParams(L=cfg.model.length, theta_cam=cfg.model.angle, frozen=tuple(cfg.frozen_dofs),
                    hold_frozen=hold, sigma_a=sa, sigma_om=so, Q=Q, R=R)
    return RefFilter(prm, ri.x0, ri.P0, ri.om0, ri.acc0)


def _c_filter(cfg, ri, Q, R, sa, so, hold=True, B=1):
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, hold, sa, so, Q, R)
    return c_oracle.OracleBatch(prm, np.tile(ri.x0, (B, 1)), ri.P0, ri.om0, ri.acc0)


def _run(step_predict, step_update, get_x, ri):
    rows = [log_row(ri.t0, get_x())]
    ip = 0
    for k in range(len(ri.t_meas)):
        for _ in range(ri.n_pred[k]):
            step_predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            ip += 1
        step_update(ri.pos[k], ri.quat[k], ri.notch[k])
        rows.append(log_row(ri.t_meas[k], get_x()))
    return np.array(rows)


def test_numpy_twin_reproduces_saved_trajectory():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _ref_filter(cfg, ri, Q, R, sa, so)
    rows = _run(f.predict, f.update, lambda: f.x, ri)
    gold = load_golden()
    assert rows.shape == gold.shape == (10, 30)
    assert np.abs(rows - gold).max() <= PRINT_ROUNDING


def test_c_oracle_reproduces_saved_trajectory():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _c_filter(cfg, ri, Q, R, sa, so)
    rows = _run(f.predict, f.update, lambda: f.x[0], ri)
    assert np.abs(rows - load_golden()).max() <= PRINT_ROUNDING
    assert (f.status == 0).all()


def test_imu_reference_trajectory_reproduced():
    _, ri, *_ = default_run(golden_compat=True)
    ref = np.loadtxt(IMU_REF)
    mine = np.array([[ri.t_imu[i], *ri.imu_ref[i, 0:3], *ri.imu_ref_v[i],
                      *np.degrees(rm.euler_xyz_extrinsic(ri.imu_ref[i, 3:7])), ri.imu_ref[i, 6], *ri.imu_ref[i, 3:6]]
                     for i in range(len(ri.t_imu))])
    assert mine.shape == ref.shape == (9, 14)
    assert np.abs(mine - ref).max() <= PRINT_ROUNDING


# SURVEY.md App. A.5 covariance checkpoints (pair-step k, stage) -> trace P, |P|_F, P[18,18], P[6,6], |K|_F
CHECKPOINTS = {
    (1, "predict"): (6.462716930457e+02, 3.609548690352e+02, 1.229782759626e+02, 3.046174197867e-04, None),
    (1, "update"): (3.762194388442e+01, 2.647976243220e+01, 3.999986981134e-04, 3.024290979091e-04, 5.976400708597e+01),
    (2, "predict"): (5.194887382974e+01, 2.961071238198e+01, 1.300912202313e+01, 3.024291006304e-04, None),
    (2, "update"): (3.253782463268e+01, 2.313687652709e+01, 3.999783822644e-04, 3.024199509430e-04, 8.351057240771e+03),
    (5, "update"): (2.631217023199e+01, 1.811089165643e+01, 3.999475958412e-04, 3.022297060144e-04, 1.860847054137e+04),
    (9, "predict"): (3.468644001108e+01, 1.967501538172e+01, 1.300882012065e+01, 3.019611765179e-04, None),
    (9, "update"): (1.838932595446e+01, 1.423250120471e+01, 3.999709484325e-04, 3.018681237562e-04, 8.215363609041e+04),
}


def test_c_oracle_covariance_checkpoints():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _c_filter(cfg, ri, Q, R, sa, so)
    for k in range(len(ri.t_meas)):
        f.predict_one(0, ri.dt[k], ri.om[k], ri.acc[k])
        for stage in ("predict", "update"):
            if stage == "update":
                _, K = f.update_one(0, ri.pos[k], ri.quat[k], ri.notch[k])
            if (k + 1, stage) in CHECKPOINTS:
                tr, fro, p18, p6, kf = CHECKPOINTS[(k + 1, stage)]
                P = f.P[0]
                got = (np.trace(P), np.linalg.norm(P), P[18, 18], P[6, 6])
                np.testing.assert_allclose(got, (tr, fro, p18, p6), rtol=2e-9)
                if kf is not None:
                    np.testing.assert_allclose(np.linalg.norm(K), kf, rtol=2e-8)


@pytest.mark.parametrize("frozen", [(1, 1, 1, 1, 1, 1), (0, 0, 0, 0, 0, 0), (1, 0, 1, 0, 0, 1)])
def test_c_oracle_matches_numpy_twin_v1_literal(frozen):
    """v1-literal plumbing (10 predicts per frame, frozen DOFs zeroed), 20 frames:
    state, covariance, Jacobians and gain of the two restatements agree."""
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=frozen)
    fn = _ref_filter(cfg, ri, Q, R, sa, so, hold=False)
    fc = _c_filter(cfg, ri, Q, R, sa, so, hold=False)
    ip = 0
    for k in range(len(ri.t_meas)):
        for _ in range(ri.n_pred[k]):
            fn.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            Fx, Fi = fc.predict_one(0, ri.dt[ip], ri.om[ip], ri.acc[ip])
            ip += 1
            assert np.abs(Fx - fn.Fx).max() < 1e-11
            assert np.abs(Fi - fn.Fi).max() < 1e-11
        Kn = fn.update(ri.pos[k], ri.quat[k], ri.notch[k])
        st, Kc = fc.update_one(0, ri.pos[k], ri.quat[k], ri.notch[k])
        assert st == 0
        assert np.abs(Kc - Kn).max() <= 1e-9 * max(1.0, np.abs(Kn).max())
        assert np.abs(fc.x[0] - fn.x).max() < 1e-9
        assert np.abs(fc.P[0] - fn.P).max() < 1e-9 * max(1.0, np.abs(fn.P).max())
    assert ip == 90


def test_c_oracle_singular_innovation_leaves_state_untouched():
    """Filter.py:356-361: singular S -> update returns None and nothing changes."""
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _c_filter(cfg, ri, Q, np.zeros((7, 7)), sa, so)
    f.P[0][:] = 0.0
    x_before = f.x.copy()
    st = f.update(ri.pos[0], ri.quat[0], ri.notch[0])
    assert st[0] & 1
    assert np.array_equal(f.x, x_before) and not f.P.any()
