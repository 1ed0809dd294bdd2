"""Device-side quirk flags (dvi_eskf_params.quirks_fixed, include/dvi_eskf.h): Q1, Q3, Q6, Q9 of SURVEY.md App. A.6 can be
switched off per handle.  There is no reference artefact for a "fixed" filter -- the reference itself never runs one --, so
each variant is checked against the NumPy restatement (oracle/eskf_ref.py) with the SAME quirk switched off, forced single
steps: every predict and every update starts from the restatement's state on both sides (fp64: state 1e-11, covariance 1e-10,
relative to the instance's magnitude).  The default handle (all quirks in force) is what every other test of the suite covers."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import eskf_ref as ref
from tests.helpers import default_run

FIX = {"Q1_jac_col_shift": 1, "Q3_dqc_axis_theta": 2, "Q6_half_angle_residual": 4, "Q9_Rc_uses_om_old": 8}
TOL_X, TOL_P = 1e-11, 1e-10


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def _twins(cfg, ri, Q, R, sa, so, x0, off):
    prm = ref.RefParams(L=cfg.model.length, theta_cam=cfg.model.angle, frozen=tuple(cfg.frozen_dofs), hold_frozen=True,
                        sigma_a=np.asarray(sa, dtype=float), sigma_om=np.asarray(so, dtype=float), Q=np.asarray(Q, dtype=float),
                        R=np.asarray(R, dtype=float), quirks=ref.V1 - frozenset(off))
    return [ref.RefFilter(prm, x0[b], ri.P0, ri.om0, ri.acc0) for b in range(x0.shape[0])]


@pytest.mark.parametrize("off", [("Q1_jac_col_shift",), ("Q3_dqc_axis_theta",), ("Q6_half_angle_residual",), ("Q9_Rc_uses_om_old",),
                                 ("Q1_jac_col_shift", "Q3_dqc_axis_theta", "Q6_half_angle_residual", "Q9_Rc_uses_om_old")])
def test_forced_steps_with_a_quirk_switched_off(off):
    from dvi_ekf_amd.batch import BatchedESKF
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=False, frozen=(0,) * 6, interframe=1)
    # (a measurement noise that leaves the innovation covariance well conditioned: with the YAML's camera attitude noise,
    #  1e-5 deg, the restatement's explicit inverse of S -- the reference's np.linalg.inv -- is itself only good to
    #  cond(S) eps ~ 1e-6, which the 3 deg residual below would multiply into every number compared here)
    R = np.diag([1e-2] * 3 + [1e-4] * 3 + [1e-4])
    B, frames = 4, 6
    rng = np.random.default_rng(11)
    x0 = np.tile(ri.x0, (B, 1))
    x0[1:, 10:13] += rng.normal(0, np.radians(5.0), (B - 1, 3))
    x0[1:, 13:16] += rng.normal(0, 10.0, (B - 1, 3))
    x0[:, 16] = rng.normal(0, 0.05, B)                  # a notch angle and rate: phi moves with every predict
    x0[:, 17] = rng.normal(0, 0.02, B)
    fixed = sum(FIX[q] for q in off)
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa, sigma_om=so,
                    hold_frozen=True, quirks_fixed=fixed)
    f.set_noise(Q, R)
    v1 = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa, sigma_om=so, hold_frozen=True)
    v1.set_noise(Q, R)
    tw = _twins(cfg, ri, Q, R, sa, so, x0, off)
    q_old = x0[:, 6:10].copy()
    ex = eP = moved = 0.0
    for k in range(frames):
        # ---- predict, forced from the restatement's state
        X = np.array([t.x for t in tw]); P = np.array([t.P for t in tw])
        om_o = np.array([t.om_old for t in tw]); acc_o = np.array([t.acc_old for t in tw])
        for h in (f, v1):
            h.set_state(x=X, P=P, om_old=om_o, acc_old=acc_o, q_old=q_old)
            h.predict(ri.dt[k], ri.om[k], ri.acc[k])
        for t in tw:
            t.predict(ri.dt[k], ri.om[k], ri.acc[k])
        q_old = np.array([t.x[6:10] for t in tw])
        s, s1 = f.get_state(), v1.get_state()
        Xr, Pr = np.array([t.x for t in tw]), np.array([t.P for t in tw])
        ex = max(ex, max(_rel(s["x"][b].cpu().numpy(), Xr[b]) for b in range(B)))
        eP = max(eP, max(_rel(s["P"][b].cpu().numpy(), Pr[b]) for b in range(B)))
        moved = max(moved, float((s["x"] - s1["x"]).abs().max()), float((s["P"] - s1["P"]).abs().max() / s1["P"].abs().max()))
        # ---- update (symmetric covariance on both sides: the device stores the lower triangle)
        for t in tw:
            t.P = np.tril(t.P) + np.tril(t.P, -1).T
        X = np.array([t.x for t in tw]); P = np.array([t.P for t in tw])
        om_o = np.array([t.om_old for t in tw]); acc_o = np.array([t.acc_old for t in tw])
        for h in (f, v1):
            h.set_state(x=X, P=P, om_old=om_o, acc_old=acc_o, q_old=q_old)
        # (a measured attitude a few degrees off the state's: the rotation residual and the camera error quaternion have something to do)
        ang = np.radians(3.0)
        dq = np.array([np.sin(ang / 2) * 0.6, np.sin(ang / 2) * 0.0, np.sin(ang / 2) * 0.8, np.cos(ang / 2)])
        quat = np.array([ref.quat_mul(ri.quat[k] / np.linalg.norm(ri.quat[k]), dq)] * B)
        pos = np.tile(ri.pos[k], (B, 1)) + 0.05
        notch = np.full(B, 0.01)
        st, _ = f.update(pos, quat, notch)
        v1.update(pos, quat, notch)
        assert int(st.max()) == 0
        for b, t in enumerate(tw):
            assert t.update(pos[b], quat[b], notch[b]) is not None
        s, s1 = f.get_state(), v1.get_state()
        Xr, Pr = np.array([t.x for t in tw]), np.array([t.P for t in tw])
        ex = max(ex, max(_rel(s["x"][b].cpu().numpy(), Xr[b]) for b in range(B)))
        eP = max(eP, max(_rel(s["P"][b].cpu().numpy(), Pr[b]) for b in range(B)))
        moved = max(moved, float((s["x"] - s1["x"]).abs().max()), float((s["P"] - s1["P"]).abs().max() / s1["P"].abs().max()))
    print(f"quirks off {off}: x {ex:.1e} P {eP:.1e} vs the restatement; the v1 handle differs by {moved:.1e}")
    assert ex <= TOL_X and eP <= TOL_P, (off, ex, eP)
    assert moved > 1e-6, "the flag changed nothing"


def test_quirk_q1_off_refuses_what_its_covariance_kernel_cannot_do():
    from dvi_ekf_amd._capi import EskfError
    from dvi_ekf_amd.batch import BatchedESKF
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=True)
    f = BatchedESKF(3, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa, sigma_om=so, hold_frozen=True, quirks_fixed=1)
    f.set_noise(Q, R)
    f.set_state(x=np.tile(ri.x0, (3, 1)), P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    with pytest.raises(EskfError):
        f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0], want_gain=True)     # gain output together with a predict
    f.set_lanes_per_instance(32)
    with pytest.raises(EskfError):
        f.predict(ri.dt[0], ri.om[0], ri.acc[0])
    f.set_lanes_per_instance(1)
    f.predict(ri.dt[0], ri.om[0], ri.acc[0])
    st, K = f.update(ri.pos[0], ri.quat[0], ri.notch[0], want_gain=True)        # an update on its own may form the gain
    assert int(st.max()) == 0 and bool(torch.isfinite(K).all())
    with pytest.raises(EskfError):
        BatchedESKF(3, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa, sigma_om=so, quirks_fixed=16)
    # a run of epochs with the flag goes epoch by epoch through the same kernel
    n_pred = [2, 3]
    f.run_epochs(n_pred, ri.dt[1:6], ri.om[1:6], ri.acc[1:6], ri.pos[1:3], ri.quat[1:3], ri.notch[1:3])
    assert bool(torch.isfinite(f.x).all())
