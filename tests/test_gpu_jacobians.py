"""Filter.Fx / Filter.Fi (Filter.py:249-268,279-285,334-342): the dense Jacobians expanded from the blocks
the kernels work with (dvi_eskf_get_jacobians) against the oracle, which obtains the camera rows by
differentiating the reference's own expressions (oracle/eskf_oracle.c: forward-mode duals)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle
from tests.helpers import default_run
from tests.test_gpu_parity import _make, _oracle, _perturbed_x0


def _check(f, o, B, dt, om, acc, tol=1e-12):
    """one predict on both sides from the oracle's state; returns max relative errors of Fx, Fi"""
    f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=o.q_old)
    f.predict(dt, om, acc)
    Fx, Fi = (a.cpu().numpy() for a in f.jacobians())
    ex = ei = 0.0
    dt = np.broadcast_to(np.asarray(dt, dtype=float), (B,))
    om, acc = np.broadcast_to(om, (B, 3)), np.broadcast_to(acc, (B, 3))
    for b in range(B):
        Fxo, Fio = o.predict_one(b, dt[b], om[b], acc[b])
        ex = max(ex, np.abs(Fx[b] - Fxo).max() / max(1.0, np.abs(Fxo).max()))
        ei = max(ei, np.abs(Fi[b] - Fio).max() / max(1.0, np.abs(Fio).max()))
        # structure: the column cursor of Filter.py:279-285
        assert np.array_equal(Fx[b][18:21, 16:19], np.eye(3)) and not Fx[b][18:21, 19:24].any()
        assert Fx[b][22, 22] == 1.0 and Fx[b][23, 23] == 1.0 and not Fx[b][21, 22:24].any()
    o.q_old = o.x[:, 6:10].copy()
    assert ex <= tol and ei <= tol, (ex, ei)
    return ex, ei


@pytest.mark.parametrize("golden,tune", [(True, False), (False, False), (False, True)])
def test_jacobians_along_the_reference_run(golden, tune):
    """golden run (1:1, dt = 1) and v1-literal run (10:1), perturbed instances, with updates in between;
    tune = True: non-zero gyro block in Q (the N6 rows of Fi are then also what the covariance kernel uses)."""
    B = 12
    cfg, ri, Q, R, sa, so = default_run(golden_compat=golden, fast=False, frozen=(0,) * 6, interframe=(1 if golden else None))
    if tune:
        Q = Q.copy()
        Q[0:3, 0:3] = 0.01 * np.diag(sa ** 2)
        Q[3:6, 3:6] = 0.01 * np.diag(so ** 2)
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, golden, x0)
    f.keep_jacobians(True)
    o = _oracle(cfg, ri, Q, R, sa, so, B, golden, x0)
    o.q_old = o.x[:, 6:10].copy()
    ip, worst = 0, (0.0, 0.0)
    for k in range(6):
        for _ in range(ri.n_pred[k]):
            e = _check(f, o, B, ri.dt[ip], ri.om[ip], ri.acc[ip])
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
            ip += 1
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
    print(f"Fx / Fi vs oracle (golden={golden}, tune={tune}): {worst[0]:.1e} / {worst[1]:.1e} over {ip} predicts x {B} instances")


def test_jacobians_at_random_configurations():
    """states nowhere near the reference trajectory: random attitudes, joint angles and notch over +-pi,
    random buffers and per-instance dt"""
    B = 512
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    rng = np.random.default_rng(99)
    x0 = np.tile(ri.x0, (B, 1))
    q = rng.standard_normal((B, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    qc = rng.standard_normal((B, 4)); qc /= np.linalg.norm(qc, axis=1, keepdims=True); qc[qc[:, 3] < 0] *= -1
    x0[:, 6:10], x0[:, 22:26] = q, qc
    x0[:, 10:13] = rng.uniform(-np.pi, np.pi, (B, 3))
    x0[:, 13:16] = rng.uniform(-30, 30, (B, 3))
    x0[:, 16:19] = rng.uniform(-np.pi, np.pi, (B, 3)) * np.array([1.0, 0.3, 0.1])
    f = _make(cfg, ri, Q, R, sa, so, B, False, x0)
    f.keep_jacobians(True)
    o = _oracle(cfg, ri, Q, R, sa, so, B, False, x0)
    o.om_old[:] = rng.normal(0, 0.3, (B, 3))
    o.acc_old[:] = rng.normal(0, 5.0, (B, 3))
    qo = rng.standard_normal((B, 4)); qo /= np.linalg.norm(qo, axis=1, keepdims=True); qo[qo[:, 3] < 0] *= -1
    o.q_old = qo
    from dvi_ekf_amd import rotmath as rm
    for b in range(B):
        o.Rwb[b] = rm.quat_to_matrix(qo[b]).reshape(9)
    dt = 10.0 ** rng.uniform(-3, 0.2, B)
    e = _check(f, o, B, dt, rng.normal(0, 0.3, (B, 3)), rng.normal(0, 5.0, (B, 3)))
    print(f"Fx / Fi vs oracle at {B} random configurations: {e[0]:.1e} / {e[1]:.1e}")


def test_jacobians_need_keeping_and_a_predict():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _make(cfg, ri, Q, R, sa, so, 2, True)
    f.predict(ri.dt[0], ri.om[0], ri.acc[0])
    with pytest.raises(_capi_error()):
        f.jacobians()                       # keeping was off during that predict
    f.keep_jacobians(True)
    with pytest.raises(_capi_error()):
        f.jacobians()                       # ... and no predict has run since
    f.predict(ri.dt[1], ri.om[1], ri.acc[1])
    Fx, Fi = f.jacobians()
    assert Fx.shape == (2, 24, 24) and Fi.shape == (2, 24, 13)


def _capi_error():
    from dvi_ekf_amd._capi import EskfError
    return EskfError
