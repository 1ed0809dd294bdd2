"""C-ABI behaviour on a real device: argument validation (negative codes, nothing thrown),
deferred-injection bookkeeping, raw buffer views."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tests.helpers import default_run


def _make(B=4, hold=True):
    from dvi_ekf_amd.batch import BatchedESKF
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa,
                    sigma_om=so, hold_frozen=hold)
    f.set_noise(Q, R)
    f.set_state(x=ri.x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    return f, ri, Q, R


def test_error_codes_instead_of_exceptions():
    from dvi_ekf_amd import _capi
    f, ri, Q, R = _make()
    L = _capi.lib()
    dp = C.POINTER(C.c_double)
    Qbad = np.ascontiguousarray(Q.copy())
    Qbad[0, 1] = 1e-3                                    # the reference never builds a non-diagonal Q
    assert L.dvi_eskf_set_noise(f._h, Qbad.ctypes.data_as(dp), None) == -1
    assert L.dvi_eskf_set_lanes_per_instance(f._h, 7) == -1
    assert L.dvi_eskf_predict(f._h, None, 0, None, None, 0, None) == -1      # missing inputs
    assert L.dvi_eskf_update(None, None, None, None, 0, None, None, None) == -1
    p = _capi.Params()
    h = C.c_void_p()
    assert L.dvi_eskf_create(C.byref(p), 0, 0, 0, C.byref(h)) == -1           # batch must be positive
    assert L.dvi_eskf_create(C.byref(p), 4, 7, 0, C.byref(h)) == -1           # unknown dtype
    assert L.dvi_eskf_create(C.byref(p), 4, 0, 4096, C.byref(h)) == -2        # no such device
    assert L.dvi_eskf_log_rows(f._h, 0.0, None, None) == -1                   # no output buffer
    assert L.dvi_eskf_set_noise_per_instance(f._h, None, None, None, None) == -1
    qn = np.full((4, 13), np.nan)
    assert L.dvi_eskf_set_noise_per_instance(f._h, qn.ctypes.data_as(dp), np.ones((4, 7)).ctypes.data_as(dp), None, None) == -1
    with pytest.raises(_capi.EskfError):
        f.set_noise(Qbad, None)
    # the handle is still usable
    f.predict(ri.dt[0], ri.om[0], ri.acc[0])
    assert bool(torch.isfinite(f.x).all())


def test_deferred_injection_is_invisible_to_the_caller():
    """After step(), every way of looking at the state shows the injected state."""
    f, ri, Q, R = _make()
    g = f.clone()
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    g.predict(ri.dt[0], ri.om[0], ri.acc[0])
    x_pred = g.x.clone()
    g.update(ri.pos[0], ri.quat[0], ri.notch[0])
    x_upd = g.x
    assert not torch.equal(x_pred, x_upd)
    assert torch.allclose(f.x, x_upd, rtol=0, atol=1e-12)
    assert torch.allclose(f.calibration(), x_upd[:, 10:16], rtol=0, atol=1e-12)
    assert torch.allclose(f.clone().x, x_upd, rtol=0, atol=1e-12)
    # a second step on top of a pending injection equals flush-then-step
    a, b = f.clone(), f.clone()
    a.step(ri.dt[1], ri.om[1], ri.acc[1], ri.pos[1], ri.quat[1], ri.notch[1])
    a.step(ri.dt[2], ri.om[2], ri.acc[2], ri.pos[2], ri.quat[2], ri.notch[2])     # pending from the first
    b.step(ri.dt[1], ri.om[1], ri.acc[1], ri.pos[1], ri.quat[1], ri.notch[1])
    _ = b.x                                                                          # flush in between
    b.step(ri.dt[2], ri.om[2], ri.acc[2], ri.pos[2], ri.quat[2], ri.notch[2])
    sa, sb = a.get_state(), b.get_state()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key


def test_flush_and_raw_views():
    from dvi_ekf_amd import _capi
    f, ri, Q, R = _make(B=3)
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    L = _capi.lib()
    assert L.dvi_eskf_flush(f._h, None) == 0 and L.dvi_eskf_flush(f._h, None) == 0     # idempotent
    assert L.dvi_eskf_raw_x(f._h) and L.dvi_eskf_raw_P(f._h)
    assert L.dvi_eskf_batch(f._h) == 3 and L.dvi_eskf_dtype(f._h) == 0
    st = f.get_state()
    assert st["x"].shape == (3, 26) and st["P"].shape == (3, 24, 24)
    assert torch.equal(st["P"], st["P"].transpose(1, 2))          # stored as its lower triangle
