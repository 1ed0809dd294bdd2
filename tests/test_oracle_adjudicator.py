"""The extended-precision build of the oracle (oracle/eskf_oracle.c with -DESKF_ORACLE_QUAD: every intermediate in
__float128) -- the adjudicator of the ill-conditioned updates in tests/test_gpu_parity.py.  CPU only: it must agree
with the fp64 oracle to rounding where the arithmetic is well-conditioned, reproduce the reference's saved
trajectory like it, and sit between the fp64 result and exactness where S is ill-conditioned."""
import numpy as np

from oracle import c_oracle
from tests.helpers import default_run, load_golden, log_row

SEL = [18, 19, 20, 21, 22, 23, 15]


def _pair(golden, frozen, B=8, interframe=None):
    cfg, ri, Q, R, sa, so = default_run(golden_compat=golden, fast=False, frozen=frozen, interframe=interframe)
    rng = np.random.default_rng(1000)
    x0 = np.tile(ri.x0, (B, 1))
    x0[1:, 10:13] += rng.normal(0, np.radians(5.0), (B - 1, 3))
    x0[1:, 13:16] += rng.normal(0, 10.0, (B - 1, 3))
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, golden, sa, so, Q, R)
    o = c_oracle.OracleBatch(prm, x0, ri.P0, ri.om0, ri.acc0)
    h = c_oracle.OracleBatch(prm, x0, ri.P0, ri.om0, ri.acc0, hp=True)
    return ri, R, o, h


def test_quad_build_reproduces_the_saved_trajectory():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, True, sa, so, Q, R)
    h = c_oracle.OracleBatch(prm, ri.x0[None, :], ri.P0, ri.om0, ri.acc0, hp=True)
    rows = [log_row(ri.t0, h.x[0])]
    for k in range(len(ri.t_meas)):
        h.predict(ri.dt[k], ri.om[k], ri.acc[k])
        assert int(h.update(ri.pos[k], ri.quat[k], ri.notch[k]).max()) == 0
        rows.append(log_row(ri.t_meas[k], h.x[0]))
    assert np.abs(np.array(rows) - load_golden()).max() <= 5.0e-10 + 1e-11


def test_quad_and_double_builds_agree_where_well_conditioned_and_separate_where_not():
    """forced steps: both start every step from the fp64 oracle's state (symmetric P, as the device stores it)"""
    ri, R, o, h = _pair(True, (0,) * 6, interframe=1)
    well, ill = [], []
    for k in range(14):
        for orc in (h,):
            orc.x[:], orc.P[:], orc.om_old[:], orc.acc_old[:], orc.Rwb[:] = o.x, o.P, o.om_old, o.acc_old, o.Rwb
        o.predict(ri.dt[k], ri.om[k], ri.acc[k])
        h.predict(ri.dt[k], ri.om[k], ri.acc[k])
        scale = np.maximum(1.0, np.abs(h.x).max(axis=1))
        assert (np.abs(o.x - h.x).max(axis=1) / scale).max() <= 1e-13                 # a predict is well-conditioned
        o.P[:] = np.tril(o.P) + np.transpose(np.tril(o.P, -1), (0, 2, 1))
        h.x[:], h.P[:] = o.x, o.P
        kappa = []
        for b in range(o.B):
            S = o.P[b][np.ix_(SEL, SEL)] + R
            d = 1.0 / np.sqrt(np.diag(S))
            kappa.append(np.linalg.cond(S * np.outer(d, d)))
        kappa = np.array(kappa)
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
        h.update(ri.pos[k], ri.quat[k], ri.notch[k])
        err = np.abs(o.x - h.x).max(axis=1) / np.maximum(1.0, np.abs(h.x).max(axis=1))
        well += list(err[kappa < 1e3])
        ill += list((err / (kappa * 2.0 ** -52))[kappa > 1e4])
    assert len(well) > 20 and max(well) <= 1e-13
    # the fp64 oracle's distance from exact stays inside the forward-error bound of a stable fp64 solve ...
    assert len(ill) > 5 and max(ill) <= 4.0
    # ... and the two builds are not the same arithmetic: they do differ at rounding level
    assert np.abs(o.P - h.P).max() > 0
