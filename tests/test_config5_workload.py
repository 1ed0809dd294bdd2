"""BASELINE config #5 workload generator (host side): the septic smooth-step path of the reference's
data/trajs/smooth_traj.py, the 1 kHz / 30 Hz schedule, and the oracle on it (CPU only)."""
import numpy as np

from dvi_ekf_amd import smooth_traj as st
from dvi_ekf_amd.workloads import long_horizon
from oracle import c_oracle


def test_traj_gen_is_the_septic_smooth_step():
    """smooth_traj.py:6-12: value z0 at t_prev, zT at T, and the first three derivatives vanish at both ends"""
    t = np.linspace(2.0, 5.0, 3001)
    z = st.traj_gen(1.5, -0.5, t, 2.0, 5.0)
    assert z[0] == 1.5 and abs(z[-1] + 0.5) < 1e-14
    h = t[1] - t[0]
    d1 = np.gradient(z, h)
    d2 = np.gradient(d1, h)
    d3 = np.gradient(d2, h)
    for d, tol in ((d1, 1e-5), (d2, 1e-2), (d3, 2e-1)):   # one-sided differences of a C^3-flat end
        assert abs(d[0]) < tol and abs(d[-1]) < tol
    s = (t - 2.0) / 3.0
    assert np.allclose(z, 1.5 - 2.0 * (35 * s ** 4 - 84 * s ** 5 + 70 * s ** 6 - 20 * s ** 7), atol=1e-14)


def test_gen_values_partitions_like_the_reference_script():
    """smooth_traj.py:15-39 with the script's own numbers: 60 frames, partitions (0, 0.17, 1): the first segment
    holds ceil(0.17 * 60) = 11 frames and reaches vals[1] at its last frame; the rest stays there"""
    t = np.linspace(0, 59, 60)
    v = st.gen_values(t, [0.0, 1.0, 1.0], 60, [0, 0.17, 1])
    assert v.shape == (60,) and v[0] == 0.0 and abs(v[10] - 1.0) < 1e-14 and np.allclose(v[10:], 1.0)
    assert (np.diff(v[:11]) > 0).all()
    # the script's amplitudes are all zero: a camera at rest
    assert not st.gen_values(t, 0 * np.array([0, 1, 1]), 60, [0, 0.17, 1]).any()
    qx, qy, qz, qw = st.gen_values_rot(t, [0.0, 90.0, 90.0], 60, "z", [0, 0.17, 1])
    assert not qx.any() and not qy.any() and abs(qz[-1] - np.sin(np.pi / 4)) < 1e-15 and abs(qw[0] - 1.0) == 0


def test_long_horizon_schedule_and_oracle_run():
    # (13 frames only: the first partition is 3 frames long, so the amplitudes are scaled down with it)
    wl = long_horizon(batch=4, n_steps=400, amplitude_xyz=(0.3, -0.2, 0.15), angle_deg=1.5)
    assert wl.n_steps == 400 and wl.n_pred.sum() == 400 and list(wl.n_pred[:6]) == [33, 33, 34, 33, 33, 34]
    assert np.allclose(wl.dt, 1e-3, atol=1e-15) and wl.om.shape == (400, 3) and wl.pos.shape == (12, 3)
    assert np.allclose(wl.t_meas, np.arange(1, 13) / 30.0)
    # frame k is applied after the last IMU sample with t_i <= t_k
    cum = np.cumsum(wl.n_pred)
    assert all(wl.t_imu[c - 1] <= tk + 1e-9 < wl.t_imu[c - 1] + 1e-3 for c, tk in zip(cum, wl.t_meas))
    assert np.array_equal(wl.x0[0, 10:16], wl.dofs_gt) and not np.array_equal(wl.x0[1, 0:6], wl.x0[0, 0:6])
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, wl.hold_frozen, wl.sigma_a,
                               wl.sigma_om, wl.Q, wl.R)
    o = c_oracle.OracleBatch(prm, wl.x0, wl.P0, wl.om0, wl.acc0)
    i = 0
    for e, n in enumerate(wl.n_pred):
        for _ in range(n):
            o.predict(wl.dt[i], wl.om[i], wl.acc[i])
            i += 1
        assert int(o.update(wl.pos[e], wl.quat[e], wl.notch[e]).max()) == 0
    # the filter tracks the camera it is shown (instance 0 starts at the true state)
    assert np.abs(o.x[0, 19:22] - wl.pos[-1]).max() < 0.05 and np.isfinite(o.x).all()
