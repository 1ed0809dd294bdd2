"""BASELINE config #5 as specified: the camera on the septic smooth-step path of the
reference's data/trajs/smooth_traj.py sampled at 30 Hz, IMU at 1 kHz from f_imu_meas on the device
(dvi_eskf_imu_synthesize), 33/33/34 predicts per update, 10 000 IMU steps, B = 8192 -- in fp64 and fp32.

What is asserted (tolerances measured on MI355X):
  * the device IMU stream equals the host restatement (<= 1e-12 relative);
  * fp64 vs the C oracle on 16 sampled instances at IMU steps 100 / 1 000 / 10 000, relative to the state's
    magnitude: <= 1e-9 at 100 and 1 000, <= 1e-7 at 10 000 (the reference filter amplifies perturbations by
    ~1e2 on this run: two instances that start 0.1 apart end 10 apart);
  * fp32 vs fp64, all 8192 instances, at the same steps: see FP32_TOL;
  * every status word of every update of both runs is 0 and everything stays finite.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle

CHECKPOINTS = (100, 1000, 10000)
FP64_TOL = {100: 1e-9, 1000: 1e-9, 10000: 1e-7}
# fp32 vs fp64, max over instances of |dx| / max(1, |x|) and of |dP| / max|P| (stated fp32 tolerance of the path)
FP32_TOL = {100: (2e-5, 2e-3), 1000: (2e-4, 5e-3), 10000: (5e-2, 5e-2)}


def _run(f, wl, om, acc, dev):
    """the whole run as three calls of run_epochs (to steps 100, 1 000, 10 000); returns {step: (x, P)}"""
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=f.dtype, device=dev)
    cum = np.cumsum(wl.n_pred)
    out, e0, i0 = {}, 0, 0
    for cp in CHECKPOINTS:
        e1 = int(np.searchsorted(cum, cp)) + 1
        assert cum[e1 - 1] == cp
        i1 = cp
        st, _ = f.run_epochs(wl.n_pred[e0:e1], t(wl.dt[i0:i1]), om[i0:i1].to(f.dtype), acc[i0:i1].to(f.dtype),
                             t(wl.pos[e0:e1]), t(wl.quat[e0:e1]), t(wl.notch[e0:e1]))
        assert int(st.abs().max()) == 0, f"status words at checkpoint {cp}"
        s = f.get_state(("x", "P"))
        out[cp] = (s["x"].double().cpu().numpy(), s["P"].double().cpu().numpy())
        e0, i0 = e1, i1
    return out


def test_config5_ten_thousand_steps_fp64_vs_oracle_and_fp32_vs_fp64():
    from dvi_ekf_amd.workloads import long_horizon
    B = 8192
    dev = torch.device("cuda", 0)
    wl = long_horizon(batch=B)
    assert wl.n_steps == 10000 and list(wl.n_pred[:3]) == [33, 33, 34] and len(wl.n_pred) == 300

    # IMU at 1 kHz from the device kernel, checked against the host restatement of f_imu_meas
    om_d, acc_d = wl.synthesize_imu_on_device(dtype=torch.float64, device=dev)
    om_h = np.vstack([wl.om0[None], wl.om])
    acc_h = np.vstack([wl.acc0[None], wl.acc])
    assert np.abs(om_d.cpu().numpy() - om_h).max() <= 1e-12 * max(1.0, np.abs(om_h).max())
    assert np.abs(acc_d.cpu().numpy() - acc_h).max() <= 1e-12 * max(1.0, np.abs(acc_h).max())
    om, acc = om_d[1:].contiguous(), acc_d[1:].contiguous()

    f64 = wl.make_filter(dtype=torch.float64, device=dev)
    f32 = wl.make_filter(dtype=torch.float32, device=dev)
    r64 = _run(f64, wl, om, acc, dev)
    r32 = _run(f32, wl, om, acc, dev)

    # the oracle on 16 sampled instances, the full horizon
    sample = np.linspace(0, B - 1, 16).astype(np.int64)
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, wl.hold_frozen, wl.sigma_a,
                               wl.sigma_om, wl.Q, wl.R)
    o = c_oracle.OracleBatch(prm, wl.x0[sample], wl.P0, wl.om0, wl.acc0)
    om_np, acc_np = om.cpu().numpy(), acc.cpu().numpy()
    i, ref = 0, {}
    for e, n in enumerate(wl.n_pred):
        for _ in range(n):
            o.predict(wl.dt[i], om_np[i], acc_np[i])
            i += 1
        assert int(o.update(wl.pos[e], wl.quat[e], wl.notch[e]).max()) == 0
        if i in CHECKPOINTS:
            ref[i] = (o.x.copy(), o.P.copy())

    rel = lambda a, b: (np.abs(a - b).reshape(a.shape[0], -1).max(axis=1) /
                        np.maximum(1.0, np.abs(b).reshape(b.shape[0], -1).max(axis=1))).max()
    relP = lambda a, b: (np.abs(a - b).reshape(a.shape[0], -1).max(axis=1) / np.abs(b).reshape(b.shape[0], -1).max(axis=1)).max()
    report = []
    for cp in CHECKPOINTS:
        x64, P64 = r64[cp]
        x32, P32 = r32[cp]
        assert np.isfinite(x64).all() and np.isfinite(P64).all() and np.isfinite(x32).all() and np.isfinite(P32).all()
        e64 = rel(x64[sample], ref[cp][0])
        eP64 = relP(P64[sample], ref[cp][1])
        e32 = rel(x32, x64)
        eP32 = relP(P32, P64)
        spread = np.abs(x64 - x64[0]).max()
        report.append(f"step {cp}: fp64 vs oracle x {e64:.1e} P {eP64:.1e} | fp32 vs fp64 x {e32:.1e} P {eP32:.1e} | "
                      f"spread of the batch {spread:.2g}")
        assert e64 <= FP64_TOL[cp] and eP64 <= 1e2 * FP64_TOL[cp], report[-1]
        assert e32 <= FP32_TOL[cp][0] and eP32 <= FP32_TOL[cp][1], report[-1]
    print("config #5 (B = 8192, 10 000 IMU steps, 300 updates):\n  " + "\n  ".join(report))
    assert np.abs(r64[10000][0]).max() < 1e3          # the run stayed in its bounded regime


def test_config5_default_yaml_noise_parity_over_the_stable_horizon():
    """The same run with the camera noise of the default config.yaml (attitude 1e-5 deg): the reference
    filter leaves its bounded regime after ~2 s of motion (workloads.long_horizon), so the comparison with the
    oracle covers the horizon on which it has not: the first 1 000 IMU steps (30 updates), fp64, <= 1e-9."""
    from dvi_ekf_amd.workloads import long_horizon
    B = 256
    dev = torch.device("cuda", 0)
    wl = long_horizon(batch=B, n_steps=1000, meas_pos_std=None, meas_theta_deg=None)
    f = wl.make_filter(dtype=torch.float64, device=dev)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    st, _ = f.run_epochs(wl.n_pred, t(wl.dt), t(wl.om), t(wl.acc), t(wl.pos), t(wl.quat), t(wl.notch))
    assert int(st.abs().max()) == 0
    sample = np.linspace(0, B - 1, 16).astype(np.int64)
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, wl.hold_frozen, wl.sigma_a,
                               wl.sigma_om, wl.Q, wl.R)
    o = c_oracle.OracleBatch(prm, wl.x0[sample], wl.P0, wl.om0, wl.acc0)
    i = 0
    for e, n in enumerate(wl.n_pred):
        for _ in range(n):
            o.predict(wl.dt[i], wl.om[i], wl.acc[i])
            i += 1
        o.update(wl.pos[e], wl.quat[e], wl.notch[e])
    x = f.x.cpu().numpy()[sample]
    err = (np.abs(x - o.x).max(axis=1) / np.maximum(1.0, np.abs(o.x).max(axis=1))).max()
    print(f"config #5 with the default YAML noise, {i} IMU steps / {len(wl.n_pred)} updates: fp64 vs oracle {err:.1e}")
    assert err <= 1e-9
