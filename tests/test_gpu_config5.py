"""BASELINE config #5 as specified: the camera on the septic smooth-step path of the
reference's data/trajs/smooth_traj.py sampled at 30 Hz, IMU at 1 kHz from f_imu_meas on the device
(dvi_eskf_imu_synthesize), 33/33/34 predicts per update, 10 000 IMU steps, B = 8192 -- in fp64 and fp32.

The reference filter is not a contraction on this run: after the camera's motion (steps 0..1 700) it amplifies
perturbations by many orders of magnitude (two fp64 implementations that agree to 1e-15 at step 1 000 are 4e-2
apart at step 10 000), so a free-running comparison of two arithmetics is a statement about the arithmetic only
up to step ~1 000.  What is asserted (tolerances measured on MI355X):
  * the device IMU stream equals the host restatement (<= 1e-12 relative);
  * fp64 vs the C oracle, 64 sampled instances, free-running: <= 1e-9 at IMU steps 100 and 1 000;
    and over the LAST hundred steps 9 900..10 000 with the oracle started from the GPU's state at step 9 900;
  * fp32 vs fp64, all 8192 instances: free-running at steps 100 and 1 000, and over 9 900..10 000 with the fp32
    filter restarted from the fp64 state at step 9 900 (FP32_TOL); the free-running difference at step 10 000 is
    reported next to the fp64-vs-oracle difference there (both are the dynamics' amplification of rounding);
  * both runs do all 10 000 steps: no update is skipped, everything stays finite and bounded.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle

CHECKPOINTS = (100, 1000, 9900, 10000)
FP64_TOL = 1e-9
# fp32 vs fp64: max over instances of |dx| / max(1, |x|) and of |dP| / max|P|  -- the stated fp32 tolerance of the path
# (measured: 1.2e-6 / 1.5e-6 at step 100, 5.0e-6 / 6.6e-5 at step 1 000, 3.3e-3 ... 4.3e-2 / 5.2e-6 ... 1.1e-5 over the last
#  hundred steps, where the dynamics amplify a perturbation ~400x per hundred steps -- the fp64-vs-oracle figure there is
#  1e-12 -- and the worst instance of 8 192 depends on which way single roundings fall: two builds that differ in where
#  a product is fused into an fma gave 3.3e-3 and 4.3e-2)
FP32_TOL = {100: (1e-5, 1e-4), 1000: (5e-5, 1e-3), "last100": (None, 1e-4)}
# The last hundred steps in x: the WORST of the 8 192 instances is printed, not asserted (it is the dynamics' 400x per
# hundred steps applied to which way single roundings fell); asserted are the median and the 99th percentile over the
# instances, and -- the statement about the fp32 kernels themselves -- ONE epoch (33 predicts + the update) forced from
# the fp64 state at step 9 900, where nothing has been amplified yet.
# (measured: median 2.9e-6, 99th percentile 6.8e-5, worst 8.8e-3 -- 3.3e-3 and 4.3e-2 in two other builds --; one epoch:
#  x 5.1e-4 at the worst instance, P 1.7e-6)
FP32_LAST100_X = {"median": 2e-5, "p99": 5e-4, "worst": 0.5}
# ONE epoch in x: 99th percentile over the instances and a hard cap on the worst one (which way a single rounding of the
# half-angle residual falls near |v| = 1 moves the worst of 8 192 instances between 5e-4 and 4e-2 from build to build);
# P: the worst instance
FP32_ONE_EPOCH = {"p99": 2e-3, "worst": 0.5, "P": 2e-5}


def _segments(f, wl, om, acc, dev, checkpoints, first=0, flagged=None):
    """runs the epochs from IMU step ``first`` to every checkpoint in turn (one run_epochs call per segment);
    returns {step: (x, P)}.  No update may be skipped (status bit 0); bit 1 -- the argument of the half-angle
    residual's asin had to be clamped to 1 -- is collected per instance in ``flagged`` when given, else it must
    not occur either."""
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=f.dtype, device=dev)
    cum = np.cumsum(wl.n_pred)
    e0 = 0 if first == 0 else int(np.searchsorted(cum, first)) + 1
    out, i0 = {}, first
    for cp in checkpoints:
        e1 = int(np.searchsorted(cum, cp)) + 1
        assert cum[e1 - 1] == cp
        st, _ = f.run_epochs(wl.n_pred[e0:e1], t(wl.dt[i0:cp]), om[i0:cp].to(f.dtype), acc[i0:cp].to(f.dtype),
                             t(wl.pos[e0:e1]), t(wl.quat[e0:e1]), t(wl.notch[e0:e1]))
        assert int((st & 1).max()) == 0, f"skipped updates before checkpoint {cp}"
        if flagged is None:
            assert int(st.max()) == 0, f"status words at checkpoint {cp}"
        else:
            flagged |= (st != 0).any(dim=0).cpu().numpy()
        s = f.get_state(("x", "P", "om_old", "acc_old", "q_old"))
        out[cp] = {k: v.double().cpu().numpy() for k, v in s.items()}
        e0, i0 = e1, cp
    return out


def _rel(a, b):
    n = a.shape[0]
    return (np.abs(a - b).reshape(n, -1).max(axis=1) / np.maximum(1.0, np.abs(b).reshape(n, -1).max(axis=1))).max()


def _relP(a, b):
    n = a.shape[0]
    return (np.abs(a - b).reshape(n, -1).max(axis=1) / np.abs(b).reshape(n, -1).max(axis=1)).max()


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
    om_np, acc_np = om.cpu().numpy(), acc.cpu().numpy()

    f64 = wl.make_filter(dtype=torch.float64, device=dev)
    f32 = wl.make_filter(dtype=torch.float32, device=dev)
    r64 = _segments(f64, wl, om, acc, dev, CHECKPOINTS)
    # fp32: where the filter's camera attitude is ~180 deg from the measured one (|v| of the unit error
    # quaternion within an ulp of 1), rounding can put |v| above 1 and the asin argument is clamped (status
    # bit 1; the reference's math.asin would raise there).  Those instances are counted and must stay rare.
    clamped = np.zeros(B, dtype=bool)
    r32 = _segments(f32, wl, om, acc, dev, CHECKPOINTS, flagged=clamped)
    assert clamped.sum() <= 0.02 * B, clamped.sum()
    # ... and the last hundred steps again, fp32 restarted from the fp64 state at step 9 900
    g32 = wl.make_filter(dtype=torch.float32, device=dev)
    s = r64[9900]
    g32.set_state(x=s["x"], P=s["P"], om_old=s["om_old"], acc_old=s["acc_old"], q_old=s["q_old"])
    w32 = _segments(g32, wl, om, acc, dev, (10000,), first=9900, flagged=np.zeros(B, dtype=bool))[10000]
    # ... and ONE epoch from that state in both precisions (a forced step: nothing amplified)
    cum_ = np.cumsum(wl.n_pred)
    e9900 = int(np.searchsorted(cum_, 9900)) + 1
    cp1 = int(cum_[e9900])
    h32, h64 = wl.make_filter(dtype=torch.float32, device=dev), wl.make_filter(dtype=torch.float64, device=dev)
    for h in (h32, h64):
        h.set_state(x=s["x"], P=s["P"], om_old=s["om_old"], acc_old=s["acc_old"], q_old=s["q_old"])
    one32 = _segments(h32, wl, om, acc, dev, (cp1,), first=9900, flagged=np.zeros(B, dtype=bool))[cp1]
    one64 = _segments(h64, wl, om, acc, dev, (cp1,), first=9900)[cp1]

    # the oracle on 64 sampled instances: free-running over the full horizon, and restarted at step 9 900
    sample = np.linspace(0, B - 1, 64).astype(np.int64)
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, wl.hold_frozen, wl.sigma_a,
                               wl.sigma_om, wl.Q, wl.R)
    o = c_oracle.OracleBatch(prm, wl.x0[sample], wl.P0, wl.om0, wl.acc0)
    cum = np.cumsum(wl.n_pred)

    def oracle_run(o, e0, e1, i):
        for e in range(e0, e1):
            for _ in range(wl.n_pred[e]):
                o.predict(wl.dt[i], om_np[i], acc_np[i])
                i += 1
            assert int(o.update(wl.pos[e], wl.quat[e], wl.notch[e]).max()) == 0
        return i

    ref, e0, i = {}, 0, 0
    for cp in CHECKPOINTS:
        e1 = int(np.searchsorted(cum, cp)) + 1
        i = oracle_run(o, e0, e1, i)
        ref[cp] = (o.x.copy(), o.P.copy())
        e0 = e1
    from dvi_ekf_amd import rotmath as rm
    ow = c_oracle.OracleBatch(prm, s["x"][sample], s["P"][sample], s["om_old"][sample], s["acc_old"][sample])
    for b in range(len(sample)):
        ow.Rwb[b] = rm.quat_to_matrix(s["q_old"][sample[b]]).reshape(9)     # R_WB_old is not Rm(x.q) after an update (Q7)
    oracle_run(ow, int(np.searchsorted(cum, 9900)) + 1, 300, 9900)

    report = []
    ok = True
    for cp in CHECKPOINTS:
        a, b32 = r64[cp], r32[cp]
        for r in (a, b32):
            assert np.isfinite(r["x"]).all() and np.isfinite(r["P"]).all()
        e64, eP64 = _rel(a["x"][sample], ref[cp][0]), _relP(a["P"][sample], ref[cp][1])
        e32, eP32 = _rel(b32["x"], a["x"]), _relP(b32["P"], a["P"])
        report.append(f"step {cp}, free-running: fp64 vs oracle x {e64:.1e} P {eP64:.1e} | fp32 vs fp64 x {e32:.1e} P {eP32:.1e} | "
                      f"spread of the batch {np.abs(a['x'] - a['x'][0]).max():.2g}")
        if cp in FP32_TOL:
            ok &= e64 <= FP64_TOL and eP64 <= 1e2 * FP64_TOL and e32 <= FP32_TOL[cp][0] and eP32 <= FP32_TOL[cp][1]
    a = r64[10000]
    e64, eP64 = _rel(a["x"][sample], ow.x), _relP(a["P"][sample], ow.P)
    e32, eP32 = _rel(w32["x"], a["x"]), _relP(w32["P"], a["P"])
    report.append(f"steps 9900..10000 from the fp64 state at 9900: fp64 vs oracle x {e64:.1e} P {eP64:.1e} | fp32 vs fp64 x {e32:.1e} P {eP32:.1e}")
    ok &= e64 <= FP64_TOL and eP64 <= 1e2 * FP64_TOL and eP32 <= FP32_TOL["last100"][1]
    per = np.abs(w32["x"] - a["x"]).max(axis=1) / np.maximum(1.0, np.abs(a["x"]).max(axis=1))
    med, p99 = float(np.median(per)), float(np.percentile(per, 99))
    e1x, e1P = _rel(one32["x"], one64["x"]), _relP(one32["P"], one64["P"])
    per1 = np.abs(one32["x"] - one64["x"]).max(axis=1) / np.maximum(1.0, np.abs(one64["x"]).max(axis=1))
    e1_p99 = float(np.percentile(per1, 99))
    report.append(f"  the same over the instances: median {med:.1e}, 99th percentile {p99:.1e}, worst {per.max():.1e}; "
                  f"ONE epoch ({cp1 - 9900} predicts + update) from the fp64 state at 9900: fp32 vs fp64 x median {np.median(per1):.1e}, "
                  f"99th percentile {e1_p99:.1e}, 99.9th {np.percentile(per1, 99.9):.1e}, worst {e1x:.1e}; P {e1P:.1e}")
    ok &= med <= FP32_LAST100_X["median"] and p99 <= FP32_LAST100_X["p99"] and float(per.max()) <= FP32_LAST100_X["worst"]
    ok &= e1_p99 <= FP32_ONE_EPOCH["p99"] and e1x <= FP32_ONE_EPOCH["worst"] and e1P <= FP32_ONE_EPOCH["P"]
    print(f"config #5 (B = 8192, 10 000 IMU steps, 300 updates; {clamped.sum()} fp32 instances with a clamped asin):\n  "
          + "\n  ".join(report))
    assert ok, report
    assert np.abs(r64[10000]["x"]).max() < 1e3 and np.abs(r32[10000]["x"]).max() < 1e3      # bounded regime, both arithmetics


def test_config5_default_yaml_noise_parity_over_the_stable_horizon():
    """The same run with the camera noise of the default config.yaml (attitude 1e-5 deg): the reference filter
    leaves its bounded regime after ~2 s of motion (workloads.long_horizon) and amplifies rounding differences
    from the start, so the comparison with the oracle covers the horizon on which that amplification still leaves
    1e-9: the first STABLE_STEPS IMU steps; the error curve to step 1 000 is printed."""
    from dvi_ekf_amd.workloads import long_horizon
    B, STABLE_STEPS = 256, 700
    dev = torch.device("cuda", 0)
    wl = long_horizon(batch=B, n_steps=1000, meas_pos_std=None, meas_theta_deg=None)
    f = wl.make_filter(dtype=torch.float64, device=dev)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    cps = (100, 200, 300, 400, 500, 600, 700, 800, 900, 1000)
    got = _segments(f, wl, t(wl.om), t(wl.acc), dev, cps)
    sample = np.linspace(0, B - 1, 64).astype(np.int64)
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, wl.hold_frozen, wl.sigma_a,
                               wl.sigma_om, wl.Q, wl.R)
    o = c_oracle.OracleBatch(prm, wl.x0[sample], wl.P0, wl.om0, wl.acc0)
    i, curve = 0, {}
    for e, n in enumerate(wl.n_pred):
        for _ in range(n):
            o.predict(wl.dt[i], wl.om[i], wl.acc[i])
            i += 1
        o.update(wl.pos[e], wl.quat[e], wl.notch[e])
        if i in cps:
            curve[i] = _rel(got[i]["x"][sample], o.x)
    holds = [k for k in cps if all(curve[j] <= 1e-9 for j in cps if j <= k)]
    last = max(holds) if holds else 0
    print("config #5 with the default YAML noise, fp64 vs oracle on 64 instances, free-running: " +
          "  ".join(f"{k}: {v:.1e}" for k, v in curve.items()) + f"  -> 1e-9 holds through step {last}")
    # (measured: 1e-9 holds through step 800 or 900 depending on the build's rounding; the asserted horizon is 700)
    assert last >= STABLE_STEPS
