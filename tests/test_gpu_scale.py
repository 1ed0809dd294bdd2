"""BASELINE.json configurations at their full sizes, through size-independent properties,
plus sampled comparisons with the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle
from tests.helpers import default_run


def _filter(cfg, Q, R, sa, so, B, hold, dtype=torch.float64):
    from dvi_ekf_amd.batch import BatchedESKF
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa,
                    sigma_om=so, hold_frozen=hold, dtype=dtype)
    f.set_noise(Q, R)
    return f


def test_config2_and_4_full_batch_identical_instances_and_symmetry():
    """B = 65 536 (config #4 size) with identical inputs: every instance must equal instance 0 bit
    for bit (no cross-talk between the instances of a wavefront / tail handling), instance 0 must
    follow the oracle, and P must stay symmetric positive semi-definite on the sane horizon."""
    B = 65536
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=False, frozen=(1,) * 6, interframe=1)
    f = _filter(cfg, Q, R, sa, so, B, True)
    f.set_state(x=ri.x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, True, sa, so, Q, R)
    o = c_oracle.OracleBatch(prm, ri.x0[None, :], ri.P0, ri.om0, ri.acc0)
    for k in range(12):
        st, _ = f.step(ri.dt[k], ri.om[k], ri.acc[k], ri.pos[k], ri.quat[k], ri.notch[k])
        o.predict(ri.dt[k], ri.om[k], ri.acc[k])
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
    assert int(st.abs().max()) == 0
    x = f.x
    assert bool((x == x[0:1]).all())
    raw = f.get_state(("P",))["P"]
    assert bool((raw == raw[0:1]).all())
    assert np.abs(x[0].cpu().numpy() - o.x[0]).max() <= 1e-9
    P0 = raw[0].cpu().numpy()
    assert np.array_equal(P0, P0.T)
    assert np.linalg.eigvalsh(P0).min() > -1e-9 * np.abs(P0).max()


def test_config2_batch1024_perturbed_instances_vs_oracle():
    """config #2: B = 1024, fp64, 1:1 pairs; instance 0 = the reference run, the others start from
    perturbed extrinsic DOFs (unfrozen); 8 pair-steps.

    EVERY instance is checked on EVERY pair-step: the oracle runs free, and each pair-step of the GPU
    starts from the oracle's state (forced step) and is adjudicated by the extended-precision build of
    the oracle (tests/test_gpu_parity.py::adjudicate) -- also the instances whose state the reference
    filter itself has blown up (|x| > 100 for about half of them after 8 pairs; DESIGN.md, Numerical
    behaviour), for which the fp64 oracle alone is not a usable truth.  A second GPU filter runs free and
    must follow the oracle to 1e-9 wherever the reference filter is still in its physical range."""
    from tests.test_gpu_parity import adjudicate, innovation_conditions
    B = 1024
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=False, frozen=(0,) * 6, interframe=1)
    rng = np.random.default_rng(1000)
    x0 = np.tile(ri.x0, (B, 1))
    x0[1:, 10:16] += rng.normal(0, 1, (B - 1, 6)) * np.sqrt(np.diag(ri.P0)[9:15])
    f = _filter(cfg, Q, R, sa, so, B, True)
    f.set_state(x=x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    forced = _filter(cfg, Q, R, sa, so, B, True)
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, True, sa, so, Q, R)
    o = c_oracle.OracleBatch(prm, x0, ri.P0, ri.om0, ri.acc0)
    h = c_oracle.OracleBatch(prm, x0, ri.P0, ri.om0, ri.acc0, hp=True)
    sane = np.ones(B, dtype=bool)
    worst = 0.0
    closest = (0.0, 0.0)
    q_old = o.x[:, 6:10].copy()
    for k in range(8):
        f.step(ri.dt[k], ri.om[k], ri.acc[k], ri.pos[k], ri.quat[k], ri.notch[k])
        # (the device stores the lower triangle of P: every side starts the pair-step from that symmetric input)
        o.P[:] = np.tril(o.P) + np.transpose(np.tril(o.P, -1), (0, 2, 1))
        forced.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
        h.x[:], h.P[:], h.om_old[:], h.acc_old[:], h.Rwb[:] = o.x, o.P, o.om_old, o.acc_old, o.Rwb
        st, _ = forced.step(ri.dt[k], ri.om[k], ri.acc[k], ri.pos[k], ri.quat[k], ri.notch[k])
        for orc in (o, h):
            orc.predict(ri.dt[k], ri.om[k], ri.acc[k])
        q_old = o.x[:, 6:10].copy()
        kappa = innovation_conditions(o.P, R)
        # (the adjudicated quantity is the whole pair-step from the common input: the quad oracle's
        #  update sees the quad prediction, the fp64 oracle's and the GPU's see their own)
        for orc in (o, h):
            orc.update(ri.pos[k], ri.quat[k], ri.notch[k])
        assert int(st.abs().max()) == 0
        s = forced.get_state(("x", "P"))
        xe, Pe, xg, xr, Pg, Pr = adjudicate(s["x"].cpu().numpy(), s["P"].cpu().numpy(), o, h, kappa)
        i = int(np.argmax(xe))
        assert xe.max() <= 0 and Pe.max() <= 0, f"pair {k} instance {i}: x gpu {xg[i]:.1e} ref {xr[i]:.1e} cond {kappa[i]:.1e}"
        if xg.max() > closest[0]:
            closest = (float(xg.max()), float(xr[int(np.argmax(xg))]))
        sane &= np.abs(o.x).max(axis=1) < 100.0
        x = f.x.cpu().numpy()
        err = np.abs(x - o.x).max(axis=1) / np.maximum(1.0, np.abs(o.x).max(axis=1))
        if k < 5:
            worst = max(worst, err[sane].max())
    print(f"config #2: all {B} instances adjudicated on all 8 pair-steps (largest GPU distance from exact "
          f"{closest[0]:.1e}, fp64 oracle there {closest[1]:.1e}); free-running: {sane.sum()} of {B} stay in range over "
          f"8 pairs, max rel |dx| of those over the first 5 pairs = {worst:.2e}; instance 0 after 8 pairs = {err[0]:.1e}")
    assert sane[0] and err[0] <= 1e-9          # the unperturbed reference run, free-running
    assert sane.sum() >= B // 4                # (the free-running maximum above is over a substantial part of the batch)
    assert worst <= 1e-9


def test_config3_predict_only_fp32_one_million_instances():
    """config #3: predict-only, B = 1 048 576, fp32, dt = 1e-3, random IMU samples.  Properties:
    finite, P symmetric, trace grows by the random-walk variance only; 64 sampled instances follow
    the fp64 oracle to fp32 accuracy (stated tolerance: 1e-4 relative on x, 1e-3 on P)."""
    B = 1 << 20
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    f = _filter(cfg, Q, R, sa, so, B, False, dtype=torch.float32)
    f.set_state(x=ri.x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    from dvi_ekf_amd import noise
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, False, sa, so, Q, R)
    sample = np.linspace(0, B - 1, 64).astype(np.int64)
    o = c_oracle.OracleBatch(prm, np.tile(ri.x0, (64, 1)), ri.P0, ri.om0, ri.acc0)
    for step in range(4):
        z = noise.randn(3 << 32, 0, B, step, 1, 6, dtype=torch.float32, device="cuda")[0]   # seed 3
        om = (z[:, 0:3] * 0.05).contiguous()
        acc = z[:, 3:6].contiguous()
        f.predict(1e-3, om, acc)
        o.predict(1e-3, om[sample].double().cpu().numpy(), acc[sample].double().cpu().numpy())
    st = f.get_state(("x", "P"))
    assert bool(torch.isfinite(st["x"]).all()) and bool(torch.isfinite(st["P"]).all())
    assert bool((st["P"] == st["P"].transpose(1, 2)).all())
    xs, Ps = st["x"][sample].double().cpu().numpy(), st["P"][sample].double().cpu().numpy()
    assert np.abs(xs - o.x).max() <= 1e-4 * np.abs(o.x).max()
    assert np.abs(Ps - o.P).max() <= 1e-3 * np.abs(o.P).max()


def test_config5_long_horizon_epochs_fp64_and_fp32():
    """config #5 shape: epochs of 33/33/34 predicts (dt = 1e-3) + one update, B = 8192.  The fp64
    run is compared with the oracle on 16 sampled instances for 6 epochs (200 predicts); the fp32
    run must stay finite with status 0."""
    B = 8192
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(1,) * 6)
    rng = np.random.default_rng(5)
    sample = np.linspace(0, B - 1, 16).astype(np.int64)
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, False, sa, so, Q, R)
    o = c_oracle.OracleBatch(prm, np.tile(ri.x0, (16, 1)), ri.P0, ri.om0, ri.acc0)
    f64 = _filter(cfg, Q, R, sa, so, B, False)
    f32 = _filter(cfg, Q, R, sa, so, B, False, dtype=torch.float32)
    for f in (f64, f32):
        f.set_state(x=ri.x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    for ep in range(6):
        n = (33, 33, 34)[ep % 3]
        dt = np.full(n, 1e-3)
        om = ri.om[ep] + rng.normal(0, 1e-3, (n, 3))
        acc = ri.acc[ep] + rng.normal(0, 1e-2, (n, 3))
        s64, _ = f64.epoch(dt, om, acc, ri.pos[ep], ri.quat[ep], ri.notch[ep])
        s32, _ = f32.epoch(dt, om, acc, ri.pos[ep], ri.quat[ep], ri.notch[ep])
        for i in range(n):
            o.predict(dt[i], om[i], acc[i])
        o.update(ri.pos[ep], ri.quat[ep], ri.notch[ep])
        assert int(s64.abs().max()) == 0 and int(s32.abs().max()) == 0
    x64 = f64.x
    assert bool((x64 == x64[0:1]).all())
    assert np.abs(x64[sample].cpu().numpy() - o.x).max() <= 1e-9
    assert bool(torch.isfinite(f32.x).all()) and bool(torch.isfinite(f32.P).all())


def test_config4_monte_carlo_driver_single_gpu():
    """config #4 on one GPU (the multi-GPU run only adds the all-gather, covered by
    tests/test_dist_gloo.py): noise streams are reproducible from the seed, instance 0 is the
    noise-free reference run, and the DOF-MSE statistic is finite."""
    from dvi_ekf_amd.montecarlo import run_monte_carlo
    a = run_monte_carlo(4096, frames=6, seed=4)
    b = run_monte_carlo(4096, frames=6, seed=4)
    assert a == b
    assert a["instances_ok"] + a["instances_flagged"] <= 4096 and a["instances_ok"] > 3000
    assert np.isfinite(a["dof_mse_mean"]) and a["dof_mse_mean"] > 0
    # instance 0 = the noise-free, unperturbed run with unfrozen DOFs: must equal the oracle's estimate
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=False, frozen=(0,) * 6, interframe=1)
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, True, sa, so, Q, R)
    o = c_oracle.OracleBatch(prm, ri.x0[None, :], ri.P0, ri.om0, ri.acc0)
    for k in range(6):
        o.predict(ri.dt[k], ri.om[k], ri.acc[k])
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
    assert np.abs(np.array(a["reference_instance_dofs"]) - o.x[0, 10:16]).max() <= 1e-9
    c = run_monte_carlo(4096, frames=6, seed=5)
    assert c["dof_mse_mean"] != a["dof_mse_mean"]


def test_config4_noisy_monte_carlo_instances_forced_step_vs_oracle_and_adjudicator():
    """config #4's actual content -- instances with their own noise on gyro, accelerometer, camera position, camera
    attitude and notch angle and their own initial extrinsics (dvi_ekf_amd.workloads.monte_carlo_pairs, the bench
    workload; counter-based Philox noise, seed 4) -- against the oracle: 256 instances, 8 predict+update pairs, every
    pair-step forced from the oracle's state and adjudicated by the extended-precision build (as
    test_config2_batch1024_perturbed_instances_vs_oracle does for the noise-free inputs); plus the free-running GPU
    run against the free-running oracle wherever the reference filter stays in its physical range."""
    from dvi_ekf_amd.workloads import monte_carlo_pairs
    from tests.test_gpu_parity import adjudicate, innovation_conditions
    B, H = 256, 8
    dev = torch.device("cuda", 0)
    wl = monte_carlo_pairs(B, horizon=H, seed=4, device=dev)
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, True, wl.sigma_a, wl.sigma_om, wl.Q, wl.R)
    x0, P0, om0, acc0 = (t.cpu().numpy() for t in (wl.x0, wl.P0, wl.om0, wl.acc0))
    o = c_oracle.OracleBatch(prm, x0, P0, om0, acc0)
    h = c_oracle.OracleBatch(prm, x0, P0, om0, acc0, hp=True)
    f = wl.make_filter()
    forced = wl.make_filter()
    inp = [[t[k].cpu().numpy() for t in (wl.dt, wl.om, wl.acc, wl.pos, wl.quat, wl.notch)] for k in range(H)]
    assert np.abs(inp[1][1] - inp[1][1][0]).max() > 0 and np.abs(inp[1][3] - inp[1][3][0]).max() > 0     # per-instance noise is there
    sane = np.ones(B, dtype=bool)
    worst_free, closest = 0.0, (0.0, 0.0)
    q_old = o.x[:, 6:10].copy()
    R = np.asarray(wl.R, dtype=float)
    for k in range(H):
        dt, om, acc, pos, quat, notch = inp[k]
        f.step(wl.dt[k], wl.om[k], wl.acc[k], wl.pos[k], wl.quat[k], wl.notch[k])
        o.P[:] = np.tril(o.P) + np.transpose(np.tril(o.P, -1), (0, 2, 1))       # (the device stores the lower triangle)
        forced.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
        h.x[:], h.P[:], h.om_old[:], h.acc_old[:], h.Rwb[:] = o.x, o.P, o.om_old, o.acc_old, o.Rwb
        st, _ = forced.step(wl.dt[k], wl.om[k], wl.acc[k], wl.pos[k], wl.quat[k], wl.notch[k])
        for orc in (o, h):
            orc.predict(dt, om, acc)
        q_old = o.x[:, 6:10].copy()
        kappa = innovation_conditions(o.P, R)
        for orc in (o, h):
            orc.update(pos, quat, notch)
        assert int(st.abs().max()) == 0 and int(np.abs(o.status).max()) == 0
        s = forced.get_state(("x", "P"))
        xe, Pe, xg, xr, Pg, Pr = adjudicate(s["x"].cpu().numpy(), s["P"].cpu().numpy(), o, h, kappa)
        i = int(np.argmax(xe))
        assert xe.max() <= 0 and Pe.max() <= 0, f"pair {k} instance {i}: x gpu {xg[i]:.1e} ref {xr[i]:.1e} cond {kappa[i]:.1e}"
        if xg.max() > closest[0]:
            closest = (float(xg.max()), float(xr[int(np.argmax(xg))]))
        sane &= np.abs(o.x).max(axis=1) < 100.0
        err = np.abs(f.x.cpu().numpy() - o.x).max(axis=1) / np.maximum(1.0, np.abs(o.x).max(axis=1))
        if k < 5:
            worst_free = max(worst_free, float(err[sane].max()))
    print(f"config #4, {B} noisy instances: all adjudicated on all {H} pair-steps (largest GPU distance from exact {closest[0]:.1e}, "
          f"fp64 oracle there {closest[1]:.1e}); free-running: {int(sane.sum())} stay in range over {H} pairs, max rel |dx| of those "
          f"over the first 5 pairs {worst_free:.1e}")
    assert sane.sum() >= B // 4 and worst_free <= 1e-9


def test_config4_full_batch_monte_carlo_is_shard_invariant():
    """config #4 at its full size: the calibration estimates of the B = 65 536 noisy Monte-Carlo batch after 6
    pair-steps, computed as one batch, equal what four shards of 16 384 compute for their instances (the shards of a
    multi-GPU run: own noise streams from (seed, global instance), other grid sizes, other wavefront neighbours) --
    bit for bit; and the run enqueued as ONE launch (dvi_eskf_run_epochs, the time loop in the kernel) gives the same bits
    again, at the full size and for a shard."""
    from dvi_ekf_amd.workloads import monte_carlo_pairs
    B, H, S = 65536, 6, 4
    dev = torch.device("cuda", 0)

    def run(n, first, one_launch=False):
        wl = monte_carlo_pairs(n, horizon=H, seed=4, device=dev, first_instance=first)
        f = wl.make_filter()
        if one_launch:      # the whole run as ONE launch: the time loop in the kernel (eskf_run_kernel)
            st, _ = f.run_epochs([1] * H, wl.dt, wl.om, wl.acc, wl.pos, wl.quat, wl.notch)
            assert int(st.max()) == 0
        else:
            for k in range(H):
                st, _ = f.step(wl.dt[k], wl.om[k], wl.acc[k], wl.pos[k], wl.quat[k], wl.notch[k])
        return f.calibration().clone(), f.x.clone()

    dofs, x = run(B, 0)
    assert len(torch.unique(dofs[:, 0])) > B // 2                 # the instances really differ
    for r in range(S):
        d_r, x_r = run(B // S, r * (B // S))
        sl = slice(r * (B // S), (r + 1) * (B // S))
        assert torch.equal(d_r, dofs[sl]) and torch.equal(x_r, x[sl]), r
    # ... and what ONE launch computes for the full batch (64-instance blocks stepping their instances through the run) and
    # for a shard (32-instance blocks): three launch paths and three block sizes, the same bits
    d1, x1 = run(B, 0, one_launch=True)
    assert torch.equal(d1, dofs) and torch.equal(x1, x)
    d2, x2 = run(B // S, B // S, one_launch=True)
    assert torch.equal(d2, dofs[B // S:2 * (B // S)]) and torch.equal(x2, x[B // S:2 * (B // S)])


def test_random_covariances_row_resident_vs_lds_image_vs_oracle():
    """4096 instances with random SPD covariances (not reachable from the filter's own history) through one
    predict+update pair: the two covariance kernels -- measurement-by-measurement update on rows in
    registers vs explicit LDL^T on an LDS image -- agree to rounding x conditioning, and a subset agrees
    with the C oracle's explicit-inverse gain."""
    from oracle import c_oracle
    from tests.test_gpu_parity import _make, _perturbed_x0, _rel
    B = 4096
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    rng = np.random.default_rng(123)
    x0 = _perturbed_x0(ri, B)
    A = rng.standard_normal((B, 24, 24))
    scale = 10.0 ** rng.uniform(-3, 0, (B, 24, 1))              # rows of very different size
    A = A * scale
    P0 = A @ A.transpose(0, 2, 1) / 24 + 1e-6 * np.eye(24)
    fa = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=1)
    fb = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=32)
    for f in (fa, fb):
        f.set_state(P=P0)
        f.predict(ri.dt[0], ri.om[0], ri.acc[0])
        st, _ = f.update(ri.pos[0], ri.quat[0], ri.notch[0])
        assert int(st.max()) == 0
    a, b = fa.get_state(), fb.get_state()
    for key, tol in (("x", 1e-9), ("P", 1e-9)):
        s = a[key].abs().amax(dim=tuple(range(1, a[key].dim())), keepdim=True)
        d = float(((a[key] - b[key]).abs() / s).max())
        print(f"random SPD, rows vs LDS image: {key} max rel {d:.2e}")
        assert d <= tol, key
    n = 64
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, False, sa, so, Q, R)
    o = c_oracle.OracleBatch(prm, x0[:n], ri.P0, ri.om0, ri.acc0)
    o.P[:] = P0[:n]
    o.predict(ri.dt[0], ri.om[0], ri.acc[0])
    o.update(ri.pos[0], ri.quat[0], ri.notch[0])
    ex = max(_rel(a["x"][i].cpu().numpy(), o.x[i], 1) for i in range(n))
    eP = max(_rel(a["P"][i].cpu().numpy(), o.P[i], 1) for i in range(n))
    print(f"random SPD, rows vs oracle: x {ex:.2e} P {eP:.2e}")
    assert ex <= 1e-8 and eP <= 1e-8


def test_eight_million_instances_address_range():
    """B = 2^23 fp64: the packed covariance record (B x 300 words = 20 GB) lies beyond 2^31 elements
    and 2^34 bytes, so every 32-bit shortcut in the kernels' addressing would show here.  The batch is
    a 96-instance pattern repeated; after two predict+update pairs (the second one goes through the P the
    first one stored) instances at the start, around the 2^31-element boundary, and at the very end
    must equal the 96-instance batch bit for bit."""
    B, M = 1 << 23, 96
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 120e9:
        pytest.skip("needs ~100 GB of device memory")
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False, frozen=(0,) * 6)
    rng = np.random.default_rng(23)
    xm = np.tile(ri.x0, (M, 1))
    xm[1:, 10:13] += rng.normal(0, np.radians(3.0), (M - 1, 3))
    xm[1:, 13:16] += rng.normal(0, 5.0, (M - 1, 3))
    small = _filter(cfg, Q, R, sa, so, M, False)
    small.set_state(x=xm, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    big = _filter(cfg, Q, R, sa, so, B, False)
    reps = (B + M - 1) // M
    xb = torch.as_tensor(xm, device="cuda").repeat(reps, 1)[:B].contiguous()
    big.set_state(x=xb, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    del xb
    torch.cuda.empty_cache()
    for k in range(2):
        sb, _ = big.step(ri.dt[k], ri.om[k], ri.acc[k], ri.pos[k], ri.quat[k], ri.notch[k])
        ss, _ = small.step(ri.dt[k], ri.om[k], ri.acc[k], ri.pos[k], ri.quat[k], ri.notch[k])
    assert int(sb.abs().max()) == 0 and int(ss.abs().max()) == 0
    xs, xbig = small.x, big.x
    boundary = (1 << 31) // 300                       # first instance whose record starts beyond 2^31 elements
    for start in (0, boundary - 200, boundary + 200, B // 2, B - 2 * M, B - M - 1):
        start -= start % M                            # align to the pattern
        assert torch.equal(xbig[start:start + M], xs), start
    tail = B - (B // M) * M                           # the last, partial repetition
    assert torch.equal(xbig[B - tail:], xs[:tail])
    assert torch.equal(big.calibration()[B - tail:], small.calibration()[:tail])
    assert bool(torch.isfinite(xbig).all())


def test_full_batch_size_independent_properties():
    """B = 65 536 (the metric's batch), two properties that need no oracle at that size:

    * an update whose measurement noise dwarfs the covariance (R = 1e30 I) is a no-op: the gain is ~1e-30, so the
      covariance comes back bit for bit and the state to rounding (the quaternions are re-normalised);
    * what a shard computes does not depend on the batch it is computed in: instances [0, 32 768) and
      [32 768, 65 536) stepped as two batches equal the same instances stepped as one, bit for bit, after three
      predict+update pairs (different grid sizes, different wavefront neighbours, same arithmetic)."""
    from tests.test_gpu_parity import _random_configuration
    B = 65536
    dev = torch.device("cuda", 0)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    rng = np.random.default_rng(65536)
    x0 = _random_configuration(rng, B)
    for a in (6, 22):                                   # (w >= 0, as every quaternion the filter itself produces)
        x0[x0[:, a + 3] < 0, a:a + 4] *= -1
    A = rng.standard_normal((B, 24, 24)).astype(np.float64) * 10.0 ** rng.uniform(-2, 0, (B, 24, 1))
    P0 = A @ A.transpose(0, 2, 1) / 24 + 1e-4 * np.eye(24)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    tx, tP = t(x0), t(P0)
    del A, P0
    pos, quat = t(x0[:, 19:22] + rng.normal(0, 1.0, (B, 3))), t(x0[:, 22:26] + rng.normal(0, 0.05, (B, 4)))
    notch = t(x0[:, 16] + rng.normal(0, 0.01, B))
    om_old, acc_old = t(rng.normal(0, 0.5, (B, 3))), t(rng.normal(0, 20.0, (B, 3)))

    # --- 1. R -> infinity
    f = _filter(cfg, Q, 1e30 * np.eye(7), sa, so, B, False)
    f.set_state(x=tx, P=tP, om_old=om_old, acc_old=acc_old)
    before = f.get_state(("x", "P"))
    st, _ = f.update(pos, quat, notch)
    assert int((st & 1).max()) == 0
    after = f.get_state(("x", "P"))
    assert torch.equal(before["P"], after["P"])
    assert float((before["x"] - after["x"]).abs().max()) <= 1e-13
    del before, after, f

    # --- 2. shard invariance
    dts = [t(rng.uniform(0.05, 0.2, B)) for _ in range(3)]
    oms = [t(rng.normal(0, 0.5, (B, 3))) for _ in range(3)]
    accs = [t(rng.normal(0, 20.0, (B, 3))) for _ in range(3)]

    def run(lo, hi):
        g = _filter(cfg, Q, R, sa, so, hi - lo, False)
        g.set_state(x=tx[lo:hi], P=tP[lo:hi], om_old=om_old[lo:hi], acc_old=acc_old[lo:hi])
        for k in range(3):
            g.step(dts[k][lo:hi], oms[k][lo:hi], accs[k][lo:hi], pos[lo:hi], quat[lo:hi], notch[lo:hi])
        return g.get_state(("x", "P"))

    whole = run(0, B)
    for lo, hi in ((0, B // 2), (B // 2, B)):
        part = run(lo, hi)
        assert torch.equal(part["x"], whole["x"][lo:hi]) and torch.equal(part["P"], whole["P"][lo:hi])
    assert bool(torch.isfinite(whole["x"]).all())
