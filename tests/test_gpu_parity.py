"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference's saved trajectory.  All tests here need a real MI355X.

Tolerances (fp64):
  * forced single step (GPU and oracle start the step from the same state):
      state  <= 1e-11 * max(1,|x|),  covariance <= 1e-10 * max(1,|P|), gain <= 1e-8 * max(1,|K|)
    -- the gain bound reflects cond(S) ~ 1e10.
  * free-running trajectory: <= 1e-9 absolute on the state (BASELINE.json north_star)
    over horizons on which the reference filter itself is not yet chaotic
    (DESIGN.md "Numerical behaviour": rounding differences are amplified
    exponentially by the reference's own dynamics after ~40 frames).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import c_oracle
from tests.helpers import default_run, load_golden, log_row

pytestmark = pytest.mark.gpu

PERT = np.array([0.05, -0.03, 0.08, 3.0, -5.0, 4.0])


def _make(cfg, ri, Q, R, sa, so, B, hold, x0=None, dtype=None, lpi=None):
    from dvi_ekf_amd.batch import BatchedESKF
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa,
                    sigma_om=so, hold_frozen=hold, dtype=dtype or torch.float64)
    f.set_noise(Q, R)
    x0 = np.tile(ri.x0, (B, 1)) if x0 is None else x0
    f.set_state(x=x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    if lpi:
        f.set_lanes_per_instance(lpi)
    return f


def _oracle(cfg, ri, Q, R, sa, so, B, hold, x0=None, hp=False):
    """hp=True: the extended-precision (quad) build of the oracle -- the adjudicator"""
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, hold, sa, so, Q, R)
    x0 = np.tile(ri.x0, (B, 1)) if x0 is None else x0
    return c_oracle.OracleBatch(prm, x0, ri.P0, ri.om0, ri.acc0, hp=hp)


def _perturbed_x0(ri, B, seed=1000):
    rng = np.random.default_rng(seed)
    x0 = np.tile(ri.x0, (B, 1))
    x0[1:, 10:13] += rng.normal(0, np.radians(5.0), (B - 1, 3))
    x0[1:, 13:16] += rng.normal(0, 10.0, (B - 1, 3))
    return x0


def _rel(a, b, B):
    a = np.asarray(a).reshape(B, -1)
    b = np.asarray(b).reshape(B, -1)
    return (np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))).max()


def test_saved_trajectory_reproduced_on_gpu():
    """config #1: default config.yaml, mandala0_mono, golden-compat plumbing."""
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _make(cfg, ri, Q, R, sa, so, 3, True)
    o = _oracle(cfg, ri, Q, R, sa, so, 1, True)
    rows = [log_row(ri.t0, f.x[0].cpu().numpy())]
    worst = 0.0
    for k in range(len(ri.t_meas)):
        f.predict(ri.dt[k], ri.om[k], ri.acc[k])
        st, _ = f.update(ri.pos[k], ri.quat[k], ri.notch[k])
        o.predict(ri.dt[k], ri.om[k], ri.acc[k])
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
        x = f.x.cpu().numpy()
        assert (st.cpu().numpy() == 0).all()
        assert np.array_equal(x[0], x[1]) and np.array_equal(x[0], x[2])   # identical instances agree bitwise
        worst = max(worst, np.abs(x[0] - o.x[0]).max())
        rows.append(log_row(ri.t_meas[k], x[0]))
    assert np.abs(np.array(rows) - load_golden()).max() <= 5.0e-10 + 1e-11
    assert worst <= 1e-9


SEL = [18, 19, 20, 21, 22, 23, 15]


def _innovation_condition(P, R):
    """cond_2 of the diagonally equilibrated innovation covariance S = H P H^T + R."""
    S = P[np.ix_(SEL, SEL)] + R
    d = 1.0 / np.sqrt(np.abs(np.diag(S)))
    return np.linalg.cond(S * np.outer(d, d)), S


# Adjudicated comparison of an update (every instance, whatever its conditioning).  The reference forms
# K = P H^T inv(S) with an explicit fp64 inverse; for an ill-conditioned S that result is itself far from
# the exact one, so "equal to the fp64 oracle" stops being the right question.  The __float128 build of
# the oracle (oracle/eskf_oracle.c, -DESKF_ORACLE_QUAD) gives the exact result of the reference's
# formulas from the same inputs; an fp64 implementation passes when its distance from that is within
#     FLOOR + max(C_REF * distance of the literal fp64 oracle from it,  C_COND * cond(S) * eps)
# -- as accurate as the reference's own arithmetic (times 2), or as the forward-error bound of a backward-stable
# fp64 solve allows (cond = 2-norm condition of the diagonally equilibrated innovation covariance, eps = 2^-52;
# the measurement-by-measurement update of the row-resident kernel is an LDL^T solve without pivoting),
# whichever is larger -- and never worse than that; FLOOR is the well-conditioned tolerance of this file's
# header.  All three start from the same SYMMETRIC covariance (see the test): the exact update of an
# ill-conditioned instance moves by 1e-8 when the last bit of P's asymmetry is dropped, which would otherwise
# be charged to whoever stores P packed.
C_REF, C_COND = 2.0, 1.0          # (measured: never more than 0.3 cond(S) eps over every update of this file and of test_gpu_scale.py)
FLOOR_X, FLOOR_P = 1e-11, 1e-10
EPS = 2.0 ** -52


def _rel_rows(a, b):
    """per instance: max |a - b| / max(1, |b|)"""
    B = a.shape[0]
    a, b = np.asarray(a).reshape(B, -1), np.asarray(b).reshape(B, -1)
    return np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))


def adjudicate(x_gpu, P_gpu, o, h, kappa):
    """o: fp64 oracle after the update, h: quad oracle after the same update from the same input, kappa [B]:
    equilibrated cond(S) of the update's input.  Returns per-instance (x excess, P excess, x error of the GPU,
    x error of the fp64 oracle, the same for P); an excess <= 0 passes."""
    ex_g, ex_r = _rel_rows(x_gpu, h.x), _rel_rows(o.x, h.x)
    eP_g, eP_r = _rel_rows(P_gpu, h.P), _rel_rows(o.P, h.P)
    bound = np.asarray(kappa, dtype=float) * EPS
    return (ex_g - (FLOOR_X + np.maximum(C_REF * ex_r, C_COND * bound)), eP_g - (FLOOR_P + np.maximum(C_REF * eP_r, C_COND * bound)),
            ex_g, ex_r, eP_g, eP_r)


def innovation_conditions(P, R):
    return np.array([_innovation_condition(P[b], R)[0] for b in range(P.shape[0])])


@pytest.mark.parametrize("golden,frozen,nframes", [(False, (0,) * 6, 30), (False, (1,) * 6, 12), (True, (0,) * 6, 20),
                                                   (False, (1, 0, 1, 0, 0, 1), 12)])
def test_forced_single_step_parity(golden, frozen, nframes):
    """Every predict and every update starts from the oracle's state on both sides.  EVERY update of
    EVERY instance is checked, through both covariance kernels (the row-resident one the benchmark times
    and the LDS-image one that also forms the gain matrix):

      * equilibrated cond(S) <= 1e4: against the fp64 oracle, tolerances of the header;
      * all of them, ill-conditioned ones included (golden-compat 1:1 schedule with unfrozen DOFs, frame 9
        onwards: |x| ~ 1e3, cond(S) up to 1e6 equilibrated, 1e10 raw): against the extended-precision
        adjudicator, see ``adjudicate`` -- no instance is skipped;
      * the gain matrix: backward stability K S = P H^T everywhere, and against the oracle's K where
        cond(S) <= 1e4."""
    B = 16
    cfg, ri, Q, R, sa, so = default_run(golden_compat=golden, fast=False, frozen=frozen,
                                        interframe=(1 if golden else None))
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, golden, x0)
    o = _oracle(cfg, ri, Q, R, sa, so, B, golden, x0)
    h = _oracle(cfg, ri, Q, R, sa, so, B, golden, x0, hp=True)
    q_old = o.x[:, 6:10].copy()
    ip = 0
    ex = eP = eK = eres = 0.0
    worst = {"x_excess": -1.0, "P_excess": -1.0, "x_gpu": 0.0, "x_ref": 0.0, "P_gpu": 0.0, "P_ref": 0.0, "kappa": 0.0,
             "ratio": 0.0, "kmax": 0.0}
    n_cmp = n_all = 0
    for k in range(nframes):
        for _ in range(ri.n_pred[k]):
            f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
            f.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            o.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            q_old = o.x[:, 6:10].copy()
            ip += 1
            s = {k_: v.cpu().numpy() for k_, v in f.get_state().items()}
            ex = max(ex, _rel(s["x"], o.x, B))
            eP = max(eP, _rel(s["P"], o.P, B))
            assert np.abs(s["om_old"] - o.om_old).max() == 0
            assert np.abs(s["q_old"] - o.x[:, 6:10]).max() < 1e-14
        # The device stores P symmetric-packed (the lower triangle); the oracle's P = Fx P Fx^T + ... carries a
        # rounding-level asymmetry (~1e-16 |P|).  Where S is ill-conditioned the EXACT update is sensitive to
        # that last bit (measured: cond 1.6e6 -> 1e-8 in the state), so all three -- GPU, fp64 oracle, quad
        # oracle -- are given the same, symmetric input: the lower triangle mirrored.
        o.P[:] = np.tril(o.P) + np.transpose(np.tril(o.P, -1), (0, 2, 1))
        P_in = o.P.copy()
        f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
        h.x[:], h.P[:] = o.x, o.P                       # the adjudicator starts from the same input
        g = f.clone()
        st, _ = f.update(ri.pos[k], ri.quat[k], ri.notch[k])
        stg, K = g.update(ri.pos[k], ri.quat[k], ri.notch[k], want_gain=True)
        K = K.cpu().numpy()
        assert (st.cpu().numpy() == 0).all() and (stg.cpu().numpy() == 0).all()
        s = {k_: v.cpu().numpy() for k_, v in f.get_state().items()}
        sg = {k_: v.cpu().numpy() for k_, v in g.get_state().items()}
        Ko = np.empty((B, 24, 7))
        for b in range(B):
            sb, Ko[b] = o.update_one(b, ri.pos[k], ri.quat[k], ri.notch[k])
            assert sb == 0
        assert (h.update(ri.pos[k], ri.quat[k], ri.notch[k]) == 0).all()
        kappa = np.array([_innovation_condition(P_in[b], R)[0] for b in range(B)])
        for name, sk in (("rows", s), ("lds", sg)):
            xe, Pe, xg, xr, Pg, Pr = adjudicate(sk["x"], sk["P"], o, h, kappa)
            worst["ratio"] = max(worst["ratio"], float((np.maximum(xg, Pg) / (kappa * EPS))[kappa > 1e3].max(initial=0.0)))
            worst["kmax"] = max(worst["kmax"], float(kappa.max()))
            i = int(np.argmax(xe))
            if xe[i] > worst["x_excess"]:
                worst.update(x_excess=xe[i], x_gpu=xg[i], x_ref=xr[i], kappa=kappa[i])
            j = int(np.argmax(Pe))
            if Pe[j] > worst["P_excess"]:
                worst.update(P_excess=Pe[j], P_gpu=Pg[j], P_ref=Pr[j])
            assert xe.max() <= 0 and Pe.max() <= 0, \
                f"{name} frame {k}: x gpu {xg[i]:.1e} ref {xr[i]:.1e} cond {kappa[i]:.1e} | P gpu {Pg[j]:.1e} ref {Pr[j]:.1e} cond {kappa[j]:.1e}"
        for b in range(B):
            S = P_in[b][np.ix_(SEL, SEL)] + R
            PH = P_in[b][:, SEL]
            eres = max(eres, (np.abs(K[b] @ S - PH) / (np.abs(K[b]) @ np.abs(S) + np.abs(PH) + 1e-300)).max())
            n_all += 1
            if kappa[b] <= 1e4:
                n_cmp += 1
                ex = max(ex, _rel(s["x"][b], o.x[b], 1), _rel(sg["x"][b], o.x[b], 1))
                eP = max(eP, _rel(s["P"][b], o.P[b], 1), _rel(sg["P"][b], o.P[b], 1))
                eK = max(eK, _rel(K[b], Ko[b], 1))
    print(f"forced parity golden={golden} frozen={frozen}: x {ex:.2e}  P {eP:.2e}  K {eK:.2e}  "
          f"solve residual {eres:.2e}  well-conditioned {n_cmp}/{n_all} updates; adjudicated all {n_all}: "
          f"closest call x: gpu {worst['x_gpu']:.1e} vs fp64 oracle {worst['x_ref']:.1e} from exact (cond {worst['kappa']:.1e}), "
          f"P: gpu {worst['P_gpu']:.1e} vs {worst['P_ref']:.1e}; largest cond {worst['kmax']:.1e}, "
          f"largest gpu error / (cond eps) where cond > 1e3: {worst['ratio']:.1f}")
    assert n_cmp > 0                        # (the direct fp64 comparisons above are maxima over a non-empty set)
    assert ex <= 1e-11 and eP <= 1e-10 and eK <= 1e-8 and eres <= 1e-10


def test_packed_covariance_sensitivity_to_the_oracle_s_raw_asymmetric_P():
    """What the symmetric-packed storage of P costs, pinned instead of argued (parity caveat 2).  The
    golden-compat 1:1 run with unfrozen DOFs, 20 frames (cond(S) up to ~2e6, |x| up to 1e3): at every update the
    fp64 oracle and the extended-precision adjudicator keep the oracle's RAW predicted covariance -- `Fx P Fx^T` in
    plain triple loops, asymmetric at the 1e-16 |P| level -- while the GPU is handed that same matrix and stores its
    lower triangle.  The exact update of an ill-conditioned instance is sensitive to that last bit, so the GPU's
    distance from the adjudicator is no longer bounded by cond(S) eps times a small constant; measured on this run
    (the raw P is asymmetric at 9e-13 of its largest entry): x up to 1.2e-8 from the exact update, up to 98 cond(S) eps
    where cond(S) > 1e3, 2.5e-12 where cond(S) <= 1e4.  Asserted here: the distance stays below
    300 cond(S) eps + the well-conditioned floor for every instance and frame, and below the 1e-9 target of the path
    wherever cond(S) <= 1e4."""
    B = 16
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=False, frozen=(0,) * 6, interframe=1)
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, True, x0)
    o = _oracle(cfg, ri, Q, R, sa, so, B, True, x0)
    h = _oracle(cfg, ri, Q, R, sa, so, B, True, x0, hp=True)
    q_old = o.x[:, 6:10].copy()
    worst_ratio, worst_x, worst_asym, worst_sane = 0.0, 0.0, 0.0, 0.0
    for k in range(20):
        o.predict(ri.dt[k], ri.om[k], ri.acc[k])                    # (1:1 schedule: one predict per frame)
        q_old_next = o.x[:, 6:10].copy()
        P_raw = o.P.copy()                                           # NOT symmetrised
        worst_asym = max(worst_asym, float((np.abs(P_raw - np.transpose(P_raw, (0, 2, 1))).reshape(B, -1).max(axis=1)
                                            / np.abs(P_raw).reshape(B, -1).max(axis=1)).max()))
        f.set_state(x=o.x, P=P_raw, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old_next)
        h.x[:], h.P[:] = o.x, P_raw
        kappa = innovation_conditions(P_raw, R)
        st, _ = f.update(ri.pos[k], ri.quat[k], ri.notch[k])
        assert (st.cpu().numpy() == 0).all()
        assert (h.update(ri.pos[k], ri.quat[k], ri.notch[k]) == 0).all()
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
        s = {k_: v.cpu().numpy() for k_, v in f.get_state(("x", "P")).items()}
        dx, dP = _rel_rows(s["x"], h.x), _rel_rows(s["P"], h.P)
        d = np.maximum(dx, dP)
        assert (d <= FLOOR_P + 300.0 * kappa * EPS).all(), (k, int(np.argmax(d / (kappa * EPS))), d.max(), kappa.max())
        worst_ratio = max(worst_ratio, float((d / (kappa * EPS))[kappa > 1e3].max(initial=0.0)))
        worst_x = max(worst_x, float(dx.max()))
        if (kappa <= 1e4).any():
            worst_sane = max(worst_sane, float(d[kappa <= 1e4].max()))
        q_old = q_old_next
    print(f"raw (asymmetric, {worst_asym:.1e} relative) predicted P handed to the packed storage: GPU up to {worst_x:.1e} from the exact "
          f"update in x, {worst_ratio:.1f} cond(S) eps where cond > 1e3; {worst_sane:.1e} where cond <= 1e4")
    assert worst_sane <= 1e-9


@pytest.mark.parametrize("golden,frozen,nframes,horizon", [(True, (1,) * 6, 40, 30), (False, (1,) * 6, 50, 40),
                                                           (False, (0,) * 6, 40, 20)])
def test_free_running_trajectory_parity(golden, frozen, nframes, horizon):
    """State trajectory max-abs error vs the oracle <= 1e-9 (fp64) with nothing re-synchronised,
    for at least `horizon` camera frames; the run continues to `nframes` to report where the
    reference's own error amplification takes the two apart (Numerical behaviour)."""
    B = 8
    cfg, ri, Q, R, sa, so = default_run(golden_compat=golden, fast=False, frozen=frozen,
                                        interframe=(1 if golden else None))
    x0 = _perturbed_x0(ri, B) if not frozen[0] else None
    f = _make(cfg, ri, Q, R, sa, so, B, golden, x0)
    o = _oracle(cfg, ri, Q, R, sa, so, B, golden, x0)
    ip = 0
    err = []
    for k in range(nframes):
        for _ in range(ri.n_pred[k]):
            f.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            o.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            ip += 1
        f.update(ri.pos[k], ri.quat[k], ri.notch[k])
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
        err.append(np.abs(f.x.cpu().numpy() - o.x).max())
    err = np.maximum.accumulate(np.array(err))
    first_bad = int(np.argmax(err > 1e-9)) if (err > 1e-9).any() else nframes
    print(f"free-running golden={golden} frozen={frozen[0]}: max|dx| after 10/20/30/{nframes} frames = "
          f"{err[9]:.1e}/{err[19]:.1e}/{err[29]:.1e}/{err[-1]:.1e}; <=1e-9 for {first_bad} frames ({ip} predicts total)")
    assert err[horizon - 1] <= 1e-9


def _close(a, b, tol=1e-12):
    for key in a:
        x, y = a[key].cpu().numpy(), b[key].cpu().numpy()
        assert np.abs(x - y).max() <= tol * max(1.0, np.abs(y).max()), key


def test_fused_step_and_epoch_match_separate_launches():
    """epoch(n) == n x predict + update and step == predict + update.  Not bitwise: between
    separate launches P is stored as its lower triangle (re-symmetrised), inside a fused
    launch the full LDS image with its rounding-level asymmetry is carried over."""
    B = 10
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    x0 = _perturbed_x0(ri, B)
    f1 = _make(cfg, ri, Q, R, sa, so, B, False, x0)
    f2 = f1.clone()
    f2b = f1.clone()
    f3 = f1.clone()
    n = int(ri.n_pred[0])
    for i in range(n):
        f1.predict(ri.dt[i], ri.om[i], ri.acc[i])
    s1, K1 = f1.update(ri.pos[0], ri.quat[0], ri.notch[0], want_gain=True)
    s2, K2 = f2.epoch(ri.dt[:n], ri.om[:n], ri.acc[:n], ri.pos[0], ri.quat[0], ri.notch[0], want_gain=True)
    _close(f1.get_state(), f2.get_state())
    _close({"K": K1}, {"K": K2}, 1e-10)
    assert torch.equal(s1, s2)
    # per-instance IMU stream laid out [n,B,...] (bitwise against the broadcast stream through the
    # same kernel: a gain request may be served by the other covariance kernel)
    f2b.epoch(ri.dt[:n], ri.om[:n], ri.acc[:n], ri.pos[0], ri.quat[0], ri.notch[0])
    _close(f2.get_state(), f2b.get_state())
    f5 = _make(cfg, ri, Q, R, sa, so, B, False, x0)
    f5.epoch(np.tile(ri.dt[:n, None], (1, B)), np.tile(ri.om[:n, None, :], (1, B, 1)),
             np.tile(ri.acc[:n, None, :], (1, B, 1)), ri.pos[0], ri.quat[0], ri.notch[0])
    a, b = f2b.get_state(), f5.get_state()
    for key in a:
        assert torch.equal(a[key], b[key]), key
    # 1:1 pair
    f3.predict(ri.dt[0], ri.om[0], ri.acc[0])
    f3.update(ri.pos[0], ri.quat[0], ri.notch[0])
    f4 = _make(cfg, ri, Q, R, sa, so, B, False, x0)
    f4.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    _close(f3.get_state(), f4.get_state())


def test_per_instance_inputs_match_broadcast():
    B = 7
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f1 = _make(cfg, ri, Q, R, sa, so, B, True)
    f2 = _make(cfg, ri, Q, R, sa, so, B, True)
    for k in range(3):
        f1.step(ri.dt[k], ri.om[k], ri.acc[k], ri.pos[k], ri.quat[k], ri.notch[k])
        f2.step(np.full(B, ri.dt[k]), np.tile(ri.om[k], (B, 1)), np.tile(ri.acc[k], (B, 1)),
                np.tile(ri.pos[k], (B, 1)), np.tile(ri.quat[k], (B, 1)), np.full(B, ri.notch[k]))
    a, b = f1.get_state(), f2.get_state()
    for key in a:
        assert torch.equal(a[key], b[key]), key


@pytest.mark.parametrize("B", [1, 8, 23])
def test_row_resident_lane_layouts_are_bitwise_identical(B):
    """Selector 2 = the row-resident kernel with 24 lanes per instance (eight instances per block of three
    wavefronts, block barriers between the exchanges) against selector 1 (32 lanes per instance, one
    wavefront per two instances): the same arithmetic per lane."""
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    x0 = _perturbed_x0(ri, B)
    fa = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=1)
    fb = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=2)
    ip = 0
    for k in range(3):
        for _ in range(ri.n_pred[k]):
            fa.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            fb.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            ip += 1
        sta, _ = fa.update(ri.pos[k], ri.quat[k], ri.notch[k])
        stb, _ = fb.update(ri.pos[k], ri.quat[k], ri.notch[k])
        assert torch.equal(sta, stb)
        a, b = fa.get_state(), fb.get_state()
        for key in a:
            assert torch.equal(a[key], b[key]), (k, key)


@pytest.mark.parametrize("tune", [False, True])
def test_row_resident_kernel_matches_lds_image_kernel(tune):
    """The two covariance kernels (rows of P in registers / full image in LDS) are the same
    arithmetic in a different order: they agree to rounding, also with gyro process noise
    (tune mode, Q[3:6] != 0) and with several predicts per update."""
    B = 23
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    if tune:
        Q = Q.copy()
        Q[3, 3] = Q[4, 4] = Q[5, 5] = 1e-4
    x0 = _perturbed_x0(ri, B)
    fa = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=32)
    fb = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=1)
    ip = 0
    for k in range(3):
        for _ in range(ri.n_pred[k]):
            fa.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            fb.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            ip += 1
        sta, Ka = fa.update(ri.pos[k], ri.quat[k], ri.notch[k], want_gain=True)
        stb, Kb = fb.update(ri.pos[k], ri.quat[k], ri.notch[k], want_gain=True)
        assert torch.equal(sta, stb)
        a, b = fa.get_state(), fb.get_state()
        for key in ("x", "P"):
            scale = a[key].abs().amax(dim=tuple(range(1, a[key].dim())), keepdim=True)
            assert float(((a[key] - b[key]).abs() / scale).max()) < 1e-12, (k, key)
        kscale = Ka.abs().amax(dim=(1, 2), keepdim=True)
        assert float(((Ka - Kb).abs() / kscale).max()) < 1e-11, k


@pytest.mark.parametrize("lpi", [1, 32])
def test_singular_innovation_sets_status_and_leaves_state(lpi):
    """Filter.py:356-361: LinAlgError path -> no change, no exception."""
    B = 5
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _make(cfg, ri, Q, np.zeros((7, 7)), sa, so, B, True, lpi=lpi)
    P0 = np.tile(ri.P0, (B, 1, 1))
    P0[1] = 0.0
    P0[3] = 0.0
    f.set_state(P=P0)
    before = f.get_state()
    st, _ = f.update(ri.pos[0], ri.quat[0], ri.notch[0])
    st = st.cpu().numpy()
    assert list(st & 1) == [0, 1, 0, 1, 0]
    after = f.get_state()
    for key in before:
        assert torch.equal(before[key][1], after[key][1]) and torch.equal(before[key][3], after[key][3])
    assert not torch.equal(before["x"][0], after["x"][0])


@pytest.mark.parametrize("lpi", [1, 32])
def test_singular_after_first_pivot_restores_prediction(lpi):
    """S singular although its first pivot is fine (R = 0, P[19,:] = 0 -> second pivot exactly 0): the
    instance is left as it was (Filter.py:356-361).  The row-resident kernel has applied the first
    measurement by then and takes it back: P is restored to rounding, the state is untouched."""
    B = 4
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _make(cfg, ri, Q, np.zeros((7, 7)), sa, so, B, True, lpi=lpi)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((24, 24))
    Pd = A @ A.T / 24 + np.eye(24)
    P0 = np.tile(Pd, (B, 1, 1))
    P0[2, 19, :] = 0.0
    P0[2, :, 19] = 0.0
    f.set_state(P=P0)
    before = f.get_state()
    st, _ = f.update(ri.pos[0], ri.quat[0], ri.notch[0])
    assert list(st.cpu().numpy() & 1) == [0, 0, 1, 0]
    after = f.get_state()
    assert torch.equal(before["x"][2], after["x"][2])
    dP = (before["P"][2] - after["P"][2]).abs().max() / before["P"][2].abs().max()
    assert float(dP) <= (1e-14 if lpi == 1 else 0.0), float(dP)
    assert not torch.equal(before["x"][0], after["x"][0])


def test_tune_mode_noise_with_gyro_block():
    """Filter.update_noise_matrices with dt != 0 (Simulator.optim_std setter): the IMU
    block of Q is non-zero and couples into the camera rows through Fi."""
    B = 4
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    Q = Q.copy()
    Q[0:3, 0:3] = 0.1 ** 2 * sa ** 2 * np.eye(3)
    Q[3:6, 3:6] = 0.1 ** 2 * so ** 2 * np.eye(3) * 1e4   # exaggerate so the coupling is visible
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, False, x0)
    o = _oracle(cfg, ri, Q, R, sa, so, B, False, x0)
    for i in range(5):
        f.predict(ri.dt[i], ri.om[i], ri.acc[i])
        o.predict(ri.dt[i], ri.om[i], ri.acc[i])
    P = f.P.cpu().numpy()
    assert np.abs(P - o.P).max() <= 1e-10 * np.abs(o.P).max()


def test_clone_is_independent():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = _make(cfg, ri, Q, R, sa, so, 4, True)
    g = f.clone()
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    assert np.array_equal(g.x.cpu().numpy()[0], ri.x0)
    assert not np.array_equal(f.x.cpu().numpy()[0], ri.x0)


def test_fp32_variant_tracks_oracle_loosely():
    """fp32 kernels (BASELINE configs 3/5) cannot meet 1e-9: cond(S) ~ 1e10.  Stated
    tolerance: forced single predict, relative 2e-5 on x and 2e-4 on P."""
    B = 6
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, False, x0, dtype=torch.float32)
    o = _oracle(cfg, ri, Q, R, sa, so, B, False, x0)
    f.predict(ri.dt[0], ri.om[0], ri.acc[0])
    o.predict(ri.dt[0], ri.om[0], ri.acc[0])
    s = f.get_state()
    assert np.abs(s["x"].cpu().numpy() - o.x).max() <= 2e-5 * np.abs(o.x).max()
    assert np.abs(s["P"].cpu().numpy() - o.P).max() <= 2e-4 * np.abs(o.P).max()


# Stated fp32 tolerances of the update / pair-step (measured on MI355X): a forced single
# step from the fp64 oracle's state, relative to the instance's magnitude.  fp32 keeps 7 digits and the
# update subtracts K S K^T from a P that shrinks by up to ten orders of magnitude in its measured block
# (R_theta ~ 3e-14 against a prior of 1e-5), so the covariance tolerance is set by the LARGEST entry of P.
# Tolerances = what the kernels measure on MI355X (deterministic: no atomics, fixed operation order) x 3, per regime -- round 5;
# until then one pair (5e-4, 5e-3) covered all of them and would not have noticed a tenfold regression of the frozen runs.
#   (golden, frozen): (predict x, predict P, update x, update P, fused epoch x, fused epoch P)
FP32_MEASURED = {(False, 0): (1.3e-7, 1.1e-7, 1.9e-4, 1.7e-5, 3.7e-4, 3.4e-5),
                 (False, 1): (1.3e-7, 9.4e-8, 4.0e-7, 7.1e-7, 9.0e-7, 1.4e-6),
                 (True, 0): (1.3e-7, 7.2e-7, 3.7e-6, 3.0e-6, 1.2e-5, 8.0e-6)}


@pytest.mark.parametrize("golden,frozen,lpi", [(False, (0,) * 6, None), (False, (1,) * 6, None), (True, (0,) * 6, None),
                                               (False, (0,) * 6, 32)])
def test_fp32_forced_single_step_update_parity(golden, frozen, lpi):
    """The fp32 update (BASELINE configs #3/#5 compute in fp32) against the fp64 oracle: every update, the
    predicts of an epoch and the fused predict+update pair start from the oracle's state rounded to fp32.
    The row-resident kernel processes the seven measurements one after the other, i.e. every pivot is a scalar
    division -- no 7x7 factorisation whose scaling (R_theta ~ 3e-14 next to R_p ~ 4e-4) would need
    equilibrating; what limits fp32 is the cancellation P - K S K^T, stated above."""
    B = 16
    cfg, ri, Q, R, sa, so = default_run(golden_compat=golden, fast=False, frozen=frozen, interframe=(1 if golden else None))
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, golden, x0, dtype=torch.float32, lpi=lpi)
    o = _oracle(cfg, ri, Q, R, sa, so, B, golden, x0)
    q_old = o.x[:, 6:10].copy()
    ip = 0
    ex_u = eP_u = ex_p = eP_p = ex_s = eP_s = 0.0
    relP = lambda a, b: (np.abs(a - b).reshape(B, -1).max(axis=1) / np.abs(b).reshape(B, -1).max(axis=1)).max()
    for k in range(8):
        xin, Pin, oin, ain, qin = o.x.copy(), o.P.copy(), o.om_old.copy(), o.acc_old.copy(), q_old.copy()
        n = int(ri.n_pred[k])
        for _ in range(n):
            f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
            f.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            o.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            q_old = o.x[:, 6:10].copy()
            ip += 1
            s = f.get_state(("x", "P"))
            ex_p = max(ex_p, _rel(s["x"].double().cpu().numpy(), o.x, B))
            eP_p = max(eP_p, relP(s["P"].double().cpu().numpy(), o.P))
        f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
        st, _ = f.update(ri.pos[k], ri.quat[k], ri.notch[k])
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
        assert (st.cpu().numpy() == 0).all()
        s = f.get_state(("x", "P"))
        ex_u = max(ex_u, _rel(s["x"].double().cpu().numpy(), o.x, B))
        eP_u = max(eP_u, relP(s["P"].double().cpu().numpy(), o.P))
        # the same frame as ONE launch (epoch: n predicts + update) from the frame's input
        f.set_state(x=xin, P=Pin, om_old=oin, acc_old=ain, q_old=qin)
        st, _ = f.epoch(ri.dt[ip - n:ip], ri.om[ip - n:ip], ri.acc[ip - n:ip], ri.pos[k], ri.quat[k], ri.notch[k])
        assert (st.cpu().numpy() == 0).all()
        s = f.get_state(("x", "P"))
        ex_s = max(ex_s, _rel(s["x"].double().cpu().numpy(), o.x, B))
        eP_s = max(eP_s, relP(s["P"].double().cpu().numpy(), o.P))
    print(f"fp32 forced single step (golden={golden}, frozen={frozen[0]}, lanes={lpi}): predict x {ex_p:.1e} P {eP_p:.1e} | "
          f"update x {ex_u:.1e} P {eP_u:.1e} | fused epoch x {ex_s:.1e} P {eP_s:.1e}")
    m = FP32_MEASURED[(golden, frozen[0])]
    for got, meas, what in zip((ex_p, eP_p, ex_u, eP_u, ex_s, eP_s), m, ("predict x", "predict P", "update x", "update P", "epoch x", "epoch P")):
        assert got <= 3.0 * meas, (what, got, meas)


@pytest.mark.parametrize("lpi", [None, 2, 32])
def test_non_finite_measurement_stays_in_its_instance(lpi):
    """A NaN / inf camera measurement propagates into the state of ITS instance exactly as NumPy
    lets it in the reference (Filter.update has no check) and is reported in the status word; every
    other instance of the batch -- including the one sharing its wavefront -- is bit for bit what it
    is without the bad measurement.  The oracle flags the same instances."""
    B = 70
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False, frozen=(0,) * 6)
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=lpi)
    o = _oracle(cfg, ri, Q, R, sa, so, B, False, x0)
    for i in range(3):
        f.predict(ri.dt[i], ri.om[i], ri.acc[i])
        o.predict(ri.dt[i], ri.om[i], ri.acc[i])
    clean = f.clone()
    pos = np.tile(ri.pos[0], (B, 1))
    quat = np.tile(ri.quat[0], (B, 1))
    notch = np.full(B, ri.notch[0])
    bad = {5: "pos", 33: "quat", 64: "notch", 69: "inf"}
    pos_b, quat_b, notch_b = pos.copy(), quat.copy(), notch.copy()
    pos_b[5, 1] = np.nan
    quat_b[33, 2] = np.nan
    notch_b[64] = np.nan
    pos_b[69, 0] = np.inf
    st, _ = f.update(pos_b, quat_b, notch_b)
    stc, _ = clean.update(pos, quat, notch)
    st, stc = st.cpu().numpy(), stc.cpu().numpy()
    good = np.array([b not in bad for b in range(B)])
    assert (stc == 0).all() and (st[good] == 0).all()
    assert all(st[b] & 2 for b in bad), st[list(bad)]
    sf, sc = f.get_state(), clean.get_state()
    for key in ("x", "P"):
        a, c = sf[key].cpu().numpy(), sc[key].cpu().numpy()
        assert np.array_equal(a[good], c[good]), key
    xs = sf["x"].cpu().numpy()
    assert all(not np.isfinite(xs[b]).all() for b in bad)
    o.update(pos_b, quat_b, notch_b)
    assert all(o.status[b] & 2 for b in bad) and (o.status[good] == 0).all()
    assert np.abs(xs[good] - o.x[good]).max() <= 1e-9
    # the next step keeps the damage where it is
    f.step(ri.dt[3], ri.om[3], ri.acc[3], ri.pos[1], ri.quat[1], ri.notch[1])
    clean.step(ri.dt[3], ri.om[3], ri.acc[3], ri.pos[1], ri.quat[1], ri.notch[1])
    assert torch.equal(f.x[torch.as_tensor(good)], clean.x[torch.as_tensor(good)])


def _random_unit_quats(rng, n):
    q = rng.standard_normal((n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def _random_configuration(rng, B):
    x0 = np.zeros((B, 26))
    x0[:, 0:3] = rng.normal(0, 30.0, (B, 3))
    x0[:, 3:6] = rng.normal(0, 5.0, (B, 3))
    x0[:, 6:10] = _random_unit_quats(rng, B)
    x0[:, 10:13] = rng.uniform(-np.pi, np.pi, (B, 3))
    x0[:, 13:16] = rng.normal(0, 20.0, (B, 3))
    x0[:, 16] = rng.uniform(-np.pi, np.pi, B)
    x0[:, 17:19] = rng.normal(0, 0.1, (B, 2))
    x0[:, 19:22] = rng.normal(0, 30.0, (B, 3))
    x0[:, 22:26] = _random_unit_quats(rng, B)
    return x0


@pytest.mark.parametrize("frozen,hold,dt_scale,seed", [((0,) * 6, False, 0.1, 11), ((1,) * 6, True, 1e-3, 12)])
def test_random_configurations_fp32_single_step(frozen, hold, dt_scale, seed):
    """The fp32 kernels at states nowhere near the reference trajectory (all Markley branches, joint angles and
    notch over +-pi, per-instance dt, random SPD covariances): one predict and one update from inputs that are
    exactly representable in fp32, against the fp64 oracle on the same inputs.  Stated tolerance, relative to the
    instance's magnitude: predict x 2e-6, P 1e-5; update x 2e-4, P 5e-4 (measured 1.1e-7 / 8.4e-7 and 1.1e-5 / 2.7e-5;
    quaternions compared up to their sign: w ~ 0 may round to either side of the w >= 0 normalisation)."""
    B = 2048
    rng = np.random.default_rng(2000 + seed)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=frozen)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    x0 = f32(_random_configuration(rng, B))
    A = rng.standard_normal((B, 24, 24)) * 10.0 ** rng.uniform(-1, 0, (B, 24, 1))
    P0 = f32(A @ A.transpose(0, 2, 1) / 24 + 1e-2 * np.eye(24))
    P0 = np.tril(P0) + np.transpose(np.tril(P0, -1), (0, 2, 1))
    om_old, acc_old = f32(rng.normal(0, 0.5, (B, 3))), f32(rng.normal(0, 20.0, (B, 3)))
    om, acc = f32(rng.normal(0, 0.5, (B, 3))), f32(rng.normal(0, 20.0, (B, 3)))
    dt = f32(dt_scale * rng.uniform(0.5, 1.5, B))
    pos = f32(x0[:, 19:22] + rng.normal(0, 1.0, (B, 3)))
    quat = f32(x0[:, 22:26] + rng.normal(0, 0.05, (B, 4)))
    notch = f32(x0[:, 16] + rng.normal(0, 0.01, B))
    Rl = np.diag([1e-2] * 3 + [1e-4] * 3 + [1e-4])           # an innovation covariance fp32 can factorise: see the forced-step test
    f = _make(cfg, ri, Q, Rl, sa, so, B, hold, x0, dtype=torch.float32)
    f.set_state(x=x0, P=P0, om_old=om_old, acc_old=acc_old)
    o = _oracle(cfg, ri, Q, Rl, sa, so, B, hold, x0)
    o.P[:], o.om_old[:], o.acc_old[:] = P0, om_old, acc_old

    def errs():
        s = {k_: v.double().cpu().numpy() for k_, v in f.get_state(("x", "P")).items()}
        x = s["x"].copy()
        for a, b in ((6, 10), (22, 26)):
            flip = (x[:, a:b] * o.x[:, a:b]).sum(axis=1) < 0
            x[flip, a:b] *= -1
        return _rel(x, o.x, B), (np.abs(s["P"] - o.P).reshape(B, -1).max(axis=1) / np.abs(o.P).reshape(B, -1).max(axis=1)).max()

    f.predict(dt, om, acc)
    o.predict(dt, om, acc)
    ex_p, eP_p = errs()
    # (the update starts from the oracle's prediction on both sides)
    f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=o.x[:, 6:10])
    st, _ = f.update(pos, quat, notch)
    o.update(pos, quat, notch)
    assert ((st.cpu().numpy() & 1) == 0).all() and ((o.status & 1) == 0).all()
    ex_u, eP_u = errs()
    print(f"fp32 random configurations frozen={frozen[0]} dt~{dt_scale}: predict x {ex_p:.1e} P {eP_p:.1e}; update x {ex_u:.1e} P {eP_u:.1e}")
    assert ex_p <= 2e-6 and eP_p <= 1e-5
    assert ex_u <= 2e-4 and eP_u <= 5e-4


@pytest.mark.parametrize("frozen,hold,dt_scale,seed", [((0,) * 6, False, 0.1, 1), ((1, 0, 1, 0, 0, 1), False, 1.0, 2),
                                                       ((1,) * 6, True, 1.0, 3), ((0,) * 6, False, 1e-3, 4)])
def test_random_configurations_single_step_parity(frozen, hold, dt_scale, seed):
    """States nowhere near the reference trajectory: random attitudes (all four branches of the
    Markley matrix-to-quaternion conversion, both signs of w), joint angles over +-pi, notch angle and
    rates, random previous / current IMU samples, per-instance dt, measurements up to a half-turn away
    from the state, random SPD covariances -- one predict and one update against the C oracle, every
    instance with its own inputs."""
    B = 2048
    rng = np.random.default_rng(1000 + seed)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=frozen)
    x0 = _random_configuration(rng, B)
    A = rng.standard_normal((B, 24, 24)) * 10.0 ** rng.uniform(-2, 0, (B, 24, 1))
    P0 = A @ A.transpose(0, 2, 1) / 24 + 1e-4 * np.eye(24)
    om_old, acc_old = rng.normal(0, 0.5, (B, 3)), rng.normal(0, 20.0, (B, 3))
    om, acc = rng.normal(0, 0.5, (B, 3)), rng.normal(0, 20.0, (B, 3))
    dt = dt_scale * rng.uniform(0.5, 1.5, B)
    pos = x0[:, 19:22] + rng.normal(0, 1.0, (B, 3))
    quat = _random_unit_quats(rng, B) * rng.uniform(0.5, 2.0, (B, 1))        # raw, un-normalised
    near = rng.random(B) < 0.5                                                # half of them near the camera state
    quat[near] = x0[near, 22:26] + rng.normal(0, 0.05, (near.sum(), 4))
    notch = rng.uniform(-np.pi, np.pi, B)
    notch[near] = x0[near, 16] + rng.normal(0, 0.01, near.sum())
    # a measurement quaternion is turned by the notch angle before it is compared (Filter.py:364-365)
    for lpi in (None, 32):
        f = _make(cfg, ri, Q, R, sa, so, B, hold, x0, lpi=lpi)
        f.set_state(x=x0, P=P0, om_old=om_old, acc_old=acc_old)
        o = _oracle(cfg, ri, Q, R, sa, so, B, hold, x0)
        o.P[:] = P0
        o.om_old[:] = om_old
        o.acc_old[:] = acc_old
        f.predict(dt, om, acc)
        o.predict(dt, om, acc)
        s = {k_: v.cpu().numpy() for k_, v in f.get_state().items()}
        ex_p, eP_p = _rel(s["x"], o.x, B), _rel(s["P"], o.P, B)
        st, _ = f.update(pos, quat, notch)
        o.update(pos, quat, notch)
        st = st.cpu().numpy()
        assert ((st & 1) == 0).all() and ((o.status & 1) == 0).all()
        assert np.array_equal(st & 2, o.status & 2)                            # same instances hit the asin clamp
        s = {k_: v.cpu().numpy() for k_, v in f.get_state().items()}
        ex_u, eP_u = _rel(s["x"], o.x, B), _rel(s["P"], o.P, B)
        w_neg = int((s["x"][:, 9] < 0).sum())
        print(f"random configurations frozen={frozen} hold={hold} dt~{dt_scale} kernel={lpi or 'rows'}: predict x {ex_p:.1e} "
              f"P {eP_p:.1e}; update x {ex_u:.1e} P {eP_u:.1e}; clamped {int((st & 2 != 0).sum())}")
        assert w_neg == 0                                                       # normalisation keeps w >= 0 (Quaternion.py:195-206)
        assert ex_p <= 1e-11 and eP_p <= 1e-10
        assert ex_u <= 1e-9 and eP_u <= 1e-8
