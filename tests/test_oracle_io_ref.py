"""oracle/io_ref.py -- the independent checker of SURVEY 8(f1) / 8(f2) -- against the reference's saved files and
against the product's host code (dvi_ekf_amd/imu.py, probe.py, traj.py): three restatements by construction
(numerical DH chain + Jacobian / Hessian here, closed forms in the product and on the device), pairwise <= 1e-12."""
import numpy as np

from dvi_ekf_amd.camera import create_camera
from dvi_ekf_amd.imu import imu_measurement, imu_pose_from_camera
from dvi_ekf_amd.probe import SimpleProbe, probe_fwkin
from dvi_ekf_amd.state import State
from dvi_ekf_amd.traj import euler_measurement_row
from oracle import io_ref
from tests.helpers import IMU_REF, default_run, load_golden


def test_imu_reference_file_reproduced_by_the_io_oracle():
    """imu_ref_mandala0_mono.txt (9 x 14, saved by the reference): f_imu over the camera samples of the default run,
    through the numerical DH chain, to the file's print rounding."""
    cfg, ri, *_ = default_run(golden_compat=True)
    cam = create_camera(cfg)
    probe = SimpleProbe(cfg.model.length, cfg.model.angle)
    ref = np.loadtxt(IMU_REF)
    R = np.array(cam.R)
    rows = []
    for row in ref:
        i = int(np.argmin(np.abs(np.asarray(cam.t) - row[0])))
        assert abs(cam.t[i] - row[0]) < 1e-9
        fk = io_ref.simple_probe_fwkin(probe.q[:6], [0.0, 0.0, 0.0], cfg.model.length, cfg.model.angle)
        p, Rwb, v = io_ref.f_imu(cam.p[:, i], R[i], cam.v[:, i], cam.om[:, i], fk)
        rows.append(io_ref.imu_ref_row(cam.t[i], p, Rwb, v))
    assert np.abs(np.array(rows) - ref).max() <= 5.0e-10 + 1e-11


def test_filter_log_row_of_the_io_oracle_matches_the_saved_trajectory_row_zero():
    """row 0 of kf_best_mandala0_mono.txt is the initial state of the default run"""
    cfg, ri, *_ = default_run(golden_compat=True)
    assert np.abs(io_ref.filter_log_row(ri.t0, ri.x0) - load_golden()[0]).max() <= 5.0e-10 + 1e-11


def test_product_host_code_agrees_with_the_io_oracle():
    rng = np.random.default_rng(3)
    from scipy.spatial.transform import Rotation
    L, th = 50.0, np.radians(30.0)
    worst = 0.0
    for _ in range(200):
        dofs = np.concatenate([rng.uniform(-np.pi, np.pi, 3), rng.normal(0, 20.0, 3)])
        notch = np.array([rng.uniform(-np.pi, np.pi), rng.normal(0, 0.5), rng.normal(0, 0.5)])
        fk_o = io_ref.simple_probe_fwkin(dofs, notch, L, th)
        fk_p = probe_fwkin(dofs, notch, L, th)
        for a, b in zip(fk_o, fk_p):
            worst = max(worst, np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(a)).max()))
        R = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix()
        p, v = rng.normal(0, 50.0, 3), rng.normal(0, 5.0, 3)
        om, acc, alp = rng.normal(0, 1.0, 3), rng.normal(0, 30.0, 3), rng.normal(0, 2.0, 3)
        for a, b in zip(io_ref.f_imu_meas(R, om, acc, alp, fk_o), imu_measurement(R, om, acc, alp, fk_p)):
            worst = max(worst, np.abs(a - b).max() / max(1.0, np.abs(a).max()))
        for a, b in zip(io_ref.f_imu(p, R, v, om, fk_o), imu_pose_from_camera([p, R, v, om, acc, alp], fk_p)):
            worst = max(worst, np.abs(a - b).max() / max(1.0, np.abs(a).max()))
    assert worst <= 1e-12, worst
    # the Hessian contraction with every joint moving: the acceleration of the chain's end by second differences of fkine
    q, qd, qdd = rng.uniform(-1, 1, 8), rng.normal(0, 0.3, 8), rng.normal(0, 0.3, 8)
    h = 1e-4
    pos = lambda s: io_ref.probe_kinematics(q + qd * s + 0.5 * qdd * s * s, qd, qdd, L, th)[0]
    fd_acc = (pos(h) - 2 * pos(0.0) + pos(-h)) / h ** 2
    assert np.abs(fd_acc - io_ref.probe_kinematics(q, qd, qdd, L, th)[4]).max() <= 1e-5
    # log rows: product host row == oracle row (angles up to the +-180 deg wrap)
    for _ in range(200):
        x = np.zeros(26)
        x[:6] = rng.normal(0, 30.0, 6)
        x[6:10] = Rotation.random(random_state=int(rng.integers(1 << 30))).as_quat()
        x[10:13] = rng.uniform(-np.pi, np.pi, 3)
        x[13:22] = rng.normal(0, 20.0, 9)
        x[22:26] = Rotation.random(random_state=int(rng.integers(1 << 30))).as_quat()
        x[6:10] *= -1.0 if x[9] < 0 else 1.0
        x[22:26] *= -1.0 if x[25] < 0 else 1.0
        a = io_ref.filter_log_row(0.75, x)
        b = np.array(euler_measurement_row(0.75, State.from_vector(x)), dtype=float)
        d = np.abs(a - b)
        ang = [7, 8, 9, 23, 24, 25]
        d[ang] = np.minimum(d[ang], np.abs(d[ang] - 360.0))
        assert (d <= 1e-9 * np.maximum(1.0, np.abs(a))).all()
