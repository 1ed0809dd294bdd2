"""Flattens one simulated run into plain arrays: initial condition, the IMU
sample stream (one predict per sample) and the camera measurements (one update
per frame).  This is the hand-over format between the one-time host preparation
and the device filter -- the same arrays drive ``BatchedESKF.run_schedule`` and
the parity tests.

Schedule semantics: Filter.run / run_one_epoch / propagate_imu
(Filter.py:144-217), initial condition: tools/utils.py:54-75 and
Filter.__init__ (Filter.py:63,78-80).  SURVEY.md App. A.1 steps 6-7.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .imu import imu_pose_from_camera
from .quaternion import Quaternion


@dataclass
class RunInputs:
    t0: float
    x0: np.ndarray          # [26]  p v q(xyzw) dofs notch p_cam q_cam(xyzw)
    P0: np.ndarray          # [24,24]
    om0: np.ndarray         # [3]   IMU sample at interpolated index 0
    acc0: np.ndarray        # [3]
    t_imu: np.ndarray       # [n]   time stamp of every predict
    dt: np.ndarray          # [n]
    om: np.ndarray          # [n,3]
    acc: np.ndarray         # [n,3]
    n_pred: np.ndarray      # [m]   predicts before update k
    t_meas: np.ndarray      # [m]
    pos: np.ndarray         # [m,3] scaled camera position
    quat: np.ndarray        # [m,4] RAW file quaternion xyzw (not normalised)
    notch: np.ndarray       # [m]
    imu_ref: np.ndarray     # [n,7] reference IMU pose per predict: p(3) R->quat xyzw(4)
    imu_ref_v: np.ndarray   # [n,3]


def initial_state(camera, imu, dofs_ic) -> np.ndarray:
    """get_ic (utils.py:54-75): IMU pose from the first camera frame through the
    real probe; the YAML ``filter.ic.x0`` block is not used by the reference."""
    cam_reference = camera.rotated if camera.rotated else camera       # utils.py:63
    notch0 = cam_reference.get_notch_vec_at(0)
    p_B, R_WB, v_B = imu.ref_vals(cam_reference.vec0, notch0)
    q = Quaternion(val=R_WB, do_normalise=True)
    return np.concatenate([p_B, v_B, q.xyzw, np.asarray(dofs_ic, dtype=float), notch0,
                           cam_reference.p0.reshape(3), cam_reference.q0.xyzw])


def build_run_inputs(config, camera, imu, probe) -> RunInputs:
    gt = [np.array(v, dtype=float) for v in probe.joint_dofs]
    x0 = initial_state(camera, imu, config.ic_imu_dofs)

    # IMU sample at index 0 of the interpolated camera (Filter.py:63; Imu.py:198-226)
    cam_i = imu.cam
    om0, acc0 = imu.eval_expr_single(cam_i.t[0], gt[0], gt[1], gt[2], cam_i.acc[:, 0], cam_i.R[0],
                                     cam_i.om[:, 0], cam_i.alp[:, 0])

    t_imu, dts, oms, accs, n_pred, refs, refv = [], [], [], [], [], [], []
    t_meas, pos, quat, notch = [], [], [], []
    old_t = camera.t[0]
    for k in range(1, len(camera.t)):
        t = camera.t[k]
        samples = cam_i.samples_between(old_t, t)
        old_ti = old_t
        for s in samples:
            gt[0][6], gt[1][6], gt[2][6] = s.notch, s.notch_d, s.notch_dd
            om, acc = imu.eval_expr_single(s.t, gt[0], gt[1], gt[2], s.acc, s.R, s.om, s.alp)
            p_ref, R_ref, v_ref = imu_pose_from_camera(s.vec, probe.fwkin(s.notch_arr))
            refs.append(np.concatenate([p_ref, Quaternion(val=R_ref, do_normalise=True).xyzw]))
            refv.append(v_ref)
            t_imu.append(s.t)
            dts.append(s.t - old_ti)
            oms.append(om)
            accs.append(acc)
            old_ti = s.t
        n_pred.append(len(samples))
        m = camera.traj.at_index(k)
        t_meas.append(t)
        pos.append(m.pos)
        quat.append(m.q.xyzw)
        notch.append(camera.get_notch_vec_at(k)[0])
        old_t = t

    return RunInputs(
        t0=float(camera.t[0]), x0=x0, P0=np.array(config.cov0_matrix, dtype=float),
        om0=np.asarray(om0), acc0=np.asarray(acc0),
        t_imu=np.array(t_imu), dt=np.array(dts), om=np.array(oms), acc=np.array(accs),
        n_pred=np.array(n_pred, dtype=np.int64), t_meas=np.array(t_meas), pos=np.array(pos),
        quat=np.array(quat), notch=np.array(notch, dtype=float),
        imu_ref=np.array(refs), imu_ref_v=np.array(refv),
    )


def noise_matrices(config, stdev_na, stdev_nom, dt: float = 0.0):
    """Q (13x13) and R (7x7) as Filter.__init__ builds them (Filter.py:67-75):
    ``dt`` is 0.0 at construction, so the IMU block of Q is zero in run mode."""
    Q = np.eye(13)
    Q[0:3, 0:3] = dt ** 2 * np.asarray(stdev_na) ** 2 * np.eye(3)
    Q[3:6, 3:6] = dt ** 2 * np.asarray(stdev_nom) ** 2 * np.eye(3)
    Q[6:13, 6:13] = np.diag(config.process_noise_rw_var)
    R = np.diag(config.meas_noise_var)
    return Q, R
