"""Input synthesis and trajectory-log rows -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product
(dvi_ekf_amd/) must not.  It is the checker of SURVEY 8(f1) / 8(f2) -- the device kernels
`eskf_imu_kernel` / `eskf_log_rows_kernel` and the product's host code `dvi_ekf_amd/imu.py`, `traj.py` --
and is written independently of both: nothing here uses the closed forms of SURVEY App. B.

What is restated, literally:
  * the probe as a numerical standard-DH chain with roboticstoolbox's geometric Jacobian and the
    reference's own Hessian contraction for the accelerations (models/Probe.py:135-167 links and base,
    204-216 v, om = J qd; 219-259 acc, alp = H qd qd + J qdd) -- for ARBITRARY joint rates, of which
    SimpleProbe only ever sets those of joint 7 (Probe.py:382-410);
  * f_imu and f_imu_meas (kinematics/equations.py:8-41,54-69);
  * the 14-column IMU reference row (models/trajectory/ImuRefTraj.py:40-55) and the 30-column filter log row
    (models/trajectory/FilterTraj.py:12-41), with scipy's Rotation for the Euler angles, as the reference does.
Pinned by tests/test_oracle_io_ref.py to the reference's saved imu_ref_mandala0_mono.txt / kf_best_mandala0_mono.txt.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.spatial.transform import Rotation


def _dh_link(theta, d, a, alpha):
    """A_i = Rz(theta) Tz(d) Tx(a) Rx(alpha) (standard DH, roboticstoolbox DHLink.A)"""
    ct, st, ca, sa = math.cos(theta), math.sin(theta), math.cos(alpha), math.sin(alpha)
    return np.array([[ct, -st * ca, st * sa, a * ct], [st, ct * ca, -ct * sa, a * st], [0.0, sa, ca, d], [0.0, 0.0, 0.0, 1.0]])


# (is_revolute, fixed theta or d, alpha, offset) of the eight links, Probe.py:147-167
def _links(L, theta_cam):
    hp = math.pi / 2
    return [(True, 0.0, hp, hp), (True, 0.0, -hp, -hp), (True, 0.0, 0.0, 0.0),
            (False, 0.0, hp, 0.0), (False, hp, hp, 0.0), (False, -hp, hp, 0.0),
            (True, L, theta_cam, 0.0), (True, 0.0, 0.0, 0.0)]


def probe_kinematics(q8, qd8, qdd8, L, theta_cam):
    """(p, R, v, om, acc, alp) of frame C in frame B for joint values / rates / accelerations q8, qd8, qdd8
    (eight joints): fkine, jacob0 and the Hessian contraction of Probe.py:204-259."""
    q8, qd8, qdd8 = (np.asarray(v, dtype=float).reshape(8) for v in (q8, qd8, qdd8))
    T = np.diag([-1.0, 1.0, -1.0, 1.0])                     # base = SE3.Ry(-180 deg), Probe.py:14,140
    frames = [T.copy()]                                      # frame i-1 carries the axis of joint i
    for (rev, fixed, alpha, offset), qi in zip(_links(L, theta_cam), q8):
        A = _dh_link(qi + offset, fixed, 0.0, alpha) if rev else _dh_link(fixed, qi, 0.0, alpha)
        T = T @ A
        frames.append(T.copy())
    p_end, R_end = T[:3, 3], T[:3, :3]
    J = np.zeros((6, 8))                                     # jacob0: world-frame geometric Jacobian
    for i, (rev, *_rest) in enumerate(_links(L, theta_cam)):
        z, o = frames[i][:3, 2], frames[i][:3, 3]
        if rev:
            J[:3, i], J[3:, i] = np.cross(z, p_end - o), z
        else:
            J[:3, i] = z
    H = np.zeros((6, 8, 8))                                  # Probe.hessian_symbolic, Probe.py:219-232
    for j in range(8):
        for i in range(j, 8):
            H[:3, i, j] = np.cross(J[3:, j], J[:3, i])
            H[3:, i, j] = np.cross(J[3:, j], J[3:, i])
            if i != j:
                H[:3, j, i] = H[:3, i, j]
    vel = J @ qd8                                            # Probe.py:204-206
    acc = np.einsum("aij,j,i->a", H, qd8, qd8) + J @ qdd8   # Probe.py:235-253
    return p_end, R_end, vel[:3], vel[3:], acc[:3], acc[3:]


def simple_probe_fwkin(dofs6, notch3, L, theta_cam):
    """SimpleProbe / SymProbe.get_est_fwkin (Probe.py:382-410,470-480): joints 1..6 = dofs at rest, joint 7 = the
    notch with its rate and acceleration, joint 8 = 0."""
    q = np.concatenate([np.asarray(dofs6, dtype=float), [notch3[0], 0.0]])
    qd = np.array([0, 0, 0, 0, 0, 0, notch3[1], 0.0])
    qdd = np.array([0, 0, 0, 0, 0, 0, notch3[2], 0.0])
    return probe_kinematics(q, qd, qdd, L, theta_cam)


def f_imu(W_p_CW, R_WC, WW_v_CW, W_om_CW, fwkin):
    """equations.py:8-13,54-60: IMU position, attitude and velocity in W from the camera's and the probe's"""
    B_p_CB, R_BC, BB_v_CB, B_om_CB, _, _ = fwkin
    R_WB = R_WC @ R_BC.T                                                        # equations.py:8
    W_p_BW = W_p_CW - R_WB @ B_p_CB                                             # :9
    W_om_BW = W_om_CW - R_WB @ B_om_CB                                          # :11
    W_omB_x_pCB = np.cross(W_om_BW, R_WB @ B_p_CB)                              # :12
    WW_v_BW = WW_v_CW - R_WB @ BB_v_CB - W_omB_x_pCB                            # :13
    return W_p_BW, R_WB, WW_v_BW


def f_imu_meas(R_WC, W_om_CW, W_acc_CW, W_alp_CW, fwkin):
    """equations.py:8-41,63-69: body-frame angular rate and acceleration (no gravity, no bias, no noise)"""
    B_p_CB, R_BC, BB_v_CB, B_om_CB, B_acc_CB, B_alp_CB = fwkin
    R_WB = R_WC @ R_BC.T
    W_om_BW = W_om_CW - R_WB @ B_om_CB
    W_omB_x_pCB = np.cross(W_om_BW, R_WB @ B_p_CB)
    W_alp_BW = W_alp_CW - R_WB @ B_alp_CB - np.cross(W_om_BW, R_WB @ B_om_CB)   # :15-19
    W_acc_BW = (W_acc_CW - R_WB @ B_acc_CB - 2 * np.cross(W_om_BW, R_WB @ BB_v_CB)
                - np.cross(W_alp_BW, R_WB @ B_p_CB) - np.cross(W_om_BW, W_omB_x_pCB))   # :20-26
    R_BW = R_BC @ R_WC.T                                                        # :29
    return R_BW @ W_om_BW, R_BW @ W_acc_BW                                      # :40-41


def _markley_xyzw(M):
    """Quaternion(val=3x3, do_normalise=True) as scipy 1.10.1 does it (no orthogonalisation), w >= 0"""
    d = np.array([M[0, 0], M[1, 1], M[2, 2], M[0, 0] + M[1, 1] + M[2, 2]])
    c = int(np.argmax(d))
    q = np.empty(4)
    if c == 3:
        q[:] = [M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1], 1 + d[3]]
    else:
        i, j, k = c, (c + 1) % 3, (c + 2) % 3
        q[i], q[j], q[k], q[3] = 1 - d[3] + 2 * M[i, i], M[j, i] + M[i, j], M[k, i] + M[i, k], M[k, j] - M[j, k]
    q /= np.linalg.norm(q)
    return -q if q[3] < 0 else q


def imu_ref_row(t, W_p_BW, R_WB, WW_v_BW):
    """ImuRefTraj.append_value (ImuRefTraj.py:40-55): t | p | v | extrinsic-xyz Euler angles [deg] | qw qx qy qz"""
    eul = Rotation.from_matrix(R_WB).as_euler("xyz", degrees=True)
    q = _markley_xyzw(np.asarray(R_WB, dtype=float))
    return np.array([t, *W_p_BW, *WW_v_BW, *eul, q[3], q[0], q[1], q[2]])


def filter_log_row(t, x26):
    """_get_euler_measurement_array (FilterTraj.py:12-34): t | p | v | euler xyz [deg] | q wxyz | dofs (rot in deg) |
    p_cam | euler xyz [deg] of q_cam | q_cam wxyz, from the 26-vector p v q(xyzw) dofs notch p_cam q_cam(xyzw)"""
    x = np.asarray(x26, dtype=float)
    q, qc = x[6:10], x[22:26]
    eul = Rotation.from_quat(q).as_euler("xyz", degrees=True)          # Quaternion.euler_xyz_deg, Quaternion.py:122-123
    eulc = Rotation.from_quat(qc).as_euler("xyz", degrees=True)
    return np.array([t, *x[0:3], *x[3:6], *eul, q[3], q[0], q[1], q[2], *np.rad2deg(x[10:13]), *x[13:16],
                     *x[19:22], *eulc, qc[3], qc[0], qc[1], qc[2]])
