"""NumPy twin of the ESKF oracle -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement of the reference's per-step predict/update
(salehahr/dvi-ekf @ v1).  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product (``dvi_ekf_amd``) never does.

It is deliberately literal: the probe kinematics are the reference's standard-DH
chain evaluated numerically (Probe.py:14,135-167) and the error-state Jacobians
are obtained by differentiating the reference's own expressions
(symbols.py:129-200) with the complex-step method, mirroring what the reference
does with ``casadi.jacobian`` (Filter.py:270-342).  The hand-derived closed
forms that the HIP kernels use live in ``oracle/closed_form.py`` and are checked
against this file.

Pinning: ``tests/test_oracle_golden.py`` checks this twin against the reference's
saved trajectory ``tests/golden/kf_best_mandala0_mono.txt`` (all 300 values to
the 9-decimal print rounding).

Quirk flags: every place where the v1 source does something other than what
its own comments / the textbook ESKF say is behind a named flag of ``RefParams.quirks``.  The default
-- all of them on -- is the v1 source, which is what the saved trajectory pins;
``tests/test_oracle_quirks.py`` switches them off one at a time and shows that the saved trajectory
resolves every one of them (the restatement is pinned quirk by quirk, not only as a whole).  The HIP
kernels and the C restatement implement the default set only.

Conventions: quaternions are stored xyzw (Quaternion.py:113-115); all maths fp64.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

NX = 26  # nominal state   (state.py:13)
NE = 24  # error state     (state.py:91)
NN = 13  # process noise   (Filter.py:52)
NM = 7   # measurement     (Filter.py:51)
SEL = (18, 19, 20, 21, 22, 23, 15)  # rows picked by H (Filter.py:83-85)

# v1 behaviours that a "corrected" filter would not have; name -> where it comes from
QUIRKS = {
    "Q1_jac_col_shift": "symbols.py:103,107; Filter.py:279-285 -- camera-row Jacobian blocks land one state block early",
    "Q2_vtr_is_ptr": "Probe.py:458 -- the 'true' v_CB is the true p_CB",
    "Q3_dqc_axis_theta": "state.py:124 -- the camera error quaternion turns about the IMU error axis",
    "Q4_q_imu_zero": "Filter.py:40,68-72,110-117 -- IMU block of Q built with dt = 0 in run mode",
    "Q5_jac_at_sigma": "Filter.py:330-333 -- Jacobians evaluated at noise = sigma, not 0",
    "Q6_half_angle_residual": "Quaternion.py:138-147 -- rotation residual asin|v| (half the angle)",
    "Q7_stale_Rwb_old": "Filter.py:227,254,327 -- Fx after an update uses the pre-update attitude",
    "Q8_jac_uses_new_dofs": "Filter.py:220-221,325-326 -- Fx uses post-predict dofs / notch with pre-predict omega, a, R",
    "Q9_Rc_uses_om_old": "equations.py:50,86 -- camera attitude integrated with omega_old, not the mean rate",
    "Q10_markley_first_order": "equations.py:46; state.py:67; scipy 1.10.1 -- Markley on the non-orthogonal R + R[.]x",
}
V1 = frozenset(QUIRKS)


def ref_process_noise(dt, sigma_a, sigma_om, rw_var, quirks=V1):
    """Q of Filter.__init__ / update_noise_matrices (Filter.py:68-72,110-117).  The run-mode filter builds it
    once with self._dt = 0.0 (Filter.py:40): the IMU block is zero (Q4); without the quirk it is what
    update_noise_matrices would give with the step in hand."""
    d = 0.0 if "Q4_q_imu_zero" in quirks else dt
    Q = np.eye(NN)
    Q[0:3, 0:3] = d ** 2 * np.asarray(sigma_a, dtype=float) ** 2 * np.eye(3)
    Q[3:6, 3:6] = d ** 2 * np.asarray(sigma_om, dtype=float) ** 2 * np.eye(3)
    Q[6:13, 6:13] = np.diag(rw_var)
    return Q


# --------------------------------------------------------------------------- #
# small rotation helpers (tools/math.py:29-39, tools/Quaternion.py)
# --------------------------------------------------------------------------- #
def skew(x):
    """[x]x, tools/math.py:29-39 (also casadi.skew)."""
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], dtype=np.result_type(x[0], float))


def quat_mul(a, b):
    """Hamilton product, xyzw (Quaternion.py:170-173)."""
    av, aw = a[:3], a[3]
    bv, bw = b[:3], b[3]
    w = aw * bw - av @ bv
    v = aw * bv + bw * av + np.cross(av, bv)
    return np.array([v[0], v[1], v[2], w])


def quat_nrm(q):
    """Quaternion.normalise (Quaternion.py:195-206): divide by norm, force w>=0."""
    d = math.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2 + q[3] ** 2)
    q = np.array([q[0] / d, q[1] / d, q[2] / d, q[3] / d])
    if q[3] < 0:
        q = -q
    return q


def quat_to_mat(q):
    """Quaternion.rot (Quaternion.py:101-107) = scipy from_quat (normalises) .as_matrix()."""
    n = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    x, y, z, w = q[0] / n, q[1] / n, q[2] / n, q[3] / n
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    return np.array(
        [
            [x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)],
            [2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)],
            [2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2],
        ]
    )


def mat_to_quat_markley(M):
    """scipy 1.10.1 Rotation.from_matrix(M).as_quat() -- Markley's method applied
    directly to M (no prior orthogonalisation), then normalised.  Used by
    Quaternion(val=3x3) (Quaternion.py:69)."""
    d = (M[0, 0], M[1, 1], M[2, 2], M[0, 0] + M[1, 1] + M[2, 2])
    c = int(np.argmax(d))
    q = np.empty(4)
    if c != 3:
        i = c
        j = (i + 1) % 3
        k = (j + 1) % 3
        q[i] = 1 - d[3] + 2 * M[i, i]
        q[j] = M[j, i] + M[i, j]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
    else:
        q[0] = M[2, 1] - M[1, 2]
        q[1] = M[0, 2] - M[2, 0]
        q[2] = M[1, 0] - M[0, 1]
        q[3] = 1 + d[3]
    return q / math.sqrt(q @ q)


def quat_from_mat(M, orthogonalise=False):
    """Quaternion(val=M, do_normalise=True) (Quaternion.py:45-46,69).  orthogonalise: what newer scipy does with
    a matrix that is not a rotation (nearest rotation by SVD first) -- the alternative to quirk Q10."""
    if orthogonalise:
        U, _, Vt = np.linalg.svd(M)
        if np.linalg.det(U @ Vt) < 0:
            U[:, -1] = -U[:, -1]
        M = U @ Vt
    return quat_nrm(mat_to_quat_markley(M))


def quat_about_axis(angle, axis):
    """Quaternion.about_axis (Quaternion.py:209-224)."""
    q = np.array([0.0, axis[0], axis[1], axis[2]])  # wxyz scratch as in the reference
    qlen = math.sqrt(q @ q)
    eps4 = np.finfo(float).eps * 4.0
    if qlen > eps4:
        q *= math.sin(angle / 2.0) / qlen
    q[0] = math.cos(angle / 2.0)
    return quat_nrm(np.array([q[1], q[2], q[3], q[0]]))


def quat_angle_axis_residual(q, half_angle=True):
    """err_q.angle * err_q.axis (Filter.py:371; Quaternion.py:138-167).
    angle = asin(|v|) -- half the rotation angle (Q6) --; axis = v/|v| unless asin(|v|) is exactly 0."""
    nv = math.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2)
    ang = math.asin(nv)
    if math.isclose(ang, 0):  # default rel_tol 1e-9, abs_tol 0 -> true only for exact 0
        return np.zeros(3)
    return (ang if half_angle else 2.0 * ang) * (q[:3] / nv)


# --------------------------------------------------------------------------- #
# probe kinematics: standard-DH chain (Probe.py:14,135-167,194-259,382-410,431-480)
# --------------------------------------------------------------------------- #
def _dh(theta, d, a, alpha):
    """Standard DH link transform Rz(theta) Tz(d) Tx(a) Rx(alpha); complex-safe."""
    ct, st = np.cos(theta), np.sin(theta)
    ca, sa = np.cos(alpha), np.sin(alpha)
    return np.array(
        [
            [ct, -st * ca, st * sa, a * ct],
            [st, ct * ca, -ct * sa, a * st],
            [0, sa, ca, d],
            [0, 0, 0, 1],
        ],
        dtype=np.result_type(theta, d, float),
    )


def probe_chain(q, L, theta_cam):
    """Forward kinematics of the 8-link probe (Probe.py:147-167) with base
    Ry(-180 deg) (Probe.py:14,140).  q = (q1..q7); q8 = 0 (Probe.py:441).
    Returns p_CB (3), R_BC (3x3) and z6 = axis of joint 7 in B (the angular
    Jacobian column of joint 7, Probe.py:204-216)."""
    hp = math.pi / 2
    dtype = np.result_type(*q, float)
    T = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=dtype)  # Ry(-pi)
    links = (
        (q[0] + hp, 0.0, hp),      # Rev  alpha= pi/2 offset= pi/2
        (q[1] - hp, 0.0, -hp),     # Rev  alpha=-pi/2 offset=-pi/2
        (q[2], 0.0, 0.0),          # Rev  alpha=0
        (0.0, q[3], hp),           # Pri  theta=0     alpha=pi/2
        (hp, q[4], hp),            # Pri  theta=pi/2  alpha=pi/2
        (-hp, q[5], hp),           # Pri  theta=-pi/2 alpha=pi/2
    )
    for th, d, al in links:
        T = T @ _dh(th, d, 0.0, al)
    z6 = T[:3, 2].copy()
    T = T @ _dh(q[6], L, 0.0, theta_cam)   # Rev d=L alpha=theta_cam
    T = T @ _dh(0.0, 0.0, 0.0, 0.0)        # Rev d=0 (q8 = 0)
    return T[:3, 3], T[:3, :3], z6


def probe_fwkin(dofs, notch, L, theta_cam):
    """SymProbe.get_est_fwkin (Probe.py:470-480): (p, R, v, om, acc, alp) at
    q=(dofs, notch[0]), qd7=notch[1], qdd7=notch[2], every other rate 0
    (Probe.py:401-410).  v = J_lin[:,6]*qd7 = z6 x (p - o6) qd7 = 0 and
    acc = 0 because p - o6 = L*z6 (link 7 has a=0, link 8 is null)."""
    q = np.concatenate([np.asarray(dofs), [notch[0]]])
    p, R, z6 = probe_chain(q, L, theta_cam)
    z = np.zeros(3, dtype=p.dtype)
    return p, R, z, z6 * notch[1], z, z6 * notch[2]


# --------------------------------------------------------------------------- #
# the filter
# --------------------------------------------------------------------------- #
@dataclass
class RefParams:
    """Constants the filter reads from config / Simulator (Filter.py:34-93)."""
    L: float = 50.0                      # config.yaml:33
    theta_cam: float = math.radians(30)  # config.yaml:34, config.py:131-133
    frozen: tuple = (1, 1, 1, 1, 1, 1)   # config.yaml:8
    hold_frozen: bool = False            # G1: False = v1 literal (zero), True = golden-compat (hold)
    sigma_a: np.ndarray = field(default_factory=lambda: np.full(3, 400e-6 * 981 * math.sqrt(10)))
    sigma_om: np.ndarray = field(default_factory=lambda: np.full(3, math.radians(0.005 * math.sqrt(10))))
    Q: np.ndarray = field(default_factory=lambda: np.eye(NN))   # 13x13 (Filter.py:68-72; ref_process_noise)
    R: np.ndarray = field(default_factory=lambda: np.eye(NM))   # 7x7   (Filter.py:75)
    quirks: frozenset = V1                # SURVEY.md App. A.6; the default is the v1 source


class RefFilter:
    """One filter instance; mirrors Filter.propagate / Filter.update (Filter.py:219-395)."""

    def __init__(self, prm: RefParams, x0, P0, om0, acc0):
        self.prm = prm
        self.x = np.array(x0, dtype=float)             # p v q dofs notch p_c q_c
        self.P = np.array(P0, dtype=float)
        self.om_old = np.array(om0, dtype=float)
        self.acc_old = np.array(acc0, dtype=float)
        self.R_WB_old = quat_to_mat(self.x[6:10])      # Filter.py:80
        self.Fx = None
        self.Fi = None
        self.K = None
        self.res = None
        H = np.zeros((NM, NE))
        H[0:6, 18:24] = np.eye(6)
        H[6, 15] = 1
        self.H = H

    # -- views ---------------------------------------------------------------
    @property
    def dofs(self):
        return self.x[10:16]

    @property
    def notch(self):
        return self.x[16:19]

    # -- predict ---------------------------------------------------------------
    def predict(self, dt, om, acc):
        """Filter.propagate (Filter.py:219-230)."""
        om = np.asarray(om, dtype=float)
        acc = np.asarray(acc, dtype=float)
        self._pre_dofs, self._pre_notch = self.x[10:16].copy(), self.x[16:19].copy()
        self._predict_nominal(dt, om, acc)
        self._predict_error(dt)
        self.P = self.Fx @ self.P @ self.Fx.T + self.Fi @ self.prm.Q @ self.Fi.T  # Filter.py:349
        self.om_old = om
        self.acc_old = acc
        self.R_WB_old = quat_to_mat(self.x[6:10])       # Filter.py:227

    def _predict_nominal(self, dt, om_n, acc_n):
        """Filter._predict_nominal (232-247) + eqns.f_predict (equations.py:44-50,72-100)
        + State.from_array (state.py:62-74)."""
        prm = self.prm
        x = self.x
        p, v = x[0:3], x[3:6]
        R = quat_to_mat(x[6:10])          # State.array -> q.rot (state.py:81)
        dofs, notch = x[10:16].copy(), x[16:19]
        p_c = x[19:22]
        R_c = quat_to_mat(x[22:26])
        B_p, R_BC, B_v, B_om, _, _ = probe_fwkin(dofs, notch, prm.L, prm.theta_cam)
        om_o, acc_o = self.om_old, self.acc_old

        om_avg = (om_o + om_n) / 2                                   # equations.py:44
        R_next = R + R @ skew(dt * om_avg)                           # equations.py:46
        acc_avg = (R @ acc_o + R_next @ acc_n) / 2                   # equations.py:47
        C_om = R_BC.T @ ((om_o if "Q9_Rc_uses_om_old" in prm.quirks else om_avg) + B_om)   # equations.py:50
        p_next = p + dt * v + (dt ** 2 / 2) * acc_avg                # equations.py:76
        v_next = v + dt * acc_avg                                    # equations.py:77
        notch_next = np.array([notch[0] + dt * notch[1], notch[1] + dt * notch[2], notch[2]])
        pc_next = p_c + dt * v + dt * R @ (B_v + np.cross(om_avg, B_p))   # equations.py:83-85
        Rc_next = R_c + R_c @ skew(dt * C_om)                        # equations.py:86

        if not prm.hold_frozen:                                      # Filter.py:243-245
            for i, fr in enumerate(prm.frozen):
                if fr:
                    dofs[i] = 0.0
        svd = "Q10_markley_first_order" not in prm.quirks
        q_next = quat_from_mat(R_next, svd)                          # state.py:67
        qc_next = quat_from_mat(Rc_next, svd)                        # state.py:72
        self.x = np.concatenate([p_next, v_next, q_next, dofs, notch_next, pc_next, qc_next])

    # reference expressions whose derivatives fill the camera rows -------------
    def _err_cam_next(self, dt, ev):
        """err_p_C_next, err_theta_C_next (Filter.py:274-277) as a function of the
        35 variables ev = [err_p_B 3, err_v_B 3, err_theta 3, err_q(1..7) 7,
        err_p_C 3, err_theta_C 3 | n_a 3, n_om 3, n_dofs 6, n_notch_acc 1].
        Complex-safe.  symbols.py:129-200; Probe.py:456-468."""
        prm = self.prm
        err_v_B = ev[3:6]
        err_theta = ev[6:9]
        err_q = ev[9:16]
        err_p_C = ev[16:19]
        err_theta_C = ev[19:22]
        n_om = ev[25:28]
        if "Q8_jac_uses_new_dofs" in prm.quirks:
            dofs, notch = self.x[10:16], self.x[16:19]     # post-predict values (Filter.py:325-326)
        else:
            dofs, notch = self._pre_dofs, self._pre_notch
        R_WB = self.R_WB_old                             # Filter.py:327
        om_b = self.om_old                               # Filter.py:328

        q_est = np.concatenate([dofs, [notch[0]]])
        p, R, _ = probe_chain(q_est, prm.L, prm.theta_cam)
        _, _, z6 = probe_chain(q_est, prm.L, prm.theta_cam)
        v = np.zeros(3)
        om = z6 * notch[1]
        p_tr, R_tr, z6_tr = probe_chain(q_est + err_q, prm.L, prm.theta_cam)   # Probe.py:465-468
        v_tr = p_tr if "Q2_vtr_is_ptr" in prm.quirks else np.zeros(3)   # Probe.py:458 (sic); v_CB = 0 at any q
        om_probe_tr = z6_tr * notch[1]                   # Probe.py:459 (only q substituted)

        R_WB_tr = R_WB @ (np.eye(3) + skew(err_theta))   # symbols.py:129
        om_tr = om_b - n_om                              # symbols.py:131

        # get_err_pc_dot (symbols.py:134-160)
        p_CB_dot = R_WB @ (v + np.cross(om_b, p))
        p_CB_dot_tr = R_WB_tr @ (v_tr + np.cross(om_tr, p_tr))
        err_p_C_dot = err_v_B + p_CB_dot_tr - p_CB_dot

        # get_err_theta_c_dot (symbols.py:186-200), quat_matrix (168-183) with w=0
        om_c = R.T @ (om_b + om)
        om_c_tr = R_tr.T @ (om_tr + om_probe_tr)
        M_low_col0 = om_c_tr - om_c                      # (matr_r - matr_l)[1:,0]
        M_low = -skew(om_c_tr) - skew(om_c)              # (w I - [v]x) - (w I + [v]x), w=0
        err_theta_c_dot = M_low_col0 + M_low @ (0.5 * err_theta_C)

        return np.concatenate([err_p_C + dt * err_p_C_dot, err_theta_C + dt * err_theta_c_dot])

    def _predict_error(self, dt):
        """Filter._predict_error (249-268) + _cam_error_jacobian (270-342)."""
        prm = self.prm
        Fx = np.eye(NE)
        Fx[0:3, 3:6] = dt * np.eye(3)
        Fx[3:6, 6:9] = -self.R_WB_old @ skew(self.acc_old) * dt
        Om = quat_nrm(np.array([*(0.5 * dt * self.om_old), 1.0]))   # Filter.py:132-134
        Fx[6:9, 6:9] = quat_to_mat(Om).T
        Fx[15:17, 16:18] += dt * np.eye(2)

        # evaluation point: error states 0, noise symbols = their std-devs (Filter.py:330-340)
        ev0 = np.zeros(35)
        if "Q5_jac_at_sigma" in prm.quirks:
            ev0[22:25] = prm.sigma_a
            ev0[25:28] = prm.sigma_om
        # n_dofs / n_notch_acc do not appear in the expressions; any value works
        h = 1e-30
        J = np.empty((6, 35))
        for k in range(35):
            ev = ev0.astype(complex)
            ev[k] += 1j * h
            J[:, k] = self._err_cam_next(dt, ev).imag / h

        # column cursor of Filter.py:279-285 over syms.err_x (symbols.py:107):
        # widths 3,3,3,6,1,3,3 = 22 columns -> state Jacobian lands in cols 0:22
        if "Q1_jac_col_shift" in prm.quirks:
            Fx[18:24, 0:22] = J[:, 0:22]
        else:   # the columns where the variables live in the 24-vector: the notch block is 3 wide, not 1
            Fx[18:24, 0:16] = J[:, 0:16]
            Fx[18:24, 16:18] = 0.0
            Fx[18:24, 18:24] = J[:, 16:22]
        # noise: widths 3,3,6,1 = 13 (symbols.py:124)
        Fi = np.zeros((NE, NN))
        Fi[3:15, 0:12] = np.eye(12)
        Fi[17, 12] = 1
        Fi[18:24, 0:13] = J[:, 22:35]
        self.Fx, self.Fi = Fx, Fi

    # -- update ------------------------------------------------------------------
    def update(self, pos, quat_raw, ang_notch):
        """Filter.update (351-395).  Returns K, or None if S is singular."""
        prm = self.prm
        H, P = self.H, self.P
        S = H @ P @ H.T + prm.R
        try:
            K = P @ H.T @ np.linalg.inv(S)
        except np.linalg.LinAlgError:
            return None

        # Quaternion(val=[0,0,ang], euler="xyz") (Quaternion.py:73): rotation about z
        notch_quat = np.array([0.0, 0.0, math.sin(ang_notch / 2), math.cos(ang_notch / 2)])
        q_meas = quat_nrm(quat_mul(notch_quat, np.asarray(quat_raw, dtype=float)))   # Filter.py:365

        res_p = np.asarray(pos, dtype=float) - self.x[19:22]
        qm_conj = np.array([-q_meas[0], -q_meas[1], -q_meas[2], q_meas[3]])
        err_q = quat_nrm(quat_mul(qm_conj, self.x[22:26]))                            # Filter.py:370
        res_q = quat_angle_axis_residual(err_q, "Q6_half_angle_residual" in prm.quirks)
        res_n = ang_notch - self.x[16]
        res = np.hstack((res_p, res_q, res_n))
        self.res = res

        e = K @ res
        for i, fr in enumerate(prm.frozen):                                            # Filter.py:377-379
            if fr:
                e[9 + i] = 0.0

        # ErrorState.from_array (state.py:105-129) and State.__add__ (46-60)
        theta, theta_c = e[6:9], e[21:24]
        dq = quat_about_axis(math.sqrt(theta @ theta), theta)
        dqc = quat_about_axis(math.sqrt(theta_c @ theta_c),
                              theta if "Q3_dqc_axis_theta" in prm.quirks else theta_c)   # sic: axis = theta (state.py:124)
        x = self.x.copy()
        x[0:3] += e[0:3]
        x[3:6] += e[3:6]
        x[6:10] = quat_nrm(quat_nrm(quat_mul(self.x[6:10], dq)))
        x[10:16] += e[9:15]
        x[16:19] += e[15:18]
        x[19:22] += e[18:21]
        x[22:26] = quat_nrm(quat_nrm(quat_mul(self.x[22:26], dqc)))
        self.x = x

        I = np.eye(NE)
        A = I - K @ H
        P = A @ P @ A.T + K @ prm.R @ K.T                                              # Filter.py:384
        G = np.eye(NE)
        G[6:9, 6:9] = np.eye(3) - skew(0.5 * theta)
        G[21:24, 21:24] = np.eye(3) - skew(0.5 * theta_c)
        self.P = G @ P @ G.T                                                           # Filter.py:390
        if "Q7_stale_Rwb_old" not in prm.quirks:
            self.R_WB_old = quat_to_mat(self.x[6:10])     # (v1 leaves the pre-update attitude in place until the next predict)
        self.K = K
        return K
