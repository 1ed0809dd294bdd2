"""Closed-form probe kinematics and camera-row Jacobian blocks -- TEST
INFRASTRUCTURE ONLY (see oracle/eskf_ref.py for the rules).

This is the executable specification of what csrc/eskf_kernels.hip computes for
``Fx`` / ``Fi``: hand-derived derivatives instead of
the complex-step differentiation used by the literal twin.  ``tests/
test_oracle_closed_form.py`` checks the two against each other at random points.

Reference lines restated: Filter.py:249-342, symbols.py:129-200,
Probe.py:147-167,204-216,456-468.
"""
from __future__ import annotations

import math

import numpy as np


def _skew(x):
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], dtype=float)


def probe_closed_form(dofs, notch_angle, L, theta_cam):
    """p_CB, R_BC, z6 and their derivatives w.r.t. (q1..q7).

    Returns dict with p (3), R (3x3), z6 (3), Jp (3x7), dR (7 x 3x3), dz6 (7 x 3)."""
    q1, q2, q3, q4, q5, q6 = (float(v) for v in dofs)
    q7 = float(notch_angle)
    s1, c1 = math.sin(q1), math.cos(q1)
    s2, c2 = math.sin(q2), math.cos(q2)
    s3, c3 = math.sin(q3), math.cos(q3)
    sp, cp = math.sin(q3 - q7), math.cos(q3 - q7)
    st, ct = math.sin(theta_cam), math.cos(theta_cam)

    z6 = np.array([-s1 * c2, -c1 * c2, s2])
    z6_1 = np.array([-c1 * c2, s1 * c2, 0.0])
    z6_2 = np.array([s1 * s2, c1 * s2, c2])

    def e5(s, c):      # s, c = sin/cos of the third angle
        return np.array([s1 * s2 * s - c1 * c, s1 * c + c1 * s2 * s, c2 * s])

    def e5_1(s, c):
        return np.array([c1 * s2 * s + s1 * c, c1 * c - s1 * s2 * s, 0.0])

    def e5_2(s, c):
        return np.array([s1 * c2 * s, c1 * c2 * s, -s2 * s])

    def e6(s, c):
        return np.array([s1 * s2 * c + c1 * s, c1 * s2 * c - s1 * s, c2 * c])

    def e6_1(s, c):
        return np.array([c1 * s2 * c - s1 * s, -s1 * s2 * c - c1 * s, 0.0])

    def e6_2(s, c):
        return np.array([s1 * c2 * c, c1 * c2 * c, -s2 * c])

    E5, E6 = e5(s3, c3), e6(s3, c3)
    p = (L - q4) * z6 + q5 * E5 + q6 * E6
    Jp = np.zeros((3, 7))
    Jp[:, 0] = (L - q4) * z6_1 + q5 * e5_1(s3, c3) + q6 * e6_1(s3, c3)
    Jp[:, 1] = (L - q4) * z6_2 + q5 * e5_2(s3, c3) + q6 * e6_2(s3, c3)
    Jp[:, 2] = q5 * E6 - q6 * E5
    Jp[:, 3] = -z6
    Jp[:, 4] = E5
    Jp[:, 5] = E6

    # R_BC = [-a, ct*g + st*z6, -st*g + ct*z6],  a = e5(phi), g = e6(phi), phi = q3 - q7
    a, g = e5(sp, cp), e6(sp, cp)

    def cols(da, dg, dz):
        return np.column_stack([-da, ct * dg + st * dz, -st * dg + ct * dz])

    R = cols(a, g, z6)
    dR = np.zeros((7, 3, 3))
    dR[0] = cols(e5_1(sp, cp), e6_1(sp, cp), z6_1)
    dR[1] = cols(e5_2(sp, cp), e6_2(sp, cp), z6_2)
    dR[2] = cols(g, -a, np.zeros(3))        # d/dphi
    dR[6] = -dR[2]                          # phi = q3 - q7
    dz6 = np.zeros((7, 3))
    dz6[0], dz6[1] = z6_1, z6_2
    return dict(p=p, R=R, z6=z6, Jp=Jp, dR=dR, dz6=dz6)


def error_jacobians(dt, dofs, notch, R_WB_old, om_old, acc_old, sigma_om, L, theta_cam, Om_old_rot_T):
    """Fx (24x24), Fi (24x13) per SURVEY.md App. A.3 (with quirks Q1, Q2, Q5, Q8)."""
    k = probe_closed_form(dofs, notch[0], L, theta_cam)
    p, R_BC, z6, Jp, dR, dz6 = k["p"], k["R"], k["z6"], k["Jp"], k["dR"], k["dz6"]
    thd = notch[1]
    om_t = np.asarray(om_old) - np.asarray(sigma_om)     # noise symbols at their std-devs (Q5)
    w = p + np.cross(om_t, p)                            # v_tr := p_tr (Q2)
    u_t = om_t + z6 * thd
    u_o = np.asarray(om_old) + z6 * thd
    a_c = R_BC.T @ u_t
    b_c = R_BC.T @ u_o

    Fx = np.eye(24)
    Fx[0:3, 3:6] = dt * np.eye(3)
    Fx[3:6, 6:9] = -dt * R_WB_old @ _skew(acc_old)
    Fx[6:9, 6:9] = Om_old_rot_T
    Fx[15, 16] += dt
    Fx[16, 17] += dt
    Fx[18:24, 0:22] = 0.0
    Fx[18:21, 3:6] = dt * np.eye(3)
    Fx[18:21, 6:9] = -dt * R_WB_old @ _skew(w)
    Fx[18:21, 9:16] = dt * R_WB_old @ (np.eye(3) + _skew(om_t)) @ Jp
    Fx[18:21, 16:19] = np.eye(3)                         # column shift (Q1)
    for i in range(7):
        Fx[21:24, 9 + i] = dt * (dR[i].T @ u_t + R_BC.T @ dz6[i] * thd)
    Fx[21:24, 19:22] = np.eye(3) - 0.5 * dt * _skew(a_c + b_c)

    Fi = np.zeros((24, 13))
    Fi[3:15, 0:12] = np.eye(12)
    Fi[17, 12] = 1.0
    Fi[18:21, 3:6] = dt * R_WB_old @ _skew(p)
    Fi[21:24, 3:6] = -dt * R_BC.T
    return Fx, Fi
