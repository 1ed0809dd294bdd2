"""Closed-form kinematics of the 8-joint DH "probe" that links IMU frame B to
camera frame C (host side, NumPy).

The reference builds this chain with roboticstoolbox + sympy + casadi
(models/Probe.py:14,135-167,194-259,382-410,431-480).  Because all link lengths
``a_i`` are zero, joint 8 is locked at 0 and only joint 7 has a rate, the chain
collapses to the expressions below; the same formulas are
implemented on the device in csrc/eskf_kernels.hip.

Joint vector: q1..q3 revolute (IMU orientation), q4..q6 prismatic (IMU
translation), q7 revolute (notch).  Filter DOFs = (q1..q6), notch = (q7, q7', q7'').
"""
from __future__ import annotations

import math

import numpy as np


def probe_pose(q, length: float, theta_cam: float):
    """Returns (p_CB, R_BC, z6) for joint values q = (q1..q7)."""
    q1, q2, q3, q4, q5, q6, q7 = (float(v) for v in q)
    s1, c1 = math.sin(q1), math.cos(q1)
    s2, c2 = math.sin(q2), math.cos(q2)
    s3, c3 = math.sin(q3), math.cos(q3)
    sp, cp = math.sin(q3 - q7), math.cos(q3 - q7)
    st, ct = math.sin(theta_cam), math.cos(theta_cam)

    z6 = np.array([-s1 * c2, -c1 * c2, s2])
    e5 = np.array([s1 * s2 * s3 - c1 * c3, s1 * c3 + c1 * s2 * s3, c2 * s3])
    e6 = np.array([s1 * s2 * c3 + c1 * s3, c1 * s2 * c3 - s1 * s3, c2 * c3])
    p = (length - q4) * z6 + q5 * e5 + q6 * e6

    # columns of R_BC: x7 = u cos(phi) + w sin(phi) ... written out per entry
    R = np.array(
        [
            [-s1 * s2 * sp + c1 * cp, s1 * s2 * ct * cp - s1 * c2 * st + c1 * ct * sp,
             -s1 * s2 * st * cp - s1 * c2 * ct - c1 * st * sp],
            [-s1 * cp - c1 * s2 * sp, -s1 * ct * sp + c1 * s2 * ct * cp - c1 * c2 * st,
             s1 * st * sp - c1 * s2 * st * cp - c1 * c2 * ct],
            [-c2 * sp, s2 * st + c2 * ct * cp, s2 * ct - c2 * st * cp],
        ]
    )
    return p, R, z6


def probe_fwkin(dofs, notch, length: float, theta_cam: float):
    """(p, R, v, om, acc, alp) of C relative to B, in B -- the six outputs of
    SymProbe.get_est_fwkin (Probe.py:470-480).  v and acc vanish identically."""
    p, R, z6 = probe_pose([*dofs, notch[0]], length, theta_cam)
    zero = np.zeros(3)
    return p, R, zero, z6 * float(notch[1]), zero, z6 * float(notch[2])


class SimpleProbe:
    """Ground-truth sensor setup: every joint constrained except the notch q7
    (Probe.py:382-410)."""

    constraints = (0.0, 0.0, 0.0, 0.0, 0.0, 20.0, None, 0.0)

    def __init__(self, scope_length: float, theta_cam: float):
        self.scope_length = float(scope_length)
        self.theta_cam = float(theta_cam)
        self.n = 8
        self.q = np.array([0.0 if c is None else c for c in self.constraints])
        self.qd = np.zeros(8)
        self.qdd = np.zeros(8)

    @property
    def joint_dofs(self):
        return [self.q.copy(), self.qd.copy(), self.qdd.copy()]

    @property
    def imu_dofs(self):
        return list(self.q[:6])

    def fwkin(self, notch):
        """Real forward kinematics as a function of the notch DOFs (Imu.py:57-64)."""
        return probe_fwkin(self.q[:6], notch, self.scope_length, self.theta_cam)


class SymProbe:
    """Holder of the estimated-kinematics relation (Probe.py:431-480).  No symbolic
    engine here: ``get_est_fwkin`` evaluates the closed form."""

    def __init__(self, probe: SimpleProbe, const_dofs: bool = False):
        self.n = probe.n
        self.q0 = probe.q.copy()
        self.scope_length = probe.scope_length
        self.theta_cam = probe.theta_cam

    def get_est_fwkin(self, dofs_est, notchdofs):
        return list(probe_fwkin(dofs_est, notchdofs, self.scope_length, self.theta_cam))
