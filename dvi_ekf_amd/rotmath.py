"""Host-side rotation helpers (NumPy, no scipy).

These are used only for one-time input preparation and for presenting device
results through the reference-shaped Python API; the per-step arithmetic runs in
the HIP kernels (csrc/eskf_kernels.hip).

Reference semantics followed: tools/Quaternion.py (xyzw storage, normalise with
w >= 0, scipy-1.10.1 ``from_matrix`` = Markley without orthogonalisation) and
tools/math.py:29-39 (skew).
"""
from __future__ import annotations

import math

import numpy as np


def skew(x) -> np.ndarray:
    x = np.asarray(x, dtype=float).reshape(3)
    return np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])


def quat_normalise(q) -> np.ndarray:
    """Unit quaternion with non-negative scalar part (Quaternion.py:195-206)."""
    q = np.asarray(q, dtype=float).reshape(4)
    q = q / math.sqrt(float(q @ q))
    return -q if q[3] < 0 else q


def quat_multiply(a, b) -> np.ndarray:
    """Hamilton product a*b, xyzw, not normalised."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array(
        [
            aw * bx + bw * ax + (ay * bz - az * by),
            aw * by + bw * ay + (az * bx - ax * bz),
            aw * bz + bw * az + (ax * by - ay * bx),
            aw * bw - (ax * bx + ay * by + az * bz),
        ]
    )


def quat_to_matrix(q) -> np.ndarray:
    """Rotation matrix of q / |q| (Quaternion.rot, Quaternion.py:101-107)."""
    q = np.asarray(q, dtype=float).reshape(4)
    x, y, z, w = q / math.sqrt(float(q @ q))
    return np.array(
        [
            [x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w],
        ]
    )


def matrix_to_quat(M) -> np.ndarray:
    """Markley's matrix->quaternion on M as given (no SVD), unit length; this is
    what scipy 1.10.1 (the reference's pin) does in Rotation.from_matrix."""
    M = np.asarray(M, dtype=float).reshape(3, 3)
    tr = M[0, 0] + M[1, 1] + M[2, 2]
    cand = (M[0, 0], M[1, 1], M[2, 2], tr)
    c = max(range(4), key=lambda n: (cand[n], -n))  # first maximum, like argmax
    q = np.empty(4)
    if c == 3:
        q[:] = (M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1], 1.0 + tr)
    else:
        i, j, k = c, (c + 1) % 3, (c + 2) % 3
        q[i] = 1.0 - tr + 2.0 * M[i, i]
        q[j] = M[j, i] + M[i, j]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
    return q / math.sqrt(float(q @ q))


def euler_xyz_extrinsic(q) -> np.ndarray:
    """Angles (a,b,c) with R = Rz(c) Ry(b) Rx(a)  == scipy as_euler('xyz')."""
    R = quat_to_matrix(q)
    return np.array(
        [math.atan2(R[2, 1], R[2, 2]), -math.asin(max(-1.0, min(1.0, R[2, 0]))), math.atan2(R[1, 0], R[0, 0])]
    )


def euler_xyz_intrinsic(q) -> np.ndarray:
    """Angles (a,b,c) with R = Rx(a) Ry(b) Rz(c)  == scipy as_euler('zyx')[::-1]."""
    R = quat_to_matrix(q)
    return np.array(
        [math.atan2(-R[1, 2], R[2, 2]), math.asin(max(-1.0, min(1.0, R[0, 2]))), math.atan2(-R[0, 1], R[0, 0])]
    )


def quat_about_z(angle: float) -> np.ndarray:
    """Quaternion(val=[0,0,angle], euler='xyz') (Quaternion.py:73)."""
    return np.array([0.0, 0.0, math.sin(0.5 * angle), math.cos(0.5 * angle)])
