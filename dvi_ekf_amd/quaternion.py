"""Host-side ``Quaternion`` with the reference's public surface
(tools/Quaternion.py:12-224) on top of :mod:`dvi_ekf_amd.rotmath` -- no scipy.

Only used for presenting results and for one-time input preparation; the per-step
quaternion algebra of the filter runs on the device.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from . import rotmath as rm


class Quaternion:
    """xyzw quaternion.  Euler angles are extrinsic (fixed-frame) xyz."""

    def __init__(self, val=None, x=None, y=None, z=None, w=None, v=None,
                 do_normalise: bool = False, euler: str = ""):
        if val is not None:
            if isinstance(val, Quaternion):
                xyzw = val.xyzw
            else:
                val = np.asarray(val, dtype=float)
                if val.shape == (3, 3):
                    xyzw = rm.matrix_to_quat(val)
                elif val.shape in ((4, 1), (4,)):
                    xyzw = val.reshape(4)
                elif val.squeeze().shape == (3,) and euler == "xyz":
                    a, b, c = val.squeeze()
                    xyzw = rm.quat_multiply(
                        rm.quat_about_z(c),
                        rm.quat_multiply([0.0, math.sin(b / 2), 0.0, math.cos(b / 2)],
                                         [math.sin(a / 2), 0.0, 0.0, math.cos(a / 2)]))
                else:
                    raise ValueError("Wrong dimensions of np.ndarray for Quaternion!")
            self.x, self.y, self.z, self.w = (float(c) for c in xyzw)
        elif v is not None and w is not None:
            self.x, self.y, self.z = (float(c) for c in np.asarray(v, dtype=float).reshape(3))
            self.w = float(w)
        elif None not in (x, y, z, w):
            self.x, self.y, self.z, self.w = float(x), float(y), float(z), float(w)
        else:
            raise TypeError(f"Wrong input type to Quaternion! Is: {type(val)}")
        if do_normalise:
            self.normalise()

    def __repr__(self):
        return f"Quaternion [x={self.x:.3f}, y={self.y:.3f}, z={self.z:.3f}, w={self.w:.3f}]"

    @property
    def v(self) -> np.ndarray:
        return np.array([self.x, self.y, self.z])

    @property
    def xyzw(self) -> np.ndarray:
        return np.array([self.x, self.y, self.z, self.w])

    @property
    def wxyz(self) -> np.ndarray:
        return np.array([self.w, self.x, self.y, self.z])

    @property
    def rot(self) -> np.ndarray:
        return rm.quat_to_matrix(self.xyzw)

    @property
    def euler_xyz_rad(self) -> np.ndarray:
        return rm.euler_xyz_extrinsic(self.xyzw)

    @property
    def euler_xyz_deg(self) -> np.ndarray:
        return np.degrees(self.euler_xyz_rad)

    @property
    def euler_zyx_rad(self) -> np.ndarray:
        return rm.euler_xyz_intrinsic(self.xyzw)[::-1]

    @property
    def euler_zyx_deg(self) -> np.ndarray:
        return np.degrees(self.euler_zyx_rad)

    @property
    def conjugate(self) -> "Quaternion":
        return Quaternion(w=self.w, v=-self.v)

    @property
    def angle(self) -> float:
        """asin|v| -- half the rotation angle (Quaternion.py:138-147)."""
        return math.asin(float(np.linalg.norm(self.v)))

    @property
    def axis(self) -> np.ndarray:
        n = float(np.linalg.norm(self.v))
        return np.zeros(3) if math.isclose(math.asin(n), 0) else self.v / n

    def __mul__(self, other):
        if isinstance(other, Quaternion):
            return Quaternion(val=rm.quat_multiply(self.xyzw, other.xyzw), do_normalise=True)
        if isinstance(other, (float, int)):
            return Quaternion(val=self.xyzw * other, do_normalise=True)
        if isinstance(other, np.ndarray) and other.shape == (3, 3):
            return self * Quaternion(val=other, do_normalise=True)
        raise TypeError

    def __add__(self, other):
        return Quaternion(val=self.xyzw + other.xyzw, do_normalise=True)

    def __sub__(self, other):
        return Quaternion(val=self.xyzw - other.xyzw, do_normalise=True)

    def normalise(self) -> None:
        self.x, self.y, self.z, self.w = (float(c) for c in rm.quat_normalise(self.xyzw))

    @staticmethod
    def about_axis(angle: float, axis) -> "Quaternion":
        """Quaternion.py:209-224."""
        u = np.asarray(axis, dtype=float).reshape(3)
        n = float(np.linalg.norm(u))
        vec = u * (math.sin(angle / 2.0) / n) if n > np.finfo(float).eps * 4.0 else u.copy()
        return Quaternion(val=np.array([*vec, math.cos(angle / 2.0)]), do_normalise=True)
