"""Septic smooth-step camera path -- the generator behind BASELINE config #5
(1 kHz IMU / 30 Hz vision, 10 000 IMU steps).

Follows the reference's ``data/trajs/smooth_traj.py``: ``traj_gen`` (lines 6-12) moves a
coordinate from ``z0`` to ``zT`` with zero rate, acceleration and jerk at both ends,
``gen_t_partition`` / ``gen_values`` (15-39) chain such moves over index partitions of the
frame axis and ``gen_values_rot`` (42-56) turns an angle profile into quaternions about one
axis.  The reference's ``__main__`` multiplies every amplitude by zero (a camera at rest,
60 frames, written with 3 decimals); here the amplitudes are arguments, the samples keep
full precision and the frame rate is a parameter (30 Hz for config #5).
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

from .camera import Camera, VisualTraj


def traj_gen(z0: float, zT: float, t, t_prev: float, T: float) -> np.ndarray:
    """smooth_traj.py:6-12: z0 + (zT - z0) (35 s^4 - 84 s^5 + 70 s^6 - 20 s^7), s = (t - t_prev) / (T - t_prev)."""
    s = (np.asarray(t, dtype=float) - t_prev) / (T - t_prev)
    return z0 + (zT - z0) * (35 * s ** 4 - 84 * s ** 5 + 70 * s ** 6 - 20 * s ** 7)


def gen_t_partition(t, max_vals: int, partitions: Sequence[float]):
    """smooth_traj.py:15-22: the frame axis cut at ceil(p * max_vals)."""
    parts, p_prev = [], 0
    for p in partitions[1:]:
        p_k = int(math.ceil(p * max_vals))
        parts.append(np.array(t[p_prev:p_k], dtype=float))
        p_prev = p_k
    return parts


def gen_values(t, vals: Sequence[float], max_vals: int, partitions: Sequence[float]) -> np.ndarray:
    """smooth_traj.py:25-39: vals[k] is reached at the end of partition k."""
    t_part = gen_t_partition(t, max_vals, partitions)
    t_prev, vprev = t_part[0][0], vals[0]
    out = []
    for i, vk in enumerate(vals[1:]):
        t_max = t_part[i][-1]
        out.append(traj_gen(vprev, vk, t_part[i], t_prev, t_max))
        t_prev, vprev = t_max, vk
    return np.concatenate(out).ravel()


def gen_values_rot(t, vals_deg: Sequence[float], max_vals: int, axis: str, partitions: Sequence[float]):
    """smooth_traj.py:42-56: rotation about one axis by the angle profile, as xyzw quaternion columns
    (the reference goes through ``SE3.R{axis}(ang, 'deg')`` and scipy; the quaternion of a rotation
    about a coordinate axis is written out here)."""
    if axis not in ("x", "y", "z"):
        raise ValueError(axis)
    half = 0.5 * np.radians(gen_values(t, vals_deg, max_vals, partitions))
    q = np.zeros((len(half), 4))
    q[:, "xyz".index(axis)] = np.sin(half)
    q[:, 3] = np.cos(half)
    return q[:, 0], q[:, 1], q[:, 2], q[:, 3]


def smooth_camera(num_frames: int = 301, rate_hz: float = 30.0, amplitude_xyz=(6.0, -4.0, 3.0), angle_deg: float = 25.0,
                  axis: str = "z", shape=(0.0, 1.0, -0.5, 0.5, 0.0), partitions=(0.0, 0.17, 0.45, 0.75, 1.0),
                  euler_rate: str = "xyz") -> Camera:
    """A ``Camera`` on the septic smooth-step path: frame k at t = k / rate_hz.  ``shape`` are the rest
    values (fractions of the amplitudes) reached at the partition boundaries -- the reference's script
    uses (0, 1, 1) over (0, 0.17, 1); the default keeps the camera moving over the whole run."""
    if len(shape) != len(partitions):
        raise ValueError("one rest value per partition boundary")
    idx = np.linspace(0, num_frames - 1, num_frames)           # the reference's frame axis (smooth_traj.py:68)
    sh = np.asarray(shape, dtype=float)
    cols = [gen_values(idx, a * sh, num_frames, partitions) for a in amplitude_xyz]
    qx, qy, qz, qw = gen_values_rot(idx, angle_deg * sh, num_frames, axis, partitions)
    table = {"t": idx / float(rate_hz), "x": cols[0], "y": cols[1], "z": cols[2], "qx": qx, "qy": qy, "qz": qz, "qw": qw}
    return Camera(traj=VisualTraj("smooth camera", table), euler_rate=euler_rate)
