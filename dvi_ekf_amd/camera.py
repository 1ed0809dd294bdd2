"""Camera trajectory: CSV reader, finite-difference rates, linear interpolation.

One-time host preparation (NumPy).  Numeric recipe follows the reference's
``models/Camera.py`` (52-170, 299-347), ``trajectory/VisualTrajectory.py``
(99-149), ``trajectory/Interpolator.py`` (25-88) and ``tools/files.py`` (19-28);
containers are this build's own.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import rotmath as rm
from .quaternion import Quaternion

_COLS = ("t", "x", "y", "z", "qx", "qy", "qz", "qw")


@dataclass(frozen=True)
class VisualMeasurementPoint:
    """One SLAM pose (measurement_point/VisualMeasurementPoint.py)."""
    t: float
    x: float
    y: float
    z: float
    q: Quaternion  # raw file values, NOT normalised (VisualTrajectory.py:120-134)

    @property
    def pos(self) -> np.ndarray:
        return np.array([self.x, self.y, self.z])

    @property
    def vec(self) -> np.ndarray:
        return np.hstack((self.pos, self.q.xyzw))


@dataclass(frozen=True)
class CameraMeasurementPoint:
    """One interpolated camera sample with its rates."""
    t: float
    p: np.ndarray
    R: np.ndarray
    v: np.ndarray
    om: np.ndarray
    acc: np.ndarray
    alp: np.ndarray
    notch: float
    notch_d: float
    notch_dd: float

    @property
    def vec(self) -> List[np.ndarray]:
        return [self.p, self.R, self.v, self.om, self.acc, self.alp]

    @property
    def notch_arr(self) -> List[float]:
        return [self.notch, self.notch_d, self.notch_dd]


class VisualTraj:
    """Time + pose columns of a camera trajectory."""

    labels = list(_COLS)

    def __init__(self, name: str, table: Optional[dict] = None):
        self.name = name
        self.is_interpolated = False
        self.interframe_vals = 0
        for c in _COLS:
            setattr(self, c, None if table is None else np.asarray(table[c], dtype=float))
        self.quats: List[Quaternion] = []
        self.notch = self.notch_d = self.notch_dd = None
        self.v = self.acc = self.om = self.alp = None
        if table is not None:
            self.gen_angle_arrays()

    @classmethod
    def from_csv(cls, name: str, filepath, cap: Optional[int] = None, scale: Optional[float] = None,
                 start_at: Optional[float] = None) -> "VisualTraj":
        """files.get_dataframe (files.py:19-28) + _filter_values + _scale_values
        (VisualTrajectory.py:76-108)."""
        filepath = str(filepath)
        if not filepath.endswith(".csv"):
            raise ValueError("camera trajectory must be a .csv file")
        with open(filepath) as f:
            header = f.readline().strip().split(",")
            rows = [ln.strip().split(",") for ln in f if ln.strip()]
        if cap:
            rows = rows[:cap]
        data = np.array(rows, dtype=float)
        table = {h: data[:, header.index(h)] for h in _COLS}
        if start_at:
            i0 = int(np.argwhere(table["t"] == start_at).item())
            table = {k: v[i0:(i0 + cap if cap else None)] for k, v in table.items()}
        if scale:
            for k in ("x", "y", "z"):
                table[k] = table[k] * scale
        return cls(name, table)

    def gen_angle_arrays(self) -> None:
        """Per-frame unit quaternions (VisualTrajectory.py:136-149)."""
        self.quats = [
            Quaternion(x=self.qx[i], y=self.qy[i], z=self.qz[i], w=self.qw[i], do_normalise=True)
            for i in range(len(self.qw))
        ]
        # extrinsic xyz Euler angles in degrees (VisualTrajectory.py:151-162)
        eul = np.array([q.euler_xyz_deg for q in self.quats])
        self.rx_deg, self.ry_deg, self.rz_deg = eul[:, 0], eul[:, 1], eul[:, 2]

    def at_index(self, index: int) -> VisualMeasurementPoint:
        q = Quaternion(x=self.qx[index], y=self.qy[index], z=self.qz[index], w=self.qw[index])
        return VisualMeasurementPoint(self.t[index], self.x[index], self.y[index], self.z[index], q)


class Camera:
    """Camera sensor: poses p, R (q) and their rates v, om, acc, alp."""

    def __init__(self, filepath=None, notch_filepath=None, traj: Optional[VisualTraj] = None,
                 max_vals: Optional[int] = None, scale: Optional[float] = None, with_notch: bool = False,
                 start_at=None, euler_rate: str = "xyz"):
        if with_notch:
            # the reference's notch reader is broken in v1 (VisualTrajectory.py:110-118
            # whitespace-splits a comma-separated file); SURVEY.md 8(f) item 4
            raise NotImplementedError("camera.with_notch is not supported (broken in the reference v1)")
        self.euler_rate = euler_rate
        self.traj = traj if traj is not None else VisualTraj.from_csv(
            "camera", filepath, cap=max_vals, scale=scale, start_at=start_at)
        self.max_vals = len(self.traj.t)
        self.t = self.traj.t
        self.dt = float(self.t[1] - self.t[0])
        self.min_t = float(self.t[0])
        self.max_t = float(self.t[-1])

        quats = self.traj.quats
        self.p = np.array((self.traj.x, self.traj.y, self.traj.z))
        self.r = np.array([q.euler_xyz_rad for q in quats]).T
        self.R = [q.rot for q in quats]
        self.q = np.array([q.xyzw for q in quats]).T

        if self.traj.is_interpolated:
            self.v, self.om, self.acc, self.alp = self.traj.v, self.traj.om, self.traj.acc, self.traj.alp
        else:
            # Camera._calc_derived_data (Camera.py:158-170)
            self.v = np.gradient(self.p, self.dt, axis=-1)
            self.acc = np.gradient(self.v, self.dt, axis=-1)
            if euler_rate == "xyz":      # v1 source: extrinsic xyz Euler angles (Camera.py:168)
                ang = self.r
            elif euler_rate == "XYZ":    # golden-compat G3: as_euler('zyx')[::-1]
                ang = np.array([rm.euler_xyz_intrinsic(q.xyzw) for q in quats]).T
            else:
                raise ValueError(euler_rate)
            self.om = np.gradient(ang, self.dt, axis=-1)
            self.alp = np.gradient(self.om, self.dt, axis=-1)

        self.p0 = self.p[:, 0].copy().reshape(3, 1)
        self.q0 = Quaternion(val=self.q[:, 0].copy(), do_normalise=True)
        self.vec0 = self.vec_at(0)
        self.with_notch = False
        self.rotated = None
        self.notch = self.notch_d = self.notch_dd = None

    @property
    def flag_interpolated(self) -> bool:
        return self.traj.is_interpolated

    def update_sim_params(self, config) -> None:
        """Camera.update_sim_params (Camera.py:143-148)."""
        config.max_vals = self.max_vals
        config.min_t = self.min_t
        config.max_t = self.max_t
        config.total_data_pts = (self.max_vals - 1) * config.interframe_vals + 1

    def interpolate(self, interframe_vals: int) -> "Camera":
        """Linear interpolation of every column, raw quaternion components
        included, then re-normalisation (Interpolator.py:25-88)."""
        t_old = self.t
        n_new = (len(t_old) - 1) * interframe_vals + 1
        t_new = np.linspace(t_old[0], t_old[-1], num=n_new)
        table = {"t": t_new}
        for c in _COLS[1:]:
            table[c] = np.interp(t_new, t_old, getattr(self.traj, c))
        out = VisualTraj(self.traj.name + " interpl", table)
        for lab in ("v", "acc", "om", "alp"):
            src = getattr(self, lab)
            setattr(out, lab, np.asarray([np.interp(t_new, t_old, src[i]) for i in range(3)]))
        out.interframe_vals = interframe_vals
        out.is_interpolated = True
        return Camera(traj=out, euler_rate=self.euler_rate)

    def get_notch_vec_at(self, i: int) -> np.ndarray:
        return np.array([0.0, 0.0, 0.0])

    def vec_at(self, i: int) -> List[np.ndarray]:
        return [self.p[:, i].copy().reshape(3, 1), self.R[i].copy(), self.v[:, i].copy().reshape(3, 1),
                self.om[:, i].copy().reshape(3, 1), self.acc[:, i].copy().reshape(3, 1),
                self.alp[:, i].copy().reshape(3, 1)]

    def samples_between(self, old_t: float, new_t: float) -> List[CameraMeasurementPoint]:
        """Samples with old_t < t_i <= new_t (CameraMeasurementArray.from_camera,
        Camera.py:320-347)."""
        old_i = max(i for i, t in enumerate(self.t) if t <= old_t)
        new_i = max(i for i, t in enumerate(self.t) if t <= new_t)
        return [
            CameraMeasurementPoint(self.t[i], self.p[:, i], self.R[i], self.v[:, i], self.om[:, i],
                                   self.acc[:, i], self.alp[:, i], 0.0, 0.0, 0.0)
            for i in range(old_i + 1, new_i + 1)
        ]


def create_camera(config) -> Camera:
    """models/Camera.py:20-38."""
    cam = Camera(
        filepath=config.traj_fp,
        notch_filepath=config.notch_fp,
        max_vals=config.max_vals,
        scale=config.camera.scale,
        with_notch=config.with_notch,
        start_at=config.camera.start_frame,
        euler_rate="XYZ" if getattr(config, "golden_compat", False) else "xyz",
    )
    cam.update_sim_params(config)
    return cam
