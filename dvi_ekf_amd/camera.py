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
        self.with_notch = False
        self.is_rotated = False
        self.v = self.acc = self.om = self.alp = None
        if table is not None:
            self.gen_angle_arrays()

    @classmethod
    def from_csv(cls, name: str, filepath, cap: Optional[int] = None, scale: Optional[float] = None,
                 start_at: Optional[float] = None, notch_filepath=None, with_notch: bool = False) -> "VisualTraj":
        """files.get_dataframe (files.py:19-28) + _filter_values + _scale_values
        (VisualTrajectory.py:76-108); with ``with_notch`` also the notch columns
        (VisualTrajectory.py:66-70)."""
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
        i0 = 0
        if start_at:
            i0 = int(np.argwhere(table["t"] == start_at).item())
            table = {k: v[i0:(i0 + cap if cap else None)] for k, v in table.items()}
        if scale:
            for k in ("x", "y", "z"):
                table[k] = table[k] * scale
        traj = cls(name, table)
        if with_notch:
            traj.read_notch_from_file(notch_filepath, start_index=i0, cap=cap)
        return traj

    def read_notch_from_file(self, notch_filepath, start_index: int = 0, cap: Optional[int] = None) -> None:
        """One line per camera frame: notch angle [rad], its rate and its acceleration.

        This is what VisualTrajectory.read_notch_from_file (110-118) is written to do; as
        shipped in v1 it cannot run (it whitespace-splits the comma-separated ``notch*.csv``
        and appends to attributes that are still ``None``), so there is no reference output
        to match -- the columns are taken as the file gives them, for either separator, and
        cut to the same frame window as the pose columns (VisualTrajectory.py:69-70)."""
        if notch_filepath is None:
            raise ValueError("camera.with_notch needs a notch trajectory file")
        rows = []
        with open(str(notch_filepath)) as f:
            for ln in f:
                parts = ln.replace(",", " ").split()
                if parts:
                    if len(parts) < 3:
                        raise ValueError(f"{notch_filepath}: expected 3 columns per line, got {len(parts)}")
                    rows.append([float(v) for v in parts[:3]])
        data = np.asarray(rows, dtype=float).reshape(-1, 3)
        stop = start_index + cap if cap else (start_index + len(self.t) if start_index else None)
        data = data[start_index:stop]
        if len(data) != len(self.t):
            raise ValueError(f"{notch_filepath}: {len(data)} notch samples for {len(self.t)} camera frames")
        self.notch, self.notch_d, self.notch_dd = data[:, 0].copy(), data[:, 1].copy(), data[:, 2].copy()
        self.with_notch = True

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
        self.euler_rate = euler_rate
        self.traj = traj if traj is not None else VisualTraj.from_csv(
            "camera", filepath, cap=max_vals, scale=scale, start_at=start_at, notch_filepath=notch_filepath,
            with_notch=with_notch)
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
        # notch (Camera.py:120-137)
        self.with_notch = bool(with_notch)
        self.is_rotated = False
        self.rotated: Optional["Camera"] = None
        self.notch = self.notch_d = self.notch_dd = None
        if self.flag_interpolated and self.traj.with_notch:
            self.notch, self.notch_d, self.notch_dd = self.traj.notch, self.traj.notch_d, self.traj.notch_dd
        if with_notch:
            if self.traj.notch is None:
                raise ValueError("with_notch: the trajectory carries no notch columns")
            self.notch, self.notch_d, self.notch_dd = self.traj.notch, self.traj.notch_d, self.traj.notch_dd
            self.gen_rotated()

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
        return self.interpolate_to(np.linspace(t_old[0], t_old[-1], num=n_new), interframe_vals)

    def interpolate_to(self, t_new, interframe_vals: int = 0) -> "Camera":
        """The interpolation of ``interpolate`` on an arbitrary time grid (an IMU clock that is not a
        multiple of the frame rate: 1 kHz against 30 Hz, BASELINE config #5)."""
        t_old = self.t
        t_new = np.asarray(t_new, dtype=float)
        table = {"t": t_new}
        for c in _COLS[1:]:
            table[c] = np.interp(t_new, t_old, getattr(self.traj, c))
        out = VisualTraj(self.traj.name + " interpl", table)
        for lab in ("v", "acc", "om", "alp"):
            src = getattr(self, lab)
            setattr(out, lab, np.asarray([np.interp(t_new, t_old, src[i]) for i in range(3)]))
        if self.with_notch:                                  # Interpolator.py:48-52,82-88
            for lab in ("notch", "notch_d", "notch_dd"):
                setattr(out, lab, np.interp(t_new, t_old, getattr(self, lab)))
            out.with_notch = True
        out.interframe_vals = interframe_vals
        out.is_interpolated = True
        cam = Camera(traj=out, euler_rate=self.euler_rate)
        cam.with_notch = self.with_notch                     # Camera.py:154-155
        return cam

    def gen_rotated(self) -> None:
        """The camera as the probe's notch joint turns it: every orientation is
        pre-multiplied by a rotation about z by the notch angle (Camera.py:172-202).
        Positions are shared; rates are re-derived from the rotated poses."""
        table = {c: np.array(getattr(self.traj, c), dtype=float, copy=True) for c in _COLS}
        rotated_quats = []
        for i in range(len(self.t)):
            notch_quat = Quaternion(val=np.array([0.0, 0.0, self.get_notch_vec_at(i)[0]]), euler="xyz")
            rotated_quats.append(notch_quat * self.traj.quats[i])
        for j, c in enumerate(("qx", "qy", "qz", "qw")):
            table[c] = np.array([q.xyzw[j] for q in rotated_quats])
        rotated_traj = VisualTraj("camera rot", table)
        rotated_traj.quats = rotated_quats                   # the products themselves, not re-normalised copies
        rotated_traj.is_rotated = True
        self.rotated = Camera(filepath=None, traj=rotated_traj, euler_rate=self.euler_rate)
        self.rotated.is_rotated = True
        self.rotated.with_notch = self.with_notch
        self.rotated.notch, self.rotated.notch_d, self.rotated.notch_dd = self.notch, self.notch_d, self.notch_dd

    def generate_notch_values(self, ang_fractions=(0, 0, 0.9, 0.9, -0.45, -0.45),
                              partitions=(0, 0.1, 0.45, 0.5, 0.9, 1)) -> np.ndarray:
        """Piecewise septic smooth-step notch trajectory over the camera's time base
        (Camera._gen_notch_values / _gen_t_partition, Camera.py:209-274): the angle
        rests at pi*ang_fractions[k] at the partition boundaries and moves between them
        with zero rate, acceleration and jerk at both ends.  Returns [n,3] = angle, rate,
        acceleration and installs the columns (then ``gen_rotated()`` is up to date)."""
        ang_vals = np.pi * np.asarray(ang_fractions, dtype=float)
        bounds = [int(np.ceil(p * self.max_vals)) for p in partitions]
        cols = [[], [], []]
        ang_prev, t_prev = ang_vals[0], float(self.t[0])
        for k, ang_k in enumerate(ang_vals[1:]):
            t_seg = np.asarray(self.t[bounds[k]:bounds[k + 1]], dtype=float)
            t_end = float(t_seg[-1])
            tau, T = t_seg - t_prev, t_end - t_prev
            d = ang_k - ang_prev
            if T == 0.0:                                      # a one-sample first segment at rest
                z, zd, zdd = ang_prev + 0 * tau, 0 * tau, 0 * tau
            else:
                z = ang_prev + d * (35 * (tau / T) ** 4 - 84 * (tau / T) ** 5 + 70 * (tau / T) ** 6 - 20 * (tau / T) ** 7)
                zd = d * (35 * 4 / T ** 4 * tau ** 3 - 84 * 5 / T ** 5 * tau ** 4
                          + 70 * 6 / T ** 6 * tau ** 5 - 20 * 7 / T ** 7 * tau ** 6)
                zdd = d * (35 * 4 * 3 / T ** 4 * tau ** 2 - 84 * 5 * 4 / T ** 5 * tau ** 3
                           + 70 * 6 * 5 / T ** 6 * tau ** 4 - 20 * 7 * 6 / T ** 7 * tau ** 5)
            for c, v in zip(cols, (z, zd, zdd)):
                c.append(v)
            t_prev, ang_prev = t_end, ang_k
        self.notch, self.notch_d, self.notch_dd = (np.concatenate(c).ravel() for c in cols)
        self.traj.notch, self.traj.notch_d, self.traj.notch_dd = self.notch, self.notch_d, self.notch_dd
        self.traj.with_notch = self.with_notch = True
        self.gen_rotated()
        return np.stack([self.notch, self.notch_d, self.notch_dd], axis=1)

    def get_notch_vec_at(self, i: int) -> np.ndarray:
        """Camera.py:276-280."""
        if self.with_notch:
            return np.array([self.notch[i], self.notch_d[i], self.notch_dd[i]])
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
        def notch_at(i):
            if self.with_notch:
                return float(self.notch[i]), float(self.notch_d[i]), float(self.notch_dd[i])
            return 0.0, 0.0, 0.0

        return [
            CameraMeasurementPoint(self.t[i], self.p[:, i], self.R[i], self.v[:, i], self.om[:, i],
                                   self.acc[:, i], self.alp[:, i], *notch_at(i))
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
    assert (cam.rotated is not None) == bool(config.with_notch)
    cam.update_sim_params(config)
    return cam
