"""``Config("config.yaml")`` -- same YAML schema and derived quantities as the
reference's ``dvi_ekf.tools.config.Config`` (config.py:214-272, config.yaml:1-81),
implemented with plain classes (the reference pins pydantic 1.x; nothing here
depends on it).

Extra, optional top-level section understood by this build (absent keys keep the
reference behaviour)::

    backend:
      batch: 1              # number of independent filter instances on the GPU
      golden_compat: false  # G1/G2/G3 plumbing of SURVEY.md 8(c) (hold frozen DOFs,
                            # 1 IMU sample per frame in fast-sim, intrinsic-XYZ rates)
"""
from __future__ import annotations

import math
from pathlib import Path
from types import SimpleNamespace
from typing import List, Optional, Union

import numpy as np
import yaml


def _deg2rad_list(v) -> List[float]:
    return [math.radians(float(x)) for x in v]


class CameraNoise:
    """config.py:60-77 -- theta / notch are given in degrees in the YAML."""

    def __init__(self, position, theta, notch):
        self.position = [float(x) for x in position]
        self.theta = _deg2rad_list(theta)
        self.notch = math.radians(float(notch))

    @property
    def vec(self) -> np.ndarray:
        return np.hstack((self.position, self.theta, self.notch))


class CameraConfig:
    """config.py:80-93."""

    def __init__(self, scale, noise, start_frame=None, total_frames=None, with_notch=False):
        self.start_frame = start_frame
        self.total_frames = None if total_frames == "all" else total_frames
        self.scale = int(scale)
        self.with_notch = bool(with_notch)
        self.noise = CameraNoise(**noise)


class ImuConfig:
    """config.py:96-120."""

    def __init__(self, interframe_vals, noise_sample_rate, gravity):
        self.interframe_vals = int(interframe_vals)
        self.noise_sample_rate = float(noise_sample_rate)
        self.gravity = float(gravity)

    @property
    def noise_gyro(self) -> float:
        return 0.005 * math.sqrt(self.noise_sample_rate)  # deg/s

    @property
    def stdev_omega(self) -> List[float]:
        return [math.radians(self.noise_gyro)] * 3  # rad/s

    @property
    def stdev_accel(self) -> np.ndarray:
        return np.array([400 * 1e-6 * self.gravity * math.sqrt(self.noise_sample_rate)] * 3)


class ModelConfig:
    """config.py:123-133 -- angle in degrees in the YAML."""

    def __init__(self, length, angle):
        self.length = float(length)
        self.angle = math.radians(float(angle))


class States:
    """config.py:136-167; the four angle groups are degrees in the YAML."""

    def __init__(self, imu_pos, imu_vel, imu_theta, dofs_rot, dofs_trans, notch, camera_pos, camera_theta):
        self.imu_pos = [float(x) for x in imu_pos]
        self.imu_vel = [float(x) for x in imu_vel]
        self.imu_theta = _deg2rad_list(imu_theta)
        self.dofs_rot = _deg2rad_list(dofs_rot)
        self.dofs_trans = [float(x) for x in dofs_trans]
        self.notch = _deg2rad_list(notch)
        self.camera_pos = [float(x) for x in camera_pos]
        self.camera_theta = _deg2rad_list(camera_theta)

    @property
    def vec(self) -> np.ndarray:
        return np.hstack(
            (self.imu_pos, self.imu_vel, self.imu_theta, self.dofs_rot, self.dofs_trans, self.notch,
             self.camera_pos, self.camera_theta)
        )


class InitialConditions:
    """config.py:170-180."""

    def __init__(self, cov0, x0):
        self.cov0 = States(**cov0)
        self.x0 = States(**x0)

    @property
    def cov0_matrix(self) -> np.ndarray:
        return np.square(np.diag(self.cov0.vec))


class DofsNoise:
    """config.py:183-198 -- rotation / notch_accel in degrees in the YAML."""

    def __init__(self, translation, rotation, notch_accel):
        self.translation = [float(x) for x in translation]
        self.rotation = _deg2rad_list(rotation)
        self.notch_accel = math.radians(float(notch_accel))

    @property
    def vec(self) -> np.ndarray:
        return np.hstack((self.rotation, self.translation, self.notch_accel))


class FilterConfig:
    """config.py:201-211."""

    def __init__(self, ic, process_noise):
        self.ic = InitialConditions(**ic)
        self.process_noise = process_noise

    @property
    def noise_dofs(self) -> DofsNoise:
        return DofsNoise(**self.process_noise["dofs"])


class Config:
    """Settings for the simulation, models, trajectories and initial conditions."""

    def __init__(self, filepath: Union[Path, str], base_dir: Optional[Union[Path, str]] = None):
        filepath = Path(filepath)
        with open(filepath) as f:
            cfg = yaml.safe_load(f)
        base = Path(base_dir) if base_dir is not None else Path(".")

        s = cfg["simulation"]
        traj_path = Path(s["traj_path"])
        if not traj_path.is_absolute():
            traj_path = base / traj_path
        self.sim = SimpleNamespace(
            mode=str(s["mode"]),
            img_path=Path(s["img_path"]),
            traj_path=traj_path,
            traj_name=Path(str(s["traj_name"])),
            notch_traj_name=Path(str(s["notch_traj_name"])),
            num_kf_runs=int(s.get("num_kf_runs", 1)),
            frozen_dofs=[int(v) for v in s["frozen_dofs"]],
            do_plot=bool(s["do_plot"]),
            do_fast_sim=bool(s["do_fast_sim"]),
        )
        self.sim.traj_fp = traj_path / f"{self.sim.traj_name}.csv"
        self.sim.notch_fp = traj_path / f"{self.sim.notch_traj_name}.csv"
        if not self.sim.traj_fp.exists():
            raise FileNotFoundError(self.sim.traj_fp)

        self.camera = CameraConfig(**cfg["camera"])
        self.imu = ImuConfig(**cfg["imu"])
        self.model = ModelConfig(**cfg["model"])
        self.filter = FilterConfig(**cfg["filter"])

        be = cfg.get("backend", {}) or {}
        self.batch = int(be.get("batch", 1))
        self.golden_compat = bool(be.get("golden_compat", False))

        # simulation params (config.py:234-248)
        self.interframe_vals = self.imu.interframe_vals
        self.cov0_matrix = self.filter.ic.cov0_matrix
        self.with_notch = self.camera.with_notch
        self.max_vals = self.camera.total_frames
        self.min_t: Optional[float] = None
        self.max_t: Optional[float] = None
        self.total_data_pts: Optional[int] = None
        if self.sim.do_fast_sim:
            self.interframe_vals = 1
            self.max_vals = 10

        # probe (config.py:250-255)
        self.frozen_dofs: List[int] = self.sim.frozen_dofs
        self.gt_joint_dofs = None
        self.gt_imu_dofs = None
        self.ic_imu_dofs = None
        self._dofs_updated = False

        # noise (config.py:257-261)
        self.process_noise_rw_std = self.filter.noise_dofs.vec / self.interframe_vals
        self.process_noise_rw_var = np.square(self.process_noise_rw_std)
        self.meas_noise_std = self.camera.noise.vec
        self.meas_noise_var = np.square(self.meas_noise_std)

        self.img_path = self.sim.img_path
        self.traj_path = self.sim.traj_path
        self.traj_name = self.sim.traj_name
        self.notch_traj_name = self.sim.notch_traj_name
        self.traj_fp = self.sim.traj_fp
        self.notch_fp = self.sim.notch_fp
        self.mse: Optional[float] = None
        self.do_plot: bool = self.sim.do_plot

    @property
    def dofs_updated(self) -> bool:
        return self._dofs_updated

    def update_dofs(self, probe) -> None:
        """config.py:278-283 (perfect initial DOFs: utils.py:16-46)."""
        self.gt_joint_dofs = [np.array(v, dtype=float) for v in probe.joint_dofs]
        self.gt_imu_dofs = np.array(probe.imu_dofs, dtype=float)
        self.ic_imu_dofs = self.gt_imu_dofs.copy()
        self._dofs_updated = True
