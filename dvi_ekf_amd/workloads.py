"""Synthetic batch workloads for benchmarking and Monte-Carlo runs.

``monte_carlo_pairs`` builds BASELINE config #2/#4: B independent filters follow the
reference's default run (mandala0_mono, default config.yaml) on the 1:1 schedule --
one predict at dt = 1.0 then one update per camera frame -- with per-instance
perturbed initial extrinsic DOFs and per-instance additive Gaussian noise on the IMU
and camera measurements.  The noise is generated on the device by ``dvi_eskf_randn``
(Philox4x32-10, key = seed * 2^32, counter = (instance, step, lane); Box-Muller -- the generator
SURVEY.md 8(d) makes part of the contract): what an instance draws depends on its global index
only, not on the batch or rank it is generated in.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch

from . import noise
from .camera import create_camera
from .config import Config
from .imu import Imu
from .probe import SimpleProbe
from .schedule import RunInputs, build_run_inputs, noise_matrices

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def default_run_inputs(config_path=None, base_dir=None, golden_compat=True, all_frames=True, frozen=None):
    """Host-side preparation of the reference's default run.  Returns
    (cfg, RunInputs, Q, R, sigma_a, sigma_om)."""
    config_path = config_path or os.path.join(REPO_ROOT, "config.yaml")
    base_dir = base_dir or os.path.dirname(os.path.abspath(config_path))
    cfg = Config(config_path, base_dir=base_dir)
    cfg.golden_compat = bool(golden_compat)
    if all_frames:
        cfg.max_vals = None
    if all_frames and not golden_compat:
        cfg.interframe_vals = cfg.imu.interframe_vals
        cfg.process_noise_rw_std = cfg.filter.noise_dofs.vec / cfg.interframe_vals
        cfg.process_noise_rw_var = np.square(cfg.process_noise_rw_std)
    if frozen is not None:
        cfg.frozen_dofs = list(frozen)
    probe = SimpleProbe(cfg.model.length, cfg.model.angle)
    cfg.update_dofs(probe)
    cam = create_camera(cfg)
    n_if = cfg.interframe_vals if golden_compat else cfg.imu.interframe_vals
    imu = Imu.create(SimpleNamespace(interframe_vals=n_if, stdev_accel=cfg.imu.stdev_accel,
                                     stdev_omega=cfg.imu.stdev_omega), cam, probe, gen_ref=True)
    ri = build_run_inputs(cfg, cam, imu, probe)
    Q, R = noise_matrices(cfg, imu.stdev_na, imu.stdev_nom)
    return cfg, ri, Q, R, np.array(imu.stdev_na), np.array(imu.stdev_nom)


@dataclass
class PairWorkload:
    """H predict+update pairs for B instances, all tensors on the device."""
    cfg: object
    ri: RunInputs
    Q: np.ndarray
    R: np.ndarray
    sigma_a: np.ndarray
    sigma_om: np.ndarray
    x0: torch.Tensor      # [B,26]
    P0: torch.Tensor      # [24,24]
    om0: torch.Tensor     # [3]
    acc0: torch.Tensor    # [3]
    dt: torch.Tensor      # [H,B]
    om: torch.Tensor      # [H,B,3]
    acc: torch.Tensor     # [H,B,3]
    pos: torch.Tensor     # [H,B,3]
    quat: torch.Tensor    # [H,B,4]
    notch: torch.Tensor   # [H,B]

    @property
    def horizon(self) -> int:
        return self.dt.shape[0]

    def make_filter(self, dtype=torch.float64):
        from .batch import BatchedESKF
        B = self.x0.shape[0]
        f = BatchedESKF(B, length=self.cfg.model.length, theta_cam=self.cfg.model.angle,
                        frozen=self.cfg.frozen_dofs, sigma_a=self.sigma_a, sigma_om=self.sigma_om,
                        hold_frozen=True, dtype=dtype, device=self.x0.device)
        f.set_noise(self.Q, self.R)
        f.set_state(x=self.x0.to(dtype), P=self.P0.to(dtype), om_old=self.om0.to(dtype), acc_old=self.acc0.to(dtype))
        return f


def monte_carlo_pairs(batch: int, horizon: int = 16, seed: int = 4, device="cuda", dtype=torch.float64,
                      first_instance: int = 0, frozen=(0, 0, 0, 0, 0, 0)) -> PairWorkload:
    """Instances [first_instance, first_instance + batch) of the Monte-Carlo batch.  Global
    instance 0 is the unperturbed, noise-free golden-compat run."""
    cfg, ri, Q, R, sa, so = default_run_inputs(golden_compat=True, all_frames=True, frozen=frozen)
    H = int(horizon)
    if H > len(ri.t_meas):
        raise ValueError("horizon exceeds the trajectory")
    dev = torch.device(device)
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype, device=dev)
    B = int(batch)
    key = int(seed) << 32
    # lanes of one pair-step: 0..2 gyro, 3..5 accelerometer, 6..8 camera position, 9..11 camera
    # attitude, 12 notch angle; the initial condition is drawn at step word IC_STEP, lanes 0..5
    z_ic = noise.randn(key, first_instance, B, noise.IC_STEP, 1, 6, dtype=dtype, device=dev)[0]
    z = noise.randn(key, first_instance, B, 0, H, 13, dtype=dtype, device=dev)

    x0 = t(ri.x0).repeat(B, 1)
    cov0 = np.sqrt(np.diag(ri.P0))
    x0[:, 10:16] += z_ic * t(cov0[9:15])                 # IC dofs = gt + N(0, cov0 dofs block)
    meas_std = np.sqrt(np.diag(R))
    dt = t(ri.dt[:H]).reshape(H, 1).repeat(1, B).contiguous()
    om = t(ri.om[:H]).reshape(H, 1, 3) + z[:, :, 0:3] * t(so)
    acc = t(ri.acc[:H]).reshape(H, 1, 3) + z[:, :, 3:6] * t(sa)
    pos = t(ri.pos[:H]).reshape(H, 1, 3) + z[:, :, 6:9] * t(meas_std[0:3])
    # small-angle rotation noise on the measured attitude: q <- q * (theta/2, 1)
    q = t(ri.quat[:H]).reshape(H, 1, 4).repeat(1, B, 1)
    th = z[:, :, 9:12] * t(meas_std[3:6]) * 0.5
    qx, qy, qz, qw = q.unbind(-1)
    tx, ty, tz = th.unbind(-1)
    quat = torch.stack([qw * tx + qx + (qy * tz - qz * ty), qw * ty + qy + (qz * tx - qx * tz),
                        qw * tz + qz + (qx * ty - qy * tx), qw - (qx * tx + qy * ty + qz * tz)], dim=-1)
    notch = (t(ri.notch[:H]).reshape(H, 1) + z[:, :, 12] * float(meas_std[6])).contiguous()
    if first_instance == 0:   # the reference run itself
        x0[0] = t(ri.x0)
        om[:, 0], acc[:, 0], pos[:, 0], quat[:, 0] = t(ri.om[:H]), t(ri.acc[:H]), t(ri.pos[:H]), t(ri.quat[:H])
        notch[:, 0] = t(ri.notch[:H])
    return PairWorkload(cfg, ri, Q, R, sa, so, x0, t(ri.P0), t(ri.om0), t(ri.acc0), dt, om.contiguous(),
                        acc.contiguous(), pos.contiguous(), quat.contiguous(), notch)
