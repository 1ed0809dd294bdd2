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
from .schedule import RunInputs, build_run_inputs, initial_state, noise_matrices

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


# ----------------------------------------------------------------------------------------------
# BASELINE config #5: 1 kHz IMU / 30 Hz vision, 10 000 IMU steps
# ----------------------------------------------------------------------------------------------
@dataclass
class LongHorizonWorkload:
    """One shared camera path / IMU stream, B filter instances that differ in their initial extrinsics.
    Host arrays (NumPy, float64); ``device_inputs`` moves them to the GPU in the filter's dtype."""
    cfg: object
    Q: np.ndarray
    R: np.ndarray
    sigma_a: np.ndarray
    sigma_om: np.ndarray
    x0: np.ndarray        # [B,26]
    P0: np.ndarray        # [24,24]
    om0: np.ndarray       # [3]  IMU sample at t = 0 (Filter.py:63)
    acc0: np.ndarray      # [3]
    n_pred: np.ndarray    # [E]  33/33/34 ...: IMU samples in (t_{k-1}, t_k]
    dt: np.ndarray        # [N]
    om: np.ndarray        # [N,3]   body rate   (f_imu_meas, equations.py:15-41,63-69)
    acc: np.ndarray       # [N,3]   body acceleration
    t_imu: np.ndarray     # [N]
    pos: np.ndarray       # [E,3]
    quat: np.ndarray      # [E,4]  raw measurement quaternions (xyzw)
    notch: np.ndarray     # [E]
    t_meas: np.ndarray    # [E]
    cam_inputs: dict      # what dvi_eskf_imu_synthesize consumes: R_WC [N+1,3,3], om/acc/alp [N+1,3] (sample 0 = t0)
    dofs_gt: np.ndarray   # [6]
    hold_frozen: bool = True
    # what was changed against the default config.yaml (None = the YAML's value): reported next to every number measured
    # on this workload -- with the YAML's camera noise the reference filter stays bounded for ~800 steps only
    meas_pos_std: object = None
    meas_theta_deg: object = None

    @property
    def n_steps(self) -> int:
        return int(self.dt.shape[0])

    def make_filter(self, dtype=torch.float64, device="cuda", batch=None, first=0):
        from .batch import BatchedESKF
        B = int(batch if batch is not None else self.x0.shape[0])
        f = BatchedESKF(B, length=self.cfg.model.length, theta_cam=self.cfg.model.angle, frozen=self.cfg.frozen_dofs,
                        sigma_a=self.sigma_a, sigma_om=self.sigma_om, hold_frozen=self.hold_frozen, dtype=dtype,
                        device=torch.device(device))
        f.set_noise(self.Q, self.R)
        f.set_state(x=self.x0[first:first + B], P=self.P0, om_old=self.om0, acc_old=self.acc0)
        return f

    def synthesize_imu_on_device(self, dtype=torch.float64, device="cuda"):
        """The IMU stream from the camera samples by the HIP library; sample 0 is the
        buffer initialisation (om0, acc0), samples 1.. drive the predicts.  Returns (om [N+1,3], acc [N+1,3])."""
        from .imu import synthesize_on_device
        c = self.cam_inputs
        return synthesize_on_device(c["R_WC"], c["om"], c["acc"], c["alp"], np.zeros(3), self.dofs_gt,
                                    self.cfg.model.length, self.cfg.model.angle, dtype=dtype, device=device)


def long_horizon(batch: int = 8192, n_steps: int = 10000, imu_hz: float = 1000.0, cam_hz: float = 30.0, seed: int = 5,
                 frozen=(1, 1, 1, 1, 1, 1), hold_frozen: bool = True, ic_sigma_scale: float = 1.0,
                 meas_pos_std=0.2, meas_theta_deg=1.0, config_path=None, **camera_kw) -> LongHorizonWorkload:
    """BASELINE config #5: the camera follows the septic smooth-step path of the
    reference's ``data/trajs/smooth_traj.py`` sampled at ``cam_hz``; its rates come from the reference's
    finite-difference recipe (Camera.py:158-170) and are interpolated linearly to the IMU clock
    (Interpolator.py:25-88); the IMU samples are ``f_imu_meas`` of those (equations.py:15-41,63-69), one per
    millisecond; camera frame k is applied after the IMU samples with t_{k-1} < t_i <= t_k
    (Camera.py:320-347) -- 33/33/34 predicts per update at 1 kHz / 30 Hz.

    Instance 0 starts at the initial condition of ``get_ic``; instance i > 0 with its IMU position and
    velocity off by N(0, cov0 block) * ic_sigma_scale and, for unfrozen DOFs, its extrinsics off by
    N(0, cov0 dofs block) * ic_sigma_scale (NumPy Philox, key ``seed``).

    Regime.  With the default ``config.yaml`` (camera attitude noise 1e-5 deg) the reference filter -- its
    quirks kept, SURVEY App. A.6 -- leaves its bounded regime within ~2 s of ANY camera motion on this
    schedule (measured with the oracle: an exponentially growing mode, |x| ~ 1e4 at 10 s, NaN for larger
    motions), also with the extrinsics frozen at their true values; only the reference's own committed
    amplitudes -- all zero, a camera at rest -- stay put.  ``meas_pos_std`` / ``meas_theta_deg`` replace
    ``camera.noise.position`` / ``.theta`` of the YAML (``None`` keeps the YAML's); the defaults (0.2, 1 deg) with
    frozen, held extrinsics keep the filter bounded over the 10 000 steps, which is what a comparison of two
    arithmetics at step 10 000 needs.  The schema and every other derived quantity are the YAML's."""
    config_path = config_path or os.path.join(REPO_ROOT, "config.yaml")
    cfg = Config(config_path, base_dir=os.path.dirname(os.path.abspath(config_path)))
    cfg.frozen_dofs = list(frozen)
    if meas_pos_std is not None:
        cfg.camera.noise.position = [float(meas_pos_std)] * 3
    if meas_theta_deg is not None:
        cfg.camera.noise.theta = [float(np.radians(meas_theta_deg))] * 3
    cfg.meas_noise_std = cfg.camera.noise.vec
    cfg.meas_noise_var = np.square(cfg.meas_noise_std)
    per_frame = imu_hz / cam_hz
    cfg.interframe_vals = per_frame                          # random-walk std per IMU step (config.py:258-259)
    cfg.process_noise_rw_std = cfg.filter.noise_dofs.vec / per_frame
    cfg.process_noise_rw_var = np.square(cfg.process_noise_rw_std)
    probe = SimpleProbe(cfg.model.length, cfg.model.angle)
    cfg.update_dofs(probe)

    from .smooth_traj import smooth_camera
    camera_kw.setdefault("shape", (0.0, 1.0, 1.0))           # the reference's own profile (smooth_traj.py:69,72-74)
    camera_kw.setdefault("partitions", (0.0, 0.17, 1.0))
    n_frames = int(np.ceil(n_steps / per_frame - 1e-9)) + 1
    cam = smooth_camera(num_frames=n_frames, rate_hz=cam_hz, **camera_kw)
    t_imu_all = np.arange(n_steps + 1) / float(imu_hz)
    t_imu_all = t_imu_all[t_imu_all <= cam.t[-1] + 1e-12]
    cam_i = cam.interpolate_to(t_imu_all)
    imu = Imu(probe, cam_i, cfg.imu.stdev_accel, cfg.imu.stdev_omega)

    gt = [np.array(v, dtype=float) for v in probe.joint_dofs]
    n_all = len(t_imu_all)
    om_all = np.empty((n_all, 3))
    acc_all = np.empty((n_all, 3))
    for i in range(n_all):
        om_all[i], acc_all[i] = imu.eval_expr_single(cam_i.t[i], gt[0], gt[1], gt[2], cam_i.acc[:, i], cam_i.R[i],
                                                     cam_i.om[:, i], cam_i.alp[:, i])
    # schedule: frame k consumes the samples with t_{k-1} < t_i <= t_k (a tolerance of a nanosecond guards
    # the frame times that are multiples of the IMU period against their rounding)
    idx_of_frame = np.searchsorted(t_imu_all, cam.t + 1e-9, side="right") - 1
    n_pred = np.diff(idx_of_frame)[: n_frames - 1]
    n_use = int(idx_of_frame[-1])
    E = len(n_pred)
    x0 = initial_state(cam, imu, cfg.ic_imu_dofs)
    B = int(batch)
    X0 = np.tile(x0, (B, 1))
    P0 = np.array(cfg.cov0_matrix, dtype=float)
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    z = rng.standard_normal((B, 6))
    z[0] = 0.0
    free = 1.0 - np.asarray(cfg.frozen_dofs, dtype=float)
    X0[:, 10:16] += z * np.sqrt(np.diag(P0)[9:15]) * free * float(ic_sigma_scale)
    zpv = rng.standard_normal((B, 6))
    zpv[0] = 0.0
    X0[:, 0:6] += zpv * np.sqrt(np.diag(P0)[0:6]) * float(ic_sigma_scale)
    Q, R = noise_matrices(cfg, imu.stdev_na, imu.stdev_nom)
    meas = [cam.traj.at_index(k) for k in range(1, E + 1)]
    cam_inputs = {"R_WC": np.array(cam_i.R[: n_use + 1]), "om": cam_i.om[:, : n_use + 1].T.copy(),
                  "acc": cam_i.acc[:, : n_use + 1].T.copy(), "alp": cam_i.alp[:, : n_use + 1].T.copy()}
    return LongHorizonWorkload(
        cfg=cfg, Q=Q, R=R, sigma_a=np.array(imu.stdev_na), sigma_om=np.array(imu.stdev_nom), x0=X0, P0=P0,
        om0=om_all[0].copy(), acc0=acc_all[0].copy(), n_pred=n_pred.astype(np.int32), dt=np.diff(t_imu_all[: n_use + 1]),
        om=om_all[1: n_use + 1].copy(), acc=acc_all[1: n_use + 1].copy(), t_imu=t_imu_all[1: n_use + 1].copy(),
        pos=np.array([m.pos for m in meas]), quat=np.array([m.q.xyzw for m in meas]), notch=np.zeros(E),
        t_meas=np.asarray(cam.t[1: E + 1], dtype=float), cam_inputs=cam_inputs, dofs_gt=np.array(probe.imu_dofs, dtype=float),
        hold_frozen=bool(hold_frozen), meas_pos_std=meas_pos_std, meas_theta_deg=meas_theta_deg)
