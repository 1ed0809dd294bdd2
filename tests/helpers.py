"""Shared test scaffolding: builds the reference's default run (mandala0_mono,
default config.yaml) from the committed fixtures under tests/golden/."""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np

from dvi_ekf_amd import rotmath as rm
from dvi_ekf_amd.camera import create_camera
from dvi_ekf_amd.config import Config
from dvi_ekf_amd.imu import Imu
from dvi_ekf_amd.probe import SimpleProbe
from dvi_ekf_amd.schedule import build_run_inputs, noise_matrices

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KF_BEST = os.path.join(GOLDEN, "data", "trajs", "kf_best_mandala0_mono.txt")
IMU_REF = os.path.join(GOLDEN, "data", "trajs", "imu_ref_mandala0_mono.txt")


def default_run(golden_compat: bool = True, fast: bool = True, frozen=None, interframe=None, with_notch: bool = False,
                max_vals=None):
    """Returns (cfg, run_inputs, Q, R, sigma_a, sigma_om).

    golden_compat=True  -> plumbing G1/G2/G3 of SURVEY.md 8(c) (what the saved file needs)
    fast=False          -> all 140 frames instead of the fast-sim 10
    with_notch=True     -> camera.with_notch: the notch90 trajectory turns the camera"""
    cfg = Config(os.path.join(GOLDEN, "config.yaml"), base_dir=GOLDEN)
    cfg.golden_compat = golden_compat
    if not fast:
        cfg.sim.do_fast_sim = False
        cfg.max_vals = None
        cfg.interframe_vals = cfg.imu.interframe_vals
        cfg.process_noise_rw_std = cfg.filter.noise_dofs.vec / cfg.interframe_vals
        cfg.process_noise_rw_var = np.square(cfg.process_noise_rw_std)
    if interframe is not None:
        cfg.interframe_vals = interframe
        cfg.process_noise_rw_std = cfg.filter.noise_dofs.vec / cfg.interframe_vals
        cfg.process_noise_rw_var = np.square(cfg.process_noise_rw_std)
    if frozen is not None:
        cfg.frozen_dofs = list(frozen)
    if with_notch:
        cfg.camera.with_notch = cfg.with_notch = True
    if max_vals is not None:
        cfg.max_vals = max_vals
    probe = SimpleProbe(cfg.model.length, cfg.model.angle)
    cfg.update_dofs(probe)
    cam = create_camera(cfg)
    n_if = cfg.interframe_vals if (golden_compat or interframe is not None) else cfg.imu.interframe_vals
    imu_cfg = SimpleNamespace(interframe_vals=n_if, stdev_accel=cfg.imu.stdev_accel,
                              stdev_omega=cfg.imu.stdev_omega)
    imu = Imu.create(imu_cfg, cam, probe, gen_ref=True)
    ri = build_run_inputs(cfg, cam, imu, probe)
    Q, R = noise_matrices(cfg, imu.stdev_na, imu.stdev_nom)
    return cfg, ri, Q, R, np.array(imu.stdev_na), np.array(imu.stdev_nom)


def log_row(t, x):
    """30-column FilterTraj row (FilterTraj.py:12-41)."""
    q, qc = x[6:10], x[22:26]
    return np.array([t, *x[0:3], *x[3:6], *np.degrees(rm.euler_xyz_extrinsic(q)), q[3], q[0], q[1], q[2],
                     *np.degrees(x[10:13]), *x[13:16], *x[19:22], *np.degrees(rm.euler_xyz_extrinsic(qc)),
                     qc[3], qc[0], qc[1], qc[2]])


def load_golden(path=KF_BEST):
    return np.loadtxt(path)  # CRLF line ends are handled by loadtxt
