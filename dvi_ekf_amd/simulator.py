"""``Simulator`` -- the reference's driver surface (filter/Simulator.py:24-252): builds the
probe, camera, IMU and initial condition from a ``Config`` and runs / tunes the filter."""
from __future__ import annotations

import copy
from types import SimpleNamespace
from typing import List, Optional

import numpy as np

from .camera import create_camera
from .filter import Filter
from .imu import Imu
from .probe import SimpleProbe, SymProbe
from .schedule import initial_state
from .state import State


class Simulator:
    def __init__(self, config):
        probe = SimpleProbe(scope_length=config.model.length, theta_cam=config.model.angle)
        self.probe = probe
        self.sym_probe = SymProbe(probe)
        self._config = None
        self.camera = None
        self.imu: Optional[Imu] = None
        self.x0: Optional[State] = None
        self._cov0: Optional[np.ndarray] = None
        config.update_dofs(probe)
        self._update_config(config)
        self.kf = Filter(self)
        self._optim_std: List[float] = [*self.config.process_noise_rw_std, *self.config.meas_noise_std]
        self.mode = config.sim.mode
        self.num_kf_runs: int = config.sim.num_kf_runs
        self.show_run_progress = True
        self.mses: List[float] = []
        self.mse_best: float = 1e10
        self.mse_avg: Optional[float] = None
        self._kf_best: Optional[Filter] = None
        self._dof_mse: Optional[float] = None

    @property
    def optim_std(self) -> List[float]:
        return self._optim_std

    @optim_std.setter
    def optim_std(self, val) -> None:
        """Simulator.py:76-86 (the only place the IMU block of Q becomes non-zero)."""
        self._optim_std = val
        self.config.process_noise_rw_std = np.asarray(val[0:7], dtype=float)
        self.config.process_noise_rw_var = np.square(self.config.process_noise_rw_std)
        self.config.meas_noise_std = val[7:8]
        self.kf.update_noise_matrices()

    @property
    def config(self):
        return self._config

    @property
    def best_run_id(self) -> int:
        return self._kf_best.run_id

    @property
    def cov0(self) -> np.ndarray:
        return self._cov0.copy()

    def _update_config(self, new_config) -> None:
        """Simulator.py:101-113.  In golden-compat mode the IMU is generated with
        Config.interframe_vals (1 in fast-sim) instead of ImuConfig.interframe_vals."""
        self._config = new_config
        self.camera = create_camera(new_config)
        n_if = new_config.interframe_vals if getattr(new_config, "golden_compat", False) else \
            new_config.imu.interframe_vals
        imu_cfg = SimpleNamespace(interframe_vals=n_if, stdev_accel=new_config.imu.stdev_accel,
                                  stdev_omega=new_config.imu.stdev_omega)
        self.imu = Imu.create(imu_cfg, self.camera, self.probe, gen_ref=True)
        self.x0 = State.from_vector(initial_state(self.camera, self.imu, new_config.ic_imu_dofs))
        self._cov0 = np.array(new_config.cov0_matrix, dtype=float)

    def run_once(self) -> None:
        self.kf.run(self.camera, 0, "KF run")
        self.mse_best = self.kf.mse
        print(f"\t MSE: {self.mse_best:.2E}")

    def run(self, disp_config=False, save_best=False, verbose=True) -> None:
        """Simulator.py:121-158."""
        self.mses = []
        for k in range(self.num_kf_runs):
            run_id = k + 1
            self.kf.run(self.camera, run_id, f"KF run {run_id}/{self.num_kf_runs}")
            self.mses.append(self.kf.mse)
            if self.kf.mse < self.mse_best:
                self.mse_best = self.kf.mse
                if save_best:
                    self._kf_best = copy.deepcopy(self.kf)
            self.reset_kf()
        self.mse_avg = sum(self.mses) / len(self.mses)
        if verbose:
            print(f"\tOptimvars: {self.optim_std}")
            print(f"\tDOF MSE: {self.mse_avg:.2E}")

    def reset_kf(self) -> None:
        self.kf = Filter(self)

    def optimise(self) -> None:
        """Tuning of the 14 noise std-devs by differential evolution (Simulator.py:163-245)."""
        from scipy.optimize import differential_evolution

        def optim_func(x):
            self.optim_std = list(x)
            self.run(verbose=False)
            return self.mse_avg

        d = np.deg2rad
        bounds = ((0, 10),) * 3 + ((0, d(5)),) * 4 + ((0, 0.2),) * 3 + ((0, d(10)),) * 3 + ((0, d(1)),)
        self.show_run_progress = False
        self.kf.show_progress = False
        ret = differential_evolution(optim_func, bounds, strategy="best1bin", maxiter=1, popsize=1, disp=True,
                                     updating="immediate")
        print(f"\nglobal minimum using params: {ret.x},\nmse = {ret.fun:.3f}")
        self._dof_mse = ret.fun

    def evaluate_population(self, candidates) -> np.ndarray:
        """DOF MSE of every candidate noise setting from ONE batched run: row i of
        ``candidates`` holds the std-devs of ``optim_std`` (process random-walk std 0:7, measurement
        std 7:); candidate i becomes filter instance i with its own diag(Q) / diag(R)
        (``dvi_eskf_set_noise_per_instance``).  Same noise matrices as ``optim_func`` of ``optimise``
        (Simulator.py:163-245) builds for one candidate; the cost is ``Filter.calculate_dof_metric``
        (Filter.py:452-455) -- in v1 the assignment of ``kf.mse`` is commented out (Filter.py:168), so the
        reference's own cost is identically 0."""
        import torch

        from .batch import BatchedESKF
        from .schedule import build_run_inputs
        X = np.atleast_2d(np.asarray(candidates, dtype=float))
        S = X.shape[0]
        kf = self.kf
        ri = build_run_inputs(self.config, self.camera, self.imu, self.probe)
        q = np.tile(np.diag(kf.Q).astype(float), (S, 1))
        q[:, 0:3] = kf._dt ** 2 * kf.stdev_na ** 2                  # Filter.update_noise_matrices (Filter.py:110-117)
        q[:, 3:6] = kf._dt ** 2 * kf.stdev_nom ** 2
        q[:, 6:13] = np.square(X[:, 0:7])
        # R stays as it is: the optim_std setter (Simulator.py:76-86) stores val[7:8] as the measurement
        # std-dev but never recomputes meas_noise_var, so the tuning loop of v1 only moves Q[6:13]
        r = np.tile(np.diag(kf.R).astype(float), (S, 1))
        dev = BatchedESKF(S, length=self.sym_probe.scope_length, theta_cam=self.sym_probe.theta_cam,
                          frozen=[bool(f) for f in self.config.frozen_dofs], sigma_a=kf.stdev_na, sigma_om=kf.stdev_nom,
                          hold_frozen=bool(getattr(self.config, "golden_compat", False)))
        dev.set_noise_per_instance(q, r)
        dev.set_state(x=np.tile(ri.x0, (S, 1)), P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
        dev.run_epochs(ri.n_pred, ri.dt, ri.om, ri.acc, ri.pos, ri.quat, ri.notch)   # Filter.run, one call
        gt = torch.as_tensor(np.asarray(self.config.gt_imu_dofs, dtype=float), device=dev.device)
        d = dev.calibration() - gt
        return ((d * d).sum(dim=1) / 6).cpu().numpy()               # Filter.calculate_dof_metric (Filter.py:452-455)

    def optimise_population(self, popsize: int = 4, maxiter: int = 1, seed: int = 0):
        """``optimise`` with every generation of the differential evolution evaluated as one batch on
        the GPU (scipy's vectorised, deferred-updating mode) instead of one filter run per candidate."""
        from scipy.optimize import differential_evolution
        d = np.deg2rad
        bounds = ((0, 10),) * 3 + ((0, d(5)),) * 4 + ((0, 0.2),) * 3 + ((0, d(10)),) * 3 + ((0, d(1)),)
        ret = differential_evolution(lambda X: self.evaluate_population(np.asarray(X).T), bounds, strategy="best1bin",
                                     maxiter=maxiter, popsize=popsize, updating="deferred", vectorized=True, seed=seed,
                                     polish=False)
        print(f"\nglobal minimum using params: {ret.x},\nmse = {ret.fun:.3f}")
        self._dof_mse = ret.fun
        return ret

    def plot(self, compact=True) -> None:
        kf = self._kf_best if self._kf_best else self.kf
        kf.plot(self.camera, compact=compact)
