"""``Filter`` -- the reference's ESKF class surface (filter/Filter.py:28-476) on top of the
HIP kernels.  Every per-step computation (nominal propagation, Jacobians, covariance
propagation, gain, injection, covariance update) runs on the GPU through the C ABI
(``BatchedESKF``); this class keeps the reference's attributes, call protocol and
trajectory log so that ``Simulator`` / ``main.py`` work unchanged.

With ``config.batch > 1`` the same call sequence drives a whole batch of instances
(identical inputs unless per-instance arrays are passed to ``propagate``/``update``);
scalar-shaped attributes (``_states``, ``_P`` ...) then refer to instance 0.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import rotmath as rm
from .batch import BatchedESKF
from .state import ErrorState, State
from .traj import FilterTraj, save_trajectory


class Filter:
    """Error-State Kalman Filter fusing camera and IMU measurements through the probe model."""

    def __init__(self, sim):
        assert sim.config.dofs_updated
        self.run_id: Optional[int] = None
        self._config = sim.config
        self._dt = 0.0
        self.show_progress = True
        self.num_meas = 7
        self.num_noise = 13
        cov0 = np.asarray(sim.cov0, dtype=float)
        assert cov0.shape == (ErrorState.SIZE, ErrorState.SIZE)
        self._frozen_dofs = [bool(fr) for fr in self._config.frozen_dofs]
        self._probe = sim.sym_probe
        self.imu = sim.imu
        self._x0: State = sim.x0

        # imu (Filter.py:63-65)
        self.imu.eval_init(self._config.gt_joint_dofs, self._x0.notch_dofs)
        self.stdev_na = np.array(self.imu.stdev_na)
        self.stdev_nom = np.array(self.imu.stdev_nom)

        self.batch = int(getattr(self._config, "batch", 1))
        self._dev = BatchedESKF(
            self.batch, length=self._probe.scope_length, theta_cam=self._probe.theta_cam,
            frozen=self._frozen_dofs, sigma_a=self.stdev_na, sigma_om=self.stdev_nom,
            hold_frozen=bool(getattr(self._config, "golden_compat", False)))
        self.R = np.diag(self._config.meas_noise_var)
        self.update_noise_matrices()

        # state, covariance and buffers live on the device (Filter.py:44-45,78-80)
        # (the sample eval_init has just appended, i.e. the last column: identical to the reference's
        #  ``imu.om.squeeze()`` (Filter.py:78-79) for a first construction; for a re-construction by
        #  Simulator.reset_kf (Simulator.py:160-161) the reference's squeeze yields a 3x2 buffer that fails at the
        #  next propagate -- here the tuning loop can re-create the filter)
        self._dev.set_state(x=self._x0.vector, P=cov0, om_old=np.asarray(self.imu.om).reshape(3, -1)[:, -1],
                            acc_old=np.asarray(self.imu.acc).reshape(3, -1)[:, -1])

        self.H = np.zeros([self.num_meas, ErrorState.SIZE])
        self.H[0:6, 18:24] = np.eye(6)
        self.H[6, 15] = 1
        # the kernels form the transition / noise Jacobians block-wise and never materialise them; the
        # reference's Fx / Fi attributes (Filter.py:259,265) are expanded on request from the blocks the
        # last predict left in the workspace (dvi_eskf_get_jacobians): ``propagate`` asks the device to keep
        # them, ``run`` -- which enqueues whole runs -- only for the last frame's predicts
        self._keeps_jacobians = False
        # ``run`` enqueues all frames but the last as ONE device run (dvi_eskf_run_epochs) and steps through the
        # last frame with the reference's own call sequence; False: every frame stepwise (one launch per call)
        self.fast_run = True
        self._log_dev: Optional[BatchedESKF] = None      # batch > 1: the filter that logs instance 0's trajectory

        self.traj = FilterTraj("kf")
        self.traj.append_row(self._log_row(self._config.min_t))
        self.mse = 0
        self.update_mse = 0
        self.status = np.zeros(self.batch, dtype=np.int32)

    # -- device-backed attributes -----------------------------------------------------------
    @property
    def device_filter(self) -> BatchedESKF:
        return self._dev

    def _log_row(self, t: float) -> np.ndarray:
        """The 30-column FilterTraj row of instance 0 (FilterTraj.py:12-41), formed on the device."""
        return self._dev.log_rows(t)[0].cpu().numpy()

    @property
    def Fx(self) -> Optional[np.ndarray]:
        """Error-state transition of the last ``propagate`` (24x24, instance 0; Filter.py:249-259,279-285);
        ``None`` before the first one."""
        return self._jacobians()[0]

    @property
    def Fi(self) -> Optional[np.ndarray]:
        """Noise Jacobian of the last ``propagate`` (24x13, instance 0; Filter.py:261-265,334-342)."""
        return self._jacobians()[1]

    def _jacobians(self):
        if not getattr(self, "_propagated", False):
            return None, None
        Fx, Fi = self._dev.jacobians()
        return Fx[0].cpu().numpy(), Fi[0].cpu().numpy()

    @property
    def _states(self) -> State:
        return State.from_vector(self._dev.x[0].cpu().numpy())

    @property
    def _P(self) -> np.ndarray:
        return self._dev.P[0].cpu().numpy()

    @property
    def states_batch(self) -> np.ndarray:
        return self._dev.x.cpu().numpy()

    @property
    def om_old(self) -> np.ndarray:
        return self._dev.get_state(("om_old",))["om_old"][0].cpu().numpy()

    @om_old.setter
    def om_old(self, val) -> None:
        self._dev.set_state(om_old=np.asarray(val, dtype=float).squeeze())

    @property
    def acc_old(self) -> np.ndarray:
        return self._dev.get_state(("acc_old",))["acc_old"][0].cpu().numpy()

    @acc_old.setter
    def acc_old(self, val) -> None:
        self._dev.set_state(acc_old=np.asarray(val, dtype=float).squeeze())

    @property
    def R_WB_old(self) -> np.ndarray:
        return rm.quat_to_matrix(self._dev.get_state(("q_old",))["q_old"][0].cpu().numpy())

    @R_WB_old.setter
    def R_WB_old(self, val) -> None:
        self._dev.set_state(q_old=rm.quat_normalise(rm.matrix_to_quat(val)))

    @property
    def u(self):
        return [self.om_old, self.acc_old]

    @property
    def Om_old(self):
        from .quaternion import Quaternion
        return Quaternion(w=1.0, v=(0.5 * self._dt * self.om_old), do_normalise=True)

    def __deepcopy__(self, memo):
        """copy.deepcopy(kf) (Simulator.py:150): the device buffers are cloned."""
        import copy
        new = object.__new__(Filter)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_dev":
                new._dev = self._dev.clone()
            elif k == "_log_dev":
                new._log_dev = None
            elif k in ("imu", "_probe", "_config"):
                setattr(new, k, v)
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new

    # -- configuration ----------------------------------------------------------------------
    def reset(self, x0, cov0, notch0) -> None:
        """Filter.reset (Filter.py:95-108)."""
        self._dt = 0.0
        self.imu.reset()
        self.imu.eval_init(self._config.gt_joint_dofs, notch0)
        self._dev.set_state(x=x0.vector, P=np.asarray(cov0, dtype=float), om_old=self.imu.om.squeeze(),
                            acc_old=self.imu.acc.squeeze())
        self.traj.reset()
        self.traj.append_row(self._log_row(self._config.min_t))
        self.mse = 0

    def update_noise_matrices(self) -> None:
        """Filter.update_noise_matrices (Filter.py:110-117) -- also the construction of Q in
        __init__ (Filter.py:67-72), where ``_dt`` is still 0."""
        Q = np.eye(self.num_noise)
        Q[0:3, 0:3] = self._dt ** 2 * self.stdev_na ** 2 * np.eye(3)
        Q[3:6, 3:6] = self._dt ** 2 * self.stdev_nom ** 2 * np.eye(3)
        Q[6:13, 6:13] = np.diag(self._config.process_noise_rw_var)
        self.Q = Q
        mv = np.asarray(self._config.meas_noise_var, dtype=float)
        if mv.shape == (7,):
            self.R = np.diag(mv)
        self._dev.set_noise(self.Q, self.R)

    # -- main loop --------------------------------------------------------------------------
    def run(self, camera, k: int, run_desc_str: str) -> None:
        """Filter.run (Filter.py:144-168): loop over camera frames 1..

        All frames but the last are enqueued as ONE device run (``dvi_eskf_run_epochs``: the time loop and the predicts
        of every frame inside the kernel); the trajectory log -- a row per IMU sample, the row of a frame's last sample
        replaced by the corrected state -- comes back from the same run and ``traj`` is built once.  The last frame is
        stepped through ``run_one_epoch`` as the reference does, so that ``Fx`` / ``Fi``, the gain returned by ``update``
        and ``update_mse`` are those of the run's last predict / update."""
        old_t = self._config.min_t
        self.run_id = k
        frames = list(enumerate(camera.t[1:], start=1))
        if self.fast_run and len(frames) > 1:
            old_t = self._run_frames_on_device(old_t, frames[:-1], camera)
            frames = frames[-1:]
        for i_cam, t in frames:
            self.run_one_epoch(old_t, t, i_cam, camera)
            self.calculate_update_mse(i_cam, camera)
            old_t = t

    def _run_frames_on_device(self, old_t: float, frames, camera) -> float:
        """``run_one_epoch`` for every frame of ``frames`` as one device run; returns the time of the last one."""
        # host-side input preparation, sample by sample as propagate_imu does it (Filter.py:187-217): the synthetic IMU
        # stream, the IMU reference trajectory, the camera measurements
        import time
        t_host0 = time.perf_counter()
        real_probe_dofs = self._config.gt_joint_dofs
        n_pred, t_pred, dts, oms, accs, pos, quat, notch = [], [], [], [], [], [], [], []
        for i_cam, t in frames:
            old_ti, n = old_t, 0
            for cam_meas in self.imu.cam.samples_between(old_t, t):
                real_probe_dofs[0][6] = cam_meas.notch
                real_probe_dofs[1][6] = cam_meas.notch_d
                real_probe_dofs[2][6] = cam_meas.notch_dd
                om, acc = self.imu.eval_expr_single(cam_meas.t, *real_probe_dofs, cam_meas.acc, cam_meas.R, cam_meas.om,
                                                    cam_meas.alp, overwrite=True)
                if self.imu.ref is not None:
                    self.imu.ref.append_value(cam_meas.t, cam_meas.vec, cam_meas.notch_arr)
                self._dt = cam_meas.t - old_ti
                dts.append(self._dt)
                oms.append(np.asarray(om, dtype=float).reshape(3))
                accs.append(np.asarray(acc, dtype=float).reshape(3))
                t_pred.append(cam_meas.t)
                old_ti = cam_meas.t
                n += 1
            n_pred.append(n)
            cam = camera.traj.at_index(i_cam)
            pos.append(np.asarray(cam.pos, dtype=float).reshape(3))
            quat.append(np.asarray(cam.q.xyzw, dtype=float))
            notch.append(float(camera.get_notch_vec_at(i_cam)[0]))
            old_t = t
        total = int(sum(n_pred))
        args = (np.asarray(n_pred, dtype=np.int32), np.asarray(dts, dtype=float).reshape(total), np.asarray(oms, dtype=float).reshape(total, 3),
                np.asarray(accs, dtype=float).reshape(total, 3), np.asarray(pos), np.asarray(quat), np.asarray(notch))
        if self._keeps_jacobians:
            self._dev.keep_jacobians(False)
            self._keeps_jacobians = False
        rows = None
        torch.cuda.synchronize(self._dev.device)
        t_dev0 = time.perf_counter()
        if total > 0:
            if self.batch == 1:
                status, rows = self._dev.run_epochs(*args, t_pred=t_pred, rows_instances=1)
            else:
                # the trajectory log is instance 0's: a one-instance filter started from instance 0's state logs it (same
                # inputs, same arithmetic), the batch itself runs without a log
                logf = self._log_filter()
                _, rows = logf.run_epochs(*args, t_pred=t_pred, rows_instances=1)
                status, _ = self._dev.run_epochs(*args)
        else:
            status, _ = self._dev.run_epochs(*args)
        st0 = status[:, 0].cpu().numpy()                     # (synchronises: the run is through)
        # where the wall time of this call went: the reference's sample-by-sample input synthesis on the host against the
        # filter run itself on the device
        self.last_run_timing = {"host_input_preparation_s": t_dev0 - t_host0, "device_run_s": time.perf_counter() - t_dev0,
                                "frames": len(frames), "imu_samples": total}
        for _ in range(int((st0 & 1).sum())):               # Filter.py:356-361: a message, nothing raised, the run goes on
            print("ERROR: Singular matrix!")
            print("Stopping simulation.")
        self.status = status[-1].cpu().numpy()
        self._propagated = False                            # (Fx / Fi: those of the next ``propagate``)
        if rows is not None:
            for row in rows[:, 0].cpu().numpy():
                self.traj.append_row(row)
        return old_t

    def _log_filter(self) -> BatchedESKF:
        """batch > 1: a one-instance filter holding instance 0's state (and the batch's noise model)."""
        if self._log_dev is None:
            self._log_dev = BatchedESKF(1, length=self._probe.scope_length, theta_cam=self._probe.theta_cam, frozen=self._frozen_dofs,
                                        sigma_a=self.stdev_na, sigma_om=self.stdev_nom,
                                        hold_frozen=bool(getattr(self._config, "golden_compat", False)), device=self._dev.device)
        self._log_dev.set_noise(self.Q, self.R)
        self._log_dev.copy_instances_from(self._dev, 0)
        return self._log_dev

    def run_one_epoch(self, old_t: float, t: float, i_cam: int, camera) -> None:
        """>= 1 predictions then one update (Filter.py:170-185)."""
        self.propagate_imu(old_t, t)
        cam_meas = camera.traj.at_index(i_cam)
        ang_notch = camera.get_notch_vec_at(i_cam)[0]
        self.update(t, cam_meas, ang_notch)

    def propagate_imu(self, t0: float, tn: float) -> None:
        """Filter.propagate_imu (Filter.py:187-217): synthesises the IMU samples between two
        camera frames (host, one-time preparation) and propagates once per sample."""
        real_probe_dofs = self._config.gt_joint_dofs
        old_ti = t0
        for cam_meas in self.imu.cam.samples_between(t0, tn):
            real_probe_dofs[0][6] = cam_meas.notch
            real_probe_dofs[1][6] = cam_meas.notch_d
            real_probe_dofs[2][6] = cam_meas.notch_dd
            om, acc = self.imu.eval_expr_single(cam_meas.t, *real_probe_dofs, cam_meas.acc, cam_meas.R, cam_meas.om,
                                                cam_meas.alp, overwrite=True)
            if self.imu.ref is not None:
                self.imu.ref.append_value(cam_meas.t, cam_meas.vec, cam_meas.notch_arr)
            self._dt = cam_meas.t - old_ti
            self.propagate(cam_meas.t, om, acc)
            old_ti = cam_meas.t

    def propagate(self, t, om, acc) -> None:
        """Filter.propagate (Filter.py:219-230) -- one kernel launch."""
        if not self._keeps_jacobians:
            self._dev.keep_jacobians(True)
            self._keeps_jacobians = True
        self._dev.predict(self._dt, np.asarray(om, dtype=float), np.asarray(acc, dtype=float))
        self._propagated = True
        self.traj.append_row(self._log_row(t))

    def update(self, t: float, camera, ang_notch: float):
        """Filter.update (Filter.py:351-395).  Returns K (24x7, instance 0) or ``None`` when
        the innovation covariance could not be factorised (status bit 0) -- like the
        reference, a message is printed and nothing is raised."""
        status, K = self._dev.update(np.asarray(camera.pos, dtype=float).reshape(3),
                                     np.asarray(camera.q.xyzw, dtype=float), float(ang_notch), want_gain=True)
        self.status = status.cpu().numpy()
        if self.status[0] & 1:
            print("ERROR: Singular matrix!")
            print("Stopping simulation.")
            return None
        self.traj.overwrite_last_row(self._log_row(t))    # FilterTraj.py:66-69
        return K[0].cpu().numpy()

    # -- metrics (host side, Filter.py:397-455) -------------------------------------------------
    def calculate_update_mse(self, i_cam, camera) -> None:
        """Filter.calculate_update_mse (Filter.py:397-418), reduced on the device from the state record
        (``dvi_eskf_update_mse``: every instance at once; ``update_mse`` is instance 0's, ``update_mse_batch``
        the device vector)."""
        cam_reference = camera.rotated if camera.rotated else camera
        tr = cam_reference.traj
        cam6 = [float(np.asarray(getattr(tr, c))[i_cam]) for c in ("x", "y", "z", "rx_deg", "ry_deg", "rz_deg")]
        imu6 = None
        if self.imu.ref is not None and self.imu.ref.num_values:
            imu6 = [float(getattr(self.imu.ref, c)[-1]) for c in self.traj.labels_imu[3:9]]
        self.update_mse_batch = self._dev.update_mse(cam6, imu6)
        self.update_mse = float(self.update_mse_batch[0])

    def calculate_metric(self, i_cam, camera) -> float:
        """Filter.calculate_metric (Filter.py:420-450; unused by v1's run loop, Filter.py:167-168): the
        squared pose errors of the rows at the update indices, averaged over 12 states and the frames,
        plus the DOF metric.  Host side, from the trajectory log."""
        cam_reference = camera.rotated if camera.rotated else camera
        n_if = int(self._config.interframe_vals)
        upd = list(range(0, len(self.traj.t), n_if))
        sum_cam = 0.0
        for compc in self.traj.labels_camera[0:6]:
            ref = np.asarray(getattr(cam_reference.traj, compc[:-1]))[: i_cam + 1]
            sum_cam += float(np.sum(np.square(ref - np.asarray(getattr(self.traj, compc))[upd])))
        sum_imu = 0.0
        if self.imu.ref is not None and self.imu.ref.num_values:
            for comp in self.traj.labels_imu[3:9]:
                a = np.asarray(getattr(self.traj, comp))[upd[:-1]]
                r = np.asarray(getattr(self.imu.ref, comp))[upd[:-1]]
                sum_imu += float(np.sum(np.square(a - r)))
        return (sum_cam + sum_imu) / (12 * len(upd)) + self.calculate_dof_metric()

    def calculate_dof_metric(self) -> float:
        res = self._states.dofs - np.asarray(self._config.gt_imu_dofs, dtype=float)
        return float(np.dot(res, res) / 6)

    def calculate_dof_metric_batch(self) -> torch.Tensor:
        """Per-instance DOF MSE [B], computed on the device."""
        gt = torch.as_tensor(np.asarray(self._config.gt_imu_dofs, dtype=float), device=self._dev.device)
        d = self._dev.calibration() - gt
        return (d * d).sum(dim=1) / 6

    def save(self) -> None:
        """Filter.save (Filter.py:457-465)."""
        self._config.mse = self.mse
        kf_filepath = os.path.join(str(self._config.traj_path), f"kf_best_{self._config.traj_name}.txt")
        save_trajectory(self.traj, kf_filepath)
        if self.imu.ref is not None:
            save_trajectory(self.imu.ref, kf_filepath.replace("kf_best", "imu_ref"))

    def plot(self, camera, compact) -> None:
        if not self._config.do_plot:
            return
        print("plotting is outside this build (visuals/* of the reference is out of scope); "
              "use Filter.traj / Filter.save() to export the trajectory")
