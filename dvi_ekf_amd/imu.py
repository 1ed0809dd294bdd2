"""Synthetic, noise-free IMU derived from the camera trajectory through the probe
kinematics (host side, NumPy; one sample per call).

Formulas: kinematics/equations.py:8-41 (f_imu, f_imu_meas) evaluated in closed
form; call protocol of models/Imu.py (87-97, 141-226, 305-315) kept so the
reference-shaped ``Filter`` can drive it.  SURVEY.md App. A.1 steps 4-6.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import rotmath as rm
from .probe import SimpleProbe
from .quaternion import Quaternion


def imu_pose_from_camera(cam_vec, fwkin):
    """f_imu (equations.py:8-13,54-60): IMU position, attitude, velocity in W."""
    W_p_CW, R_WC, WW_v_CW, W_om_CW, _, _ = (np.asarray(a, dtype=float) for a in cam_vec)
    B_p_CB, R_BC, BB_v_CB, B_om_CB, _, _ = fwkin
    R_WB = R_WC @ R_BC.T
    W_om_BW = W_om_CW.reshape(3) - R_WB @ B_om_CB
    lever = R_WB @ B_p_CB
    W_p_BW = W_p_CW.reshape(3) - lever
    WW_v_BW = WW_v_CW.reshape(3) - R_WB @ BB_v_CB - np.cross(W_om_BW, lever)
    return W_p_BW, R_WB, WW_v_BW


def imu_measurement(R_WC, W_om_CW, W_acc_CW, W_alp_CW, fwkin):
    """f_imu_meas (equations.py:15-41,63-69): body-frame angular rate and
    acceleration; no gravity, bias or noise."""
    B_p_CB, R_BC, BB_v_CB, B_om_CB, B_acc_CB, B_alp_CB = fwkin
    R_WB = np.asarray(R_WC, dtype=float) @ R_BC.T
    om_c = np.asarray(W_om_CW, dtype=float).reshape(3)
    W_om = om_c - R_WB @ B_om_CB
    lever = R_WB @ B_p_CB
    om_x_lever = np.cross(W_om, lever)
    W_alp = np.asarray(W_alp_CW, dtype=float).reshape(3) - R_WB @ B_alp_CB - np.cross(W_om, R_WB @ B_om_CB)
    W_acc = (np.asarray(W_acc_CW, dtype=float).reshape(3) - R_WB @ B_acc_CB
             - 2 * np.cross(W_om, R_WB @ BB_v_CB) - np.cross(W_alp, lever) - np.cross(W_om, om_x_lever))
    return R_WB.T @ W_om, R_WB.T @ W_acc


def synthesize_on_device(R_WC, om_CW, acc_CW, alp_CW, notch, dofs6, length, theta_cam, p_CW=None, v_CW=None,
                         dtype=None, device=None):
    """The whole IMU stream of a run in one launch of the HIP library (``dvi_eskf_imu_synthesize``,
    SURVEY 8(f1)): what ``Imu.eval_expr_single`` (Imu.py:141-196) evaluates per sample on the host.

    R_WC [n,3,3], om_CW/acc_CW/alp_CW [n,3], notch [n,3] or [3] = (q7, q7', q7''), dofs6 = the six
    ground-truth joint values.  Returns device tensors ``om_B, acc_B`` [n,3] and, when ``p_CW`` and
    ``v_CW`` are given, also the reference pose ``p_BW [n,3], R_WB [n,3,3], v_BW [n,3]``
    (``Imu.ref_vals``, Imu.py:305-315).  No CPU fallback: raises if the library is missing."""
    import ctypes as C

    import torch

    from . import _capi
    lib = _capi.lib()
    dtype = dtype or torch.float64
    device = torch.device(device if device is not None else "cuda:0")

    def dev(a, shape):
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(device=device, dtype=dtype)
        return t.reshape(shape).contiguous()

    n = int(np.asarray(om_CW).reshape(-1, 3).shape[0])
    tR, tom, tacc, talp = dev(R_WC, (n, 9)), dev(om_CW, (n, 3)), dev(acc_CW, (n, 3)), dev(alp_CW, (n, 3))
    notch = np.asarray(notch, dtype=np.float64)
    per_sample = notch.ndim == 2
    tn = dev(notch, (n, 3) if per_sample else (3,))
    want_ref = p_CW is not None and v_CW is not None
    tp = dev(p_CW, (n, 3)) if want_ref else None
    tv = dev(v_CW, (n, 3)) if want_ref else None
    out_om = torch.empty((n, 3), dtype=dtype, device=device)
    out_acc = torch.empty((n, 3), dtype=dtype, device=device)
    out_p = torch.empty((n, 3), dtype=dtype, device=device) if want_ref else None
    out_R = torch.empty((n, 9), dtype=dtype, device=device) if want_ref else None
    out_v = torch.empty((n, 3), dtype=dtype, device=device) if want_ref else None
    prm = _capi.Params()
    prm.L, prm.theta_cam = float(length), float(theta_cam)
    d6 = (C.c_double * 6)(*[float(v) for v in dofs6])
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _capi.check(lib.dvi_eskf_imu_synthesize(C.byref(prm), 0 if dtype == torch.float64 else 1, device.index or 0, n,
                                                ptr(tR), ptr(tom), ptr(tacc), ptr(talp), ptr(tn), int(per_sample), d6,
                                                ptr(tp), ptr(tv), ptr(out_om), ptr(out_acc), ptr(out_p), ptr(out_R),
                                                ptr(out_v), stream), "dvi_eskf_imu_synthesize")
    if want_ref:
        return out_om, out_acc, out_p, out_R.reshape(n, 3, 3), out_v
    return out_om, out_acc


class ImuRefTraj:
    """Desired IMU trajectory log, 14 columns (trajectory/ImuRefTraj.py:18-55)."""

    labels = ["t", "x", "y", "z", "vx", "vy", "vz", "rx", "ry", "rz", "qw", "qx", "qy", "qz"]

    def __init__(self, name: str, imu: "Imu"):
        self.name = name
        self._imu = imu
        self.reset()

    def reset(self) -> None:
        for lab in self.labels:
            setattr(self, lab, None)

    @property
    def num_values(self) -> int:
        return 0 if self.t is None else len(self.t)

    def append_value(self, t: float, cam_meas_vec, current_notch) -> None:
        p, R_WB, v = self._imu.ref_vals(cam_meas_vec, current_notch)
        q = Quaternion(val=R_WB, do_normalise=True)
        # scipy from_matrix(...).as_euler on the (orthonormal) product R_WC R_BC^T
        data = [t, *p, *v, *q.euler_xyz_deg, *q.wxyz]
        for i, lab in enumerate(self.labels):
            if getattr(self, lab) is None:
                setattr(self, lab, [data[i]])
            else:
                getattr(self, lab).append(data[i])


class Imu:
    def __init__(self, probe: SimpleProbe, cam, stdev_na=None, stdev_nom=None, gen_ref: bool = False):
        self.stdev_na = stdev_na
        self.stdev_nom = stdev_nom
        self.probe = probe
        self.cam = cam            # interpolated camera
        self.t: List[float] = []
        self._om: List[np.ndarray] = []
        self._acc: List[np.ndarray] = []
        self.ref: Optional[ImuRefTraj] = ImuRefTraj("imu ref", self) if gen_ref else None

    @staticmethod
    def create(config, camera, probe, gen_ref: bool = False) -> "Imu":
        """Imu.create (Imu.py:87-97): ``config`` is the ImuConfig in the reference;
        any object with interframe_vals/stdev_accel/stdev_omega works."""
        cam_reference = camera.rotated if camera.rotated else camera   # Imu.py:89: the IMU rides the rotated camera
        return Imu(probe, cam_reference.interpolate(config.interframe_vals), config.stdev_accel,
                   config.stdev_omega, gen_ref=gen_ref)

    def reset(self) -> None:
        self.t, self._om, self._acc = [], [], []
        if self.ref is not None:
            self.ref.reset()

    @property
    def om(self) -> np.ndarray:
        return np.array(self._om).T

    @property
    def acc(self) -> np.ndarray:
        return np.array(self._acc).T

    def eval_expr_single(self, t, q, qd, qdd, W_acc_C, R_WC, W_om_C, W_alp_C,
                         append_array: bool = False, filepath: str = "", overwrite: bool = False):
        """Imu.eval_expr_single (Imu.py:141-196): q/qd/qdd are the 8 joint values,
        of which only the notch entries [6] vary."""
        from .probe import probe_fwkin
        fk = probe_fwkin(q[:6], (q[6], qd[6], qdd[6]), self.probe.scope_length, self.probe.theta_cam)
        res_om, res_acc = imu_measurement(R_WC, W_om_C, W_acc_C, W_alp_C, fk)
        if append_array:
            self.t.append(t)
            self._om.append(res_om)
            self._acc.append(res_acc)
        if overwrite:
            self.t, self._om, self._acc = [t], [res_om], [res_acc]
        return res_om, res_acc

    def eval_init(self, probe_joint_dofs, notch_dofs, overwrite: bool = False) -> None:
        """Imu.eval_init (Imu.py:198-226): IMU sample at interpolated index 0."""
        q0, qd0, qdd0 = probe_joint_dofs
        q0[6], qd0[6], qdd0[6] = notch_dofs
        self.eval_expr_single(self.cam.t[0], q0, qd0, qdd0, self.cam.acc[:, 0], self.cam.R[0],
                              self.cam.om[:, 0], self.cam.alp[:, 0], append_array=True, overwrite=overwrite)

    def ref_vals(self, cam_meas_vec, current_notch):
        """Imu.ref_vals (Imu.py:305-315): W_p_BW, R_WB, WW_v_BW."""
        return list(imu_pose_from_camera(cam_meas_vec, self.probe.fwkin(current_notch)))
