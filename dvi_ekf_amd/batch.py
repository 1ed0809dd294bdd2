"""``BatchedESKF`` -- a batch of independent filter instances living on one
MI355X, driven through the C ABI (include/dvi_eskf.h).  PyTorch is used only as
the owner of device buffers and the source of the HIP stream.

Per-instance semantics are those of the reference's ``Filter.propagate`` /
``Filter.update`` (Filter.py:219-395).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _capi


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedESKF:
    def __init__(self, batch: int, *, length: float, theta_cam: float, frozen: Sequence[int],
                 sigma_a, sigma_om, hold_frozen: bool = False, dtype: torch.dtype = torch.float64,
                 device: Optional[torch.device] = None, _handle=None, quirks_fixed: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedESKF needs a HIP device: the filter arithmetic only exists as HIP kernels")
        self.lib = _capi.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError("device must be a HIP (cuda:N) device")
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("dtype must be float64 or float32")
        self.dtype = dtype
        self.batch = int(batch)
        self._prm = dict(length=float(length), theta_cam=float(theta_cam), frozen=[int(bool(f)) for f in frozen],
                         sigma_a=[float(v) for v in sigma_a], sigma_om=[float(v) for v in sigma_om],
                         hold_frozen=bool(hold_frozen), quirks_fixed=int(quirks_fixed))
        self.Q = np.eye(13)
        self.R = np.eye(7)
        if _handle is not None:
            self._h = _handle
            return
        p = _capi.Params()
        p.L, p.theta_cam, p.hold_frozen = self._prm["length"], self._prm["theta_cam"], int(hold_frozen)
        p.frozen[:] = self._prm["frozen"]
        p.sigma_a[:] = self._prm["sigma_a"]
        p.sigma_om[:] = self._prm["sigma_om"]
        p.quirks_fixed = self._prm["quirks_fixed"]      # _capi.FIX_Q1 | FIX_Q3 | FIX_Q6 | FIX_Q9 (include/dvi_eskf.h); 0 = the v1 source
        h = C.c_void_p()
        _capi.check(self.lib.dvi_eskf_create(C.byref(p), self.batch, _capi.F64 if dtype == torch.float64 else _capi.F32,
                                             self.device.index or 0, C.byref(h)), "dvi_eskf_create")
        self._h = h

    # -- lifetime --------------------------------------------------------------------
    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and getattr(self, "lib", None) is not None:
            self.lib.dvi_eskf_destroy(h)
            self._h = None

    def clone(self) -> "BatchedESKF":
        """copy.deepcopy semantics (Simulator.py:150): independent device buffers."""
        h = C.c_void_p()
        _capi.check(self.lib.dvi_eskf_clone(self._h, self._stream(), C.byref(h)), "dvi_eskf_clone")
        other = BatchedESKF(self.batch, dtype=self.dtype, device=self.device, _handle=h, **self._prm)
        other.Q, other.R = self.Q.copy(), self.R.copy()
        return other

    def __deepcopy__(self, memo):
        return self.clone()

    def copy_state_from(self, other: "BatchedESKF") -> None:
        """Device-to-device restore of x, P and the buffers from another handle."""
        _capi.check(self.lib.dvi_eskf_copy_state(self._h, other._h, self._stream()), "dvi_eskf_copy_state")

    def copy_instances_from(self, other: "BatchedESKF", first: int = 0) -> None:
        """This filter's instances <- instances [first, first + self.batch) of ``other`` (device-to-device)."""
        _capi.check(self.lib.dvi_eskf_copy_instances(self._h, other._h, int(first), self._stream()), "dvi_eskf_copy_instances")

    # -- helpers ---------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, shape_tail, allow_broadcast=True):
        """Returns (tensor, per_instance flag).  A tensor already on the device with the
        right dtype is used as is (zero copy)."""
        if a is None:
            return None, 0
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
        t = t.to(device=self.device, dtype=self.dtype).contiguous()
        tail = tuple(shape_tail)
        if t.shape == (self.batch, *tail):
            return t, 1
        if allow_broadcast and (t.shape == tail or t.shape == (1, *tail) or (tail == () and t.numel() == 1)):
            return t.reshape((1, *tail)), 0
        raise ValueError(f"expected shape {(self.batch, *tail)} or {tail}, got {tuple(t.shape)}")

    def set_noise_per_instance(self, q_diag, r_diag, sigma_om=None) -> None:
        """One noise model per instance: q_diag [B,13] = diag(Q), r_diag [B,7] = diag(R), sigma_om [B,3]
        (optional) = Jacobian evaluation point.  ``set_noise`` returns to a shared model."""
        q = np.ascontiguousarray(np.asarray(q_diag, dtype=np.float64).reshape(self.batch, 13))
        r = np.ascontiguousarray(np.asarray(r_diag, dtype=np.float64).reshape(self.batch, 7))
        so = None if sigma_om is None else np.ascontiguousarray(np.asarray(sigma_om, dtype=np.float64).reshape(self.batch, 3))
        dp = C.POINTER(C.c_double)
        _capi.check(self.lib.dvi_eskf_set_noise_per_instance(self._h, q.ctypes.data_as(dp), r.ctypes.data_as(dp),
                                                             None if so is None else so.ctypes.data_as(dp),
                                                             self._stream()), "dvi_eskf_set_noise_per_instance")
        self.Q, self.R = np.diag(q[0]), np.diag(r[0])

    def log_rows(self, t: float) -> torch.Tensor:
        """The 30-column trajectory-log row of every instance (FilterTraj.py:12-41), formed on the
        device: [B,30]."""
        rows = torch.empty((self.batch, 30), dtype=self.dtype, device=self.device)
        _capi.check(self.lib.dvi_eskf_log_rows(self._h, float(t), _ptr(rows), self._stream()), "dvi_eskf_log_rows")
        return rows

    def reserve(self, n_pred: int) -> None:
        """Size the workspace for epochs of up to ``n_pred`` predicts (needed before capturing such an
        epoch into a HIP graph; optional otherwise)."""
        _capi.check(self.lib.dvi_eskf_reserve(self._h, int(n_pred)), "dvi_eskf_reserve")

    def flush(self) -> None:
        _capi.check(self.lib.dvi_eskf_flush(self._h, self._stream()), "dvi_eskf_flush")

    def keep_jacobians(self, on: bool = True) -> None:
        """Let every predict leave what ``jacobians()`` needs (Filter.Fx / Filter.Fi) in the workspace."""
        _capi.check(self.lib.dvi_eskf_keep_jacobians(self._h, int(bool(on))), "dvi_eskf_keep_jacobians")

    def jacobians(self):
        """(Fx [B,24,24], Fi [B,24,13]) of the last predict (Filter.py:249-268,279-285)."""
        Fx = torch.empty((self.batch, 24, 24), dtype=self.dtype, device=self.device)
        Fi = torch.empty((self.batch, 24, 13), dtype=self.dtype, device=self.device)
        _capi.check(self.lib.dvi_eskf_get_jacobians(self._h, _ptr(Fx), _ptr(Fi), self._stream()), "dvi_eskf_get_jacobians")
        return Fx, Fi

    def update_mse(self, cam_ref6, imu_ref6=None) -> torch.Tensor:
        """Filter.calculate_update_mse (Filter.py:397-418) of every instance [B], reduced on the device."""
        dp = C.POINTER(C.c_double)
        cam = np.ascontiguousarray(np.asarray(cam_ref6, dtype=np.float64).reshape(6))
        imu = None if imu_ref6 is None else np.ascontiguousarray(np.asarray(imu_ref6, dtype=np.float64).reshape(6))
        out = torch.empty(self.batch, dtype=self.dtype, device=self.device)
        _capi.check(self.lib.dvi_eskf_update_mse(self._h, cam.ctypes.data_as(dp), None if imu is None else imu.ctypes.data_as(dp),
                                                 _ptr(out), self._stream()), "dvi_eskf_update_mse")
        return out

    def set_lanes_per_instance(self, lanes: int) -> None:
        _capi.check(self.lib.dvi_eskf_set_lanes_per_instance(self._h, int(lanes)), "set_lanes_per_instance")

    # -- noise / state ----------------------------------------------------------------
    def set_noise(self, Q=None, R=None) -> None:
        """Filter.Q / Filter.R (Filter.py:67-75,110-117); both must be diagonal."""
        dp = C.POINTER(C.c_double)
        q = None if Q is None else np.ascontiguousarray(np.asarray(Q, dtype=np.float64).reshape(13, 13))
        r = None if R is None else np.ascontiguousarray(np.asarray(R, dtype=np.float64).reshape(7, 7))
        _capi.check(self.lib.dvi_eskf_set_noise(self._h, None if q is None else q.ctypes.data_as(dp),
                                                None if r is None else r.ctypes.data_as(dp)), "dvi_eskf_set_noise")
        if q is not None:
            self.Q = q.copy()
        if r is not None:
            self.R = r.copy()

    def set_state(self, x=None, P=None, om_old=None, acc_old=None, q_old=None) -> None:
        tx, _ = self._bcast(x, (26,))
        tP, _ = self._bcast(P, (24, 24))
        to, _ = self._bcast(om_old, (3,))
        ta, _ = self._bcast(acc_old, (3,))
        tq, _ = self._bcast(q_old, (4,))
        _capi.check(self.lib.dvi_eskf_set_state(self._h, _ptr(tx), _ptr(tP), _ptr(to), _ptr(ta), _ptr(tq),
                                                self._stream()), "dvi_eskf_set_state")

    def _bcast(self, a, tail):
        if a is None:
            return None, 0
        t, per = self._dev(a, tail)
        if not per:
            t = t.expand(self.batch, *tail).contiguous()
        return t, 1

    def get_state(self, what=("x", "P", "om_old", "acc_old", "q_old")):
        out = {}
        mk = lambda *s: torch.empty((self.batch, *s), dtype=self.dtype, device=self.device)
        bufs = {"x": mk(26) if "x" in what else None, "P": mk(24, 24) if "P" in what else None,
                "om_old": mk(3) if "om_old" in what else None, "acc_old": mk(3) if "acc_old" in what else None,
                "q_old": mk(4) if "q_old" in what else None}
        _capi.check(self.lib.dvi_eskf_get_state(self._h, _ptr(bufs["x"]), _ptr(bufs["P"]), _ptr(bufs["om_old"]),
                                                _ptr(bufs["acc_old"]), _ptr(bufs["q_old"]), self._stream()),
                    "dvi_eskf_get_state")
        for k in what:
            out[k] = bufs[k]
        return out

    @property
    def x(self) -> torch.Tensor:
        return self.get_state(("x",))["x"]

    @property
    def P(self) -> torch.Tensor:
        return self.get_state(("P",))["P"]

    def calibration(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Estimated extrinsic DOFs [B,6] (state.dofs); ``out``: a preallocated [B,6] device tensor to fill."""
        if out is None:
            out = torch.empty((self.batch, 6), dtype=self.dtype, device=self.device)
        elif out.shape != (self.batch, 6) or out.dtype != self.dtype or out.device != self.device or not out.is_contiguous():
            raise ValueError("out must be a contiguous [B,6] tensor of the filter's dtype on its device")
        _capi.check(self.lib.dvi_eskf_get_calibration(self._h, _ptr(out), self._stream()), "dvi_eskf_get_calibration")
        return out

    # -- the hot path -------------------------------------------------------------------
    def predict(self, dt, om, acc) -> None:
        """Filter.propagate (Filter.py:219-230) for every instance."""
        tdt, pdt = self._dev(dt, ())
        tom, pim = self._dev(om, (3,))
        tac, pa = self._dev(acc, (3,))
        if pim != pa:
            raise ValueError("om and acc must both be per-instance or both broadcast")
        _capi.check(self.lib.dvi_eskf_predict(self._h, _ptr(tdt), pdt, _ptr(tom), _ptr(tac), pim, self._stream()),
                    "dvi_eskf_predict")

    def _meas(self, pos, quat, notch):
        tp, pp = self._dev(pos, (3,))
        tq, pq = self._dev(quat, (4,))
        tn, pn = self._dev(notch, ())
        if not (pp == pq == pn):
            if pp or pq or pn:   # mixed: expand the broadcast ones
                tp, _ = self._bcast(tp if pp else tp.reshape(3), (3,))
                tq, _ = self._bcast(tq if pq else tq.reshape(4), (4,))
                tn, _ = self._bcast(tn if pn else tn.reshape(()), ())
                pp = 1
        return tp, tq, tn, pp

    def update(self, pos, quat, notch, want_gain: bool = False):
        """Filter.update (Filter.py:351-395).  Returns (status[B] int32, K[B,24,7] | None)."""
        tp, tq, tn, per = self._meas(pos, quat, notch)
        K = torch.empty((self.batch, 24, 7), dtype=self.dtype, device=self.device) if want_gain else None
        status = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        _capi.check(self.lib.dvi_eskf_update(self._h, _ptr(tp), _ptr(tq), _ptr(tn), per, _ptr(K), _ptr(status),
                                             self._stream()), "dvi_eskf_update")
        return status, K

    def step(self, dt, om, acc, pos, quat, notch, want_gain: bool = False, status: Optional[torch.Tensor] = None):
        """One predict + one update in a single launch."""
        tdt, pdt = self._dev(dt, ())
        tom, pim = self._dev(om, (3,))
        tac, pa = self._dev(acc, (3,))
        if pim != pa:
            raise ValueError("om and acc must both be per-instance or both broadcast")
        tp, tq, tn, per = self._meas(pos, quat, notch)
        K = torch.empty((self.batch, 24, 7), dtype=self.dtype, device=self.device) if want_gain else None
        if status is None:
            status = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        _capi.check(self.lib.dvi_eskf_step(self._h, _ptr(tdt), pdt, _ptr(tom), _ptr(tac), pim, _ptr(tp), _ptr(tq),
                                           _ptr(tn), per, _ptr(K), _ptr(status), self._stream()), "dvi_eskf_step")
        return status, K

    def step_from(self, src: "BatchedESKF", dt, om, acc, pos, quat, notch, want_gain: bool = False,
                  status: Optional[torch.Tensor] = None):
        """``self <- step(src)``: one predict + one update that READS the state of ``src`` (left untouched)
        and writes the result here -- a restart from a snapshot without copying it first."""
        tdt, pdt = self._dev(dt, ())
        tom, pim = self._dev(om, (3,))
        tac, pa = self._dev(acc, (3,))
        if pim != pa:
            raise ValueError("om and acc must both be per-instance or both broadcast")
        tp, tq, tn, per = self._meas(pos, quat, notch)
        K = torch.empty((self.batch, 24, 7), dtype=self.dtype, device=self.device) if want_gain else None
        if status is None:
            status = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        _capi.check(self.lib.dvi_eskf_step_from(self._h, src._h, 1, _ptr(tdt), pdt, _ptr(tom), _ptr(tac), pim, _ptr(tp),
                                                _ptr(tq), _ptr(tn), per, _ptr(K), _ptr(status), self._stream()),
                    "dvi_eskf_step_from")
        return status, K

    def run_epochs(self, n_pred, dt, om, acc, pos, quat, notch, t_log=None, src: Optional["BatchedESKF"] = None,
                   status: Optional[torch.Tensor] = None, t_pred=None, rows_instances: int = 0):
        """A whole run (Filter.run, Filter.py:144-168) enqueued by one call: epoch e = n_pred[e]
        predicts + the update with measurement e.  dt [total] | [total,B]; om/acc [total,3] |
        [total,B,3]; pos [E,3] | [E,B,3]; quat [E,4] | [E,B,4]; notch [E] | [E,B].
        Returns (status [E,B] int32, rows [E,B,30] or None); rows are the trajectory-log rows
        after every update when their time stamps ``t_log`` [E] are given.  ``src``: the first epoch
        starts from that filter's state (dvi_eskf_run_epochs_from).  ``status``: a caller-owned [E,B]
        int32 tensor to write the status words into (none is allocated then).  Tensor arguments of the
        filter's dtype on its device are used as they are.
        ``t_pred`` [total] instead of ``t_log``: the reference's own log -- one row per IMU sample (the propagated
        state, Filter.py:230), the row of an epoch's last sample replaced by the state after the update
        (Filter.py:393) -- for the first ``rows_instances`` instances (0 = all): rows [total,n,30]."""
        n_pred = np.ascontiguousarray(np.asarray(n_pred, dtype=np.int32))
        E, total = int(n_pred.shape[0]), int(n_pred.sum())

        def dev(a):
            t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
            return t.to(self.device, self.dtype).contiguous()

        tdt, tom, tac, tp, tq, tn = dev(dt), dev(om), dev(acc), dev(pos), dev(quat), dev(notch)
        pdt = 1 if tdt.dim() == 2 else 0
        pim = 1 if tom.dim() == 3 else 0
        per = 1 if tp.dim() == 3 else 0
        B = self.batch
        if tdt.shape != ((total, B) if pdt else (total,)):
            raise ValueError("dt must be [sum(n_pred)] or [sum(n_pred),B]")
        if tom.shape != ((total, B, 3) if pim else (total, 3)) or tac.shape != tom.shape:
            raise ValueError("om/acc must be [sum(n_pred),3] or [sum(n_pred),B,3]")
        if tp.shape != ((E, B, 3) if per else (E, 3)) or tq.shape != ((E, B, 4) if per else (E, 4)) or \
                tn.shape != ((E, B) if per else (E,)):
            raise ValueError("pos/quat/notch must be [E,3]/[E,4]/[E] or [E,B,3]/[E,B,4]/[E,B]")
        if status is None:
            status = torch.zeros((E, B), dtype=torch.int32, device=self.device)
        elif status.shape != (E, B) or status.dtype != torch.int32 or not status.is_contiguous():
            raise ValueError("status must be a contiguous int32 [E,B] tensor")
        rows = tl = tpr = None
        if t_log is not None and t_pred is not None:
            raise ValueError("t_log (a row per update) and t_pred (a row per IMU sample) exclude each other")
        if t_log is not None:
            tl = np.ascontiguousarray(np.asarray(t_log, dtype=np.float64))
            if tl.shape != (E,):
                raise ValueError("t_log must be [E]")
            rows = torch.empty((E, B, 30), dtype=self.dtype, device=self.device)
        n_rows_inst = int(rows_instances) if 0 < int(rows_instances) < B else B
        if t_pred is not None:
            tpr = np.ascontiguousarray(np.asarray(t_pred, dtype=np.float64))
            if tpr.shape != (total,):
                raise ValueError("t_pred must be [sum(n_pred)]")
            rows = torch.empty((total, n_rows_inst, 30), dtype=self.dtype, device=self.device)
        sc = _capi.Schedule()
        sc.n_epochs = E
        sc.n_pred = n_pred.ctypes.data_as(C.POINTER(C.c_int32))
        sc.dt, sc.dt_per_instance = _ptr(tdt), pdt
        sc.om, sc.acc, sc.imu_per_instance = _ptr(tom), _ptr(tac), pim
        sc.pos, sc.quat_xyzw, sc.notch, sc.meas_per_instance = _ptr(tp), _ptr(tq), _ptr(tn), per
        sc.status = _ptr(status)
        sc.t_log = tl.ctypes.data_as(C.POINTER(C.c_double)) if tl is not None else None
        sc.rows_out = _ptr(rows) if tl is not None else None
        sc.t_pred = tpr.ctypes.data_as(C.POINTER(C.c_double)) if tpr is not None else None
        sc.rows_pred_out = _ptr(rows) if tpr is not None else None
        sc.rows_instances = n_rows_inst if tpr is not None else 0
        if src is not None:
            _capi.check(self.lib.dvi_eskf_run_epochs_from(self._h, src._h, C.byref(sc), self._stream()), "dvi_eskf_run_epochs_from")
        else:
            _capi.check(self.lib.dvi_eskf_run_epochs(self._h, C.byref(sc), self._stream()), "dvi_eskf_run_epochs")
        return status, rows

    def gather_calibration(self, rccl_comm, world_size: int) -> torch.Tensor:
        """[world*B,6] calibration estimates of every rank: ncclAllGather on the caller's
        communicator (an ``ncclComm_t`` as an integer / c_void_p).  With torch.distributed use
        ``dvi_ekf_amd.dist.gather_calibration`` instead -- same collective through the process group."""
        out = torch.empty((int(world_size) * self.batch, 6), dtype=self.dtype, device=self.device)
        _capi.check(self.lib.dvi_eskf_gather_calibration(self._h, _ptr(out), C.c_void_p(int(rccl_comm)), self._stream()),
                    "dvi_eskf_gather_calibration")
        return out

    def epoch(self, dt, om, acc, pos, quat, notch, want_gain: bool = False):
        """n predicts then one update in a single launch (Filter.run_one_epoch).
        dt [n] | [n,B]; om/acc [n,3] | [n,B,3]."""
        tdt = torch.as_tensor(np.asarray(dt) if not isinstance(dt, torch.Tensor) else dt).to(self.device, self.dtype)
        tom = torch.as_tensor(np.asarray(om) if not isinstance(om, torch.Tensor) else om).to(self.device, self.dtype)
        tac = torch.as_tensor(np.asarray(acc) if not isinstance(acc, torch.Tensor) else acc).to(self.device, self.dtype)
        tdt, tom, tac = tdt.contiguous(), tom.contiguous(), tac.contiguous()
        n = tdt.shape[0]
        pdt = 1 if tdt.dim() == 2 else 0
        pim = 1 if tom.dim() == 3 else 0
        if pdt and tdt.shape != (n, self.batch):
            raise ValueError("dt must be [n] or [n,B]")
        if tom.shape != ((n, self.batch, 3) if pim else (n, 3)) or tac.shape != tom.shape:
            raise ValueError("om/acc must be [n,3] or [n,B,3]")
        tp, tq, tn, per = self._meas(pos, quat, notch)
        K = torch.empty((self.batch, 24, 7), dtype=self.dtype, device=self.device) if want_gain else None
        status = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        _capi.check(self.lib.dvi_eskf_epoch(self._h, n, _ptr(tdt), pdt, _ptr(tom), _ptr(tac), pim, _ptr(tp), _ptr(tq),
                                            _ptr(tn), per, _ptr(K), _ptr(status), self._stream()), "dvi_eskf_epoch")
        return status, K
