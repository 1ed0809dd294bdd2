"""ctypes binding of the C oracle (oracle/eskf_oracle.c) -- TEST INFRASTRUCTURE
ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never from the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libeskf_oracle.so")
_SO_HP = os.path.join(_HERE, "_build", "libeskf_oracle_hp.so")   # __float128 build: the adjudicator


class OracleParams(C.Structure):
    _fields_ = [
        ("L", C.c_double), ("theta_cam", C.c_double), ("frozen", C.c_int32 * 6), ("hold_frozen", C.c_int32),
        ("sigma_a", C.c_double * 3), ("sigma_om", C.c_double * 3), ("Q", C.c_double * 169), ("R", C.c_double * 49),
    ]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, n) for n in ("eskf_oracle.c", "eskf_oracle.h", "Makefile")]
    stale = any(not os.path.exists(so) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in src)
                for so in (_SO, _SO_HP))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_libs = {}


def lib(hp: bool = False):
    """hp=True: the extended-precision (quad) build of the same source -- same entry points, same double
    arrays in and out, every intermediate in __float128."""
    if hp not in _libs:
        build()
        _lib = _libs[hp] = C.CDLL(_SO_HP if hp else _SO)
        dp = C.POINTER(C.c_double)
        pp = C.POINTER(OracleParams)
        _lib.eskf_oracle_init_rwb.argtypes = [dp, dp]
        _lib.eskf_oracle_predict.argtypes = [pp, dp, dp, dp, dp, dp, C.c_double, dp, dp, dp, dp]
        _lib.eskf_oracle_update.argtypes = [pp, dp, dp, dp, dp, C.c_double, dp]
        _lib.eskf_oracle_update.restype = C.c_int
        _lib.eskf_oracle_predict_batch.argtypes = [pp, C.c_int64, dp, dp, dp, dp, dp, dp, C.c_int, dp, dp]
        _lib.eskf_oracle_update_batch.argtypes = [pp, C.c_int64, dp, dp, dp, dp, dp, C.POINTER(C.c_int32)]
        _lib.eskf_oracle_num_threads.restype = C.c_int
        _lib.eskf_oracle_set_num_threads.restype = C.c_int
        _lib.eskf_oracle_set_num_threads.argtypes = [C.c_int]
    return _libs[hp]


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_params(L, theta_cam, frozen, hold_frozen, sigma_a, sigma_om, Q, R) -> OracleParams:
    prm = OracleParams()
    prm.L, prm.theta_cam, prm.hold_frozen = float(L), float(theta_cam), int(bool(hold_frozen))
    prm.frozen[:] = [int(bool(v)) for v in frozen]
    prm.sigma_a[:] = [float(v) for v in sigma_a]
    prm.sigma_om[:] = [float(v) for v in sigma_om]
    prm.Q[:] = np.asarray(Q, dtype=float).reshape(169).tolist()
    prm.R[:] = np.asarray(R, dtype=float).reshape(49).tolist()
    return prm


class OracleBatch:
    """B independent filter instances stepped by the C oracle (OpenMP over instances)."""

    def __init__(self, prm: OracleParams, x0, P0, om0, acc0, hp: bool = False):
        self.prm = prm
        self.hp = bool(hp)
        self.x = np.array(np.atleast_2d(np.asarray(x0, dtype=float)), order="C")              # [B,26] (own copy)
        self.B = self.x.shape[0]
        P0 = np.asarray(P0, dtype=float)
        self.P = np.ascontiguousarray(np.broadcast_to(P0, (self.B, 24, 24)).copy())
        self.om_old = np.ascontiguousarray(np.broadcast_to(np.asarray(om0, dtype=float), (self.B, 3)).copy())
        self.acc_old = np.ascontiguousarray(np.broadcast_to(np.asarray(acc0, dtype=float), (self.B, 3)).copy())
        self.Rwb = np.empty((self.B, 9))
        for b in range(self.B):
            lib(self.hp).eskf_oracle_init_rwb(_p(self.x[b]), _p(self.Rwb[b]))
        self.status = np.zeros(self.B, dtype=np.int32)

    def predict(self, dt, om, acc):
        dt = np.ascontiguousarray(np.atleast_1d(np.asarray(dt, dtype=float)))
        om = np.ascontiguousarray(np.broadcast_to(np.asarray(om, dtype=float), (self.B, 3)))
        acc = np.ascontiguousarray(np.broadcast_to(np.asarray(acc, dtype=float), (self.B, 3)))
        stride = 0 if dt.size == 1 else 1
        assert stride == 0 or dt.size == self.B
        lib(self.hp).eskf_oracle_predict_batch(C.byref(self.prm), self.B, _p(self.x), _p(self.P), _p(self.om_old),
                                        _p(self.acc_old), _p(self.Rwb), _p(dt), stride, _p(om), _p(acc))

    def update(self, pos, quat, notch):
        pos = np.ascontiguousarray(np.broadcast_to(np.asarray(pos, dtype=float), (self.B, 3)))
        quat = np.ascontiguousarray(np.broadcast_to(np.asarray(quat, dtype=float), (self.B, 4)))
        notch = np.ascontiguousarray(np.broadcast_to(np.asarray(notch, dtype=float), (self.B,)))
        lib(self.hp).eskf_oracle_update_batch(C.byref(self.prm), self.B, _p(self.x), _p(self.P), _p(pos), _p(quat),
                                       _p(notch), self.status.ctypes.data_as(C.POINTER(C.c_int32)))
        return self.status

    # single-instance helpers with Jacobian / gain outputs
    def predict_one(self, b, dt, om, acc):
        Fx = np.empty((24, 24))
        Fi = np.empty((24, 13))
        om = np.ascontiguousarray(om, dtype=float)
        acc = np.ascontiguousarray(acc, dtype=float)
        lib(self.hp).eskf_oracle_predict(C.byref(self.prm), _p(self.x[b]), _p(self.P[b]), _p(self.om_old[b]),
                                  _p(self.acc_old[b]), _p(self.Rwb[b]), float(dt), _p(om), _p(acc), _p(Fx), _p(Fi))
        return Fx, Fi

    def update_one(self, b, pos, quat, notch):
        K = np.empty((24, 7))
        pos = np.ascontiguousarray(pos, dtype=float)
        quat = np.ascontiguousarray(quat, dtype=float)
        st = lib(self.hp).eskf_oracle_update(C.byref(self.prm), _p(self.x[b]), _p(self.P[b]), _p(pos), _p(quat),
                                      float(notch), _p(K))
        return st, K


def num_threads() -> int:
    return int(lib().eskf_oracle_num_threads())


def set_num_threads(n: int) -> int:
    """OpenMP threads of the batched drivers from now on; returns the number in effect."""
    return int(lib().eskf_oracle_set_num_threads(int(n)))
