"""``State`` (26) and ``ErrorState`` (24): host-side value objects with the reference's
surface (filter/state.py:11-129).  The device kernels implement the same injection
(csrc/eskf_kernels.hip, update phase); these classes exist so that code written against
the reference API (``kf._states.dofs``, ``State.from_array`` ...) keeps working."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from .quaternion import Quaternion


@dataclass(frozen=True)
class State:
    SIZE = 26

    p: np.ndarray            # W_p_BW
    v: np.ndarray            # WW_v_BW
    q: Quaternion            # R_WB
    dofs: np.ndarray         # 6 extrinsic DOFs
    notch_dofs: np.ndarray   # notch angle, rate, acceleration
    p_cam: np.ndarray        # W_p_CW
    q_cam: Quaternion        # R_WC

    def __post_init__(self):
        n = len(self.p) + len(self.v) + 4 + len(self.dofs) + len(self.notch_dofs) + len(self.p_cam) + 4
        assert n == State.SIZE

    @staticmethod
    def from_vector(x26) -> "State":
        """From the flat device layout p v q(xyzw) dofs notch p_cam q_cam(xyzw)."""
        x = np.asarray(x26, dtype=float).reshape(26)
        return State(x[0:3].copy(), x[3:6].copy(), Quaternion(val=x[6:10]), x[10:16].copy(), x[16:19].copy(),
                     x[19:22].copy(), Quaternion(val=x[22:26]))

    @property
    def vector(self) -> np.ndarray:
        return np.concatenate([np.reshape(self.p, 3), np.reshape(self.v, 3), self.q.xyzw, np.reshape(self.dofs, 6),
                               np.reshape(self.notch_dofs, 3), np.reshape(self.p_cam, 3), self.q_cam.xyzw])

    def __add__(self, err: "ErrorState") -> "State":
        """Error injection (state.py:46-60)."""
        q = self.q * err.dq
        q.normalise()
        q_cam = self.q_cam * err.dqc
        q_cam.normalise()
        return State(np.reshape(self.p, 3) + err.dp.reshape(3), np.reshape(self.v, 3) + err.dv.reshape(3), q,
                     self.dofs + err.ddofs, self.notch_dofs + err.dndofs.reshape(3),
                     np.reshape(self.p_cam, 3) + err.dpc.reshape(3), q_cam)

    @staticmethod
    def from_array(vec: List[np.ndarray]) -> "State":
        """state.py:62-74: [p, v, R_WB, dofs, notch, notchd, notchdd, p_cam, R_WC]."""
        return State(np.squeeze(vec[0]), np.squeeze(vec[1]), Quaternion(val=np.squeeze(vec[2]), do_normalise=True),
                     np.squeeze(vec[3]), np.array(vec[4:7], dtype=float).squeeze(), np.squeeze(vec[7]),
                     Quaternion(val=np.squeeze(vec[8]), do_normalise=True))

    @property
    def array(self) -> List[np.ndarray]:
        return [self.p, self.v, self.q.rot, self.dofs, self.notch_dofs, self.p_cam, self.q_cam.rot]


@dataclass(frozen=True)
class ErrorState:
    SIZE = State.SIZE - 2

    dp: np.ndarray
    dv: np.ndarray
    dq: Quaternion
    ddofs: np.ndarray
    dndofs: np.ndarray
    dpc: np.ndarray
    dqc: Quaternion
    theta: np.ndarray
    theta_c: np.ndarray

    @staticmethod
    def from_array(vec: np.ndarray) -> "ErrorState":
        """state.py:105-129 -- note dqc rotates about the IMU error axis `theta` (line 124)."""
        vec = np.asarray(vec, dtype=float)
        assert len(vec) == ErrorState.SIZE
        theta, theta_c = vec[6:9], vec[21:24]
        dq = Quaternion.about_axis(float(np.linalg.norm(theta)), axis=theta)
        dqc = Quaternion.about_axis(float(np.linalg.norm(theta_c)), axis=theta)
        return ErrorState(np.asarray(vec[0:3]), np.asarray(vec[3:6]), dq, np.asarray(vec[9:15]),
                          np.asarray(vec[15:18]), np.asarray(vec[18:21]), dqc, np.asarray(theta), np.asarray(theta_c))
