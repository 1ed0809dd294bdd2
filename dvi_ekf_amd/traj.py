"""Filter trajectory log (30 columns) and the reference's on-disk text format.

Row layout: trajectory/FilterTraj.py:12-41; writer: tools/files.py:68-82
(``%.6f`` for t, ``%.9f`` otherwise, trailing space, one row per line)."""
from __future__ import annotations

from typing import List

import numpy as np

from .state import State


def euler_measurement_row(t: float, state: State) -> List[float]:
    return [t, *state.p, *state.v, *state.q.euler_xyz_deg, *state.q.wxyz, *np.rad2deg(state.dofs[:3]),
            *state.dofs[3:], *state.p_cam, *state.q_cam.euler_xyz_deg, *state.q_cam.wxyz]


class FilterTraj:
    labels_imu = ["x", "y", "z", "vx", "vy", "vz", "rx", "ry", "rz", "qw", "qx", "qy", "qz"]
    labels_imu_dofs = ["dof1", "dof2", "dof3", "dof4", "dof5", "dof6"]
    labels_camera = ["xc", "yc", "zc", "rx_degc", "ry_degc", "rz_degc", "qwc", "qxc", "qyc", "qzc"]
    labels = ["t", *labels_imu, *labels_imu_dofs, *labels_camera]

    def __init__(self, name: str):
        self.name = name
        self.reset()

    def reset(self) -> None:
        for lab in self.labels:
            setattr(self, lab, None)

    @property
    def num_values(self) -> int:
        return 0 if self.t is None else len(self.t)

    def append_propagated_states(self, t: float, state: State) -> None:
        data = euler_measurement_row(t, state)
        for i, lab in enumerate(self.labels):
            if getattr(self, lab) is None:
                setattr(self, lab, [data[i]])
            else:
                getattr(self, lab).append(data[i])

    def append_row(self, data) -> None:
        """A ready-made 30-column row (formed on the device, BatchedESKF.log_rows)."""
        for i, lab in enumerate(self.labels):
            if getattr(self, lab) is None:
                setattr(self, lab, [float(data[i])])
            else:
                getattr(self, lab).append(float(data[i]))

    def overwrite_last_row(self, data) -> None:
        for i, lab in enumerate(self.labels):
            getattr(self, lab)[-1] = float(data[i])

    def append_updated_states(self, t: float, state: State) -> None:
        """The row of an update step overwrites the last propagated row (FilterTraj.py:66-69)."""
        data = euler_measurement_row(t, state)
        for i, lab in enumerate(self.labels):
            getattr(self, lab)[-1] = data[i]

    def as_array(self) -> np.ndarray:
        return np.array([getattr(self, lab) for lab in self.labels], dtype=float).T


def save_trajectory(trajectory, filename: str) -> None:
    with open(filename, "w+") as f:
        out = []
        for n in range(trajectory.num_values):
            row = ""
            for lab in trajectory.labels:
                v = getattr(trajectory, lab)[n]
                row += f"{v:.6f}" if lab == "t" else f" {v:.9f}"
            out.append(row + " \n")
        f.write("".join(out))
