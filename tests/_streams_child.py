"""Child process of tests/test_gpu_capi.py::test_run_epochs_on_three_streams_equals_epoch_by_epoch: DVI_ESKF_EPOCH_STREAMS (read once per
process) asks for three parts of the batch on three streams; the run must equal the epochs called one by one bit for bit.
usage: _streams_child.py <repo root> <batch> <f64|f32>"""
import sys

import numpy as np
import torch

sys.path.insert(0, sys.argv[1])
from dvi_ekf_amd.batch import BatchedESKF      # noqa: E402
from tests.helpers import default_run          # noqa: E402


def main(B, dtype):
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False)
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=[0] * 6, sigma_a=sa, sigma_om=so, hold_frozen=False, dtype=dtype)
    f.set_noise(Q, R)
    rng = np.random.default_rng(23)
    x0 = np.tile(ri.x0, (B, 1))
    x0[:, 10:13] += rng.normal(0, 0.02, (B, 3))
    x0[:, 13:16] += rng.normal(0, 2.0, (B, 3))
    f.set_state(x=x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    n_pred = np.array([5, 2, 9, 3], dtype=np.int32)
    E, total = len(n_pred), int(n_pred.sum())
    pos = ri.pos[:E, None, :] + rng.normal(0, 1e-3, (E, B, 3))
    quat = np.repeat(ri.quat[:E, None, :], B, axis=1)
    notch = rng.normal(0, 1e-3, (E, B))
    dt, om, acc = ri.dt[:total], ri.om[:total], ri.acc[:total]
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])        # (an injection is pending at the start)
    g = f.clone()
    status, _ = f.run_epochs(n_pred, dt, om, acc, pos, quat, notch)
    off = 0
    for e in range(E):
        n = int(n_pred[e])
        st, _ = g.epoch(dt[off:off + n], om[off:off + n], acc[off:off + n], pos[e], quat[e], notch[e])
        off += n
        assert torch.equal(status[e], st), e
    sf, sg = f.get_state(), g.get_state()
    for key in sf:
        assert torch.equal(sf[key], sg[key]), key
    assert bool(torch.isfinite(sf["x"]).all())
    print("streams child ok")


if __name__ == "__main__":
    main(int(sys.argv[2]), torch.float64 if sys.argv[3] == "f64" else torch.float32)
