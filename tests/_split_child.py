"""Child process of tests/test_gpu_capi.py::test_run_epochs_error_after_the_fork_still_joins_the_side_streams: runs with a
TEST BUILD of the library (DVI_ESKF_LIB, -DESKF_TEST_HOOKS) whose two-stream path of dvi_eskf_run_epochs can be made to fail at a
given epoch (DVI_ESKF_TEST_FAIL_SPLIT_EPOCH).  usage: _split_child.py <repo root>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, sys.argv[1])
from dvi_ekf_amd._capi import EskfError        # noqa: E402
from dvi_ekf_amd.batch import BatchedESKF      # noqa: E402
from tests.helpers import default_run          # noqa: E402


def raises(fn):
    try:
        fn()
    except EskfError:
        return True
    return False


def main():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False)
    B = 8320            # 65 x 128, more than the long-horizon kernel takes: the two halves of the batch on two streams
    mk = lambda: BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=[0] * 6, sigma_a=sa, sigma_om=so, hold_frozen=False)
    f = mk()
    f.set_noise(Q, R)
    f.set_state(x=np.tile(ri.x0, (B, 1)), P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    n_pred = np.array([4, 3, 5, 2], dtype=np.int32)
    E, total = len(n_pred), int(n_pred.sum())
    dev = f.device
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    args = (n_pred, t(ri.dt[:total]), t(ri.om[:total]), t(ri.acc[:total]), t(ri.pos[:E]), t(ri.quat[:E]), t(ri.notch[:E]))
    f.reserve(int(n_pred.max()))
    c = f.clone()
    c.reserve(int(n_pred.max()))
    ref = f.clone()
    ref.run_epochs(*args)                                       # what an undisturbed run gives
    os.environ["DVI_ESKF_TEST_FAIL_SPLIT_EPOCH"] = "2"
    assert raises(lambda: f.run_epochs(*args)), "the hook did not fire: not the two-stream path, or not the test build"
    torch.cuda.synchronize(dev)                                 # (no fault, nothing left running unordered)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):                     # capture_end() in __exit__ raises if a stream is left unjoined
        assert raises(lambda: c.run_epochs(*args))
    torch.cuda.synchronize(dev)
    del os.environ["DVI_ESKF_TEST_FAIL_SPLIT_EPOCH"]
    # the handle is still usable: restored and run again it gives the undisturbed result
    h2 = mk()
    h2.set_noise(Q, R)
    h2.set_state(x=np.tile(ri.x0, (B, 1)), P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    f.copy_state_from(h2)
    f.run_epochs(*args)
    sf, sg = f.get_state(), ref.get_state()
    for key in sf:
        assert torch.equal(sf[key], sg[key]), key
    print("split child ok")


if __name__ == "__main__":
    main()
