"""Child process of tests/test_gpu_step_kernel.py: a run of pair steps enqueued as ONE launch (dvi_eskf_run_epochs with
one predict per epoch -> eskf_run_kernel, the time loop in the kernel) against the same steps enqueued one by one, in the
same process: state, covariance and every epoch's status words must be the same bits.  The block size of the run kernel
comes from the environment (DVI_ESKF_RUN_ROUNDS), hence the child.  usage: _run_child.py <batch> <f64|f32> <repo root> [steps]"""
import sys

import numpy as np
import torch

sys.path.insert(0, sys.argv[3])
from dvi_ekf_amd.workloads import monte_carlo_pairs      # noqa: E402


def same(a, b, what):
    for k in a:
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        assert np.array_equal(x, y, equal_nan=True), (what, k, float(np.nanmax(np.abs(x - y))))


def main(B, dtype, pin=False, H=6):
    dev = torch.device("cuda", 0)
    wl = monte_carlo_pairs(B, horizon=H, seed=4, device=dev, dtype=dtype)
    f = wl.make_filter(dtype=dtype)
    if pin:     # per-instance noise records (a tuning population): the one-launch kernels with the 664-word images
        rng = np.random.default_rng(5)
        q13 = np.tile(np.diag(wl.Q), (B, 1)) * rng.uniform(0.5, 2.0, (B, 13))
        q13[:, 3:6] = 0.0
        f.set_noise_per_instance(q13, np.tile(np.diag(wl.R), (B, 1)) * rng.uniform(0.5, 2.0, (B, 7)))
    snap = f.clone()
    # (a NaN camera position in epoch 2 of instance 3: flagged in that epoch's status words, on both paths)
    pos = wl.pos.clone()
    pos[2, 3, 0] = float("nan")
    # A: step by step
    sts = []
    for k in range(H):
        st, _ = f.step(wl.dt[k], wl.om[k], wl.acc[k], pos[k], wl.quat[k], wl.notch[k])
        sts.append(st.clone())
    ref = f.get_state()
    ref_st = torch.stack(sts)
    assert int(ref_st[2, 3]) & 2
    # B: one launch
    g = snap.clone()
    st, rows = g.run_epochs([1] * H, wl.dt, wl.om, wl.acc, pos, wl.quat, wl.notch)
    assert rows is None
    assert torch.equal(st, ref_st), "status words"
    same(g.get_state(), ref, "run_epochs")
    # C: the same run started from the snapshot handle, over a filter that holds something else (and a pending injection)
    h2 = f.clone()
    h2.step(wl.dt[0], wl.om[0], wl.acc[0], wl.pos[0], wl.quat[0], wl.notch[0])
    st2 = torch.full((H, B), -1, dtype=torch.int32, device=dev)
    h2.run_epochs([1] * H, wl.dt, wl.om, wl.acc, pos, wl.quat, wl.notch, src=snap, status=st2)
    assert torch.equal(st2, ref_st), "status words (from)"
    same(h2.get_state(), ref, "run_epochs_from")
    # D: continuing a run: the second half enqueued on its own must pick up the injection the first half left pending
    g2 = snap.clone()
    # (three steps, then the rest: an odd and -- with H = 6 -- another odd number of steps, i.e. runs that end on either
    #  order of the rounds; what stays in LDS between the steps of a run is all in memory when the launch has finished)
    g2.run_epochs([1] * 3, wl.dt[:3], wl.om[:3], wl.acc[:3], pos[:3], wl.quat[:3], wl.notch[:3])
    g2.run_epochs([1] * (H - 3), wl.dt[3:], wl.om[3:], wl.acc[3:], pos[3:], wl.quat[3:], wl.notch[3:])
    same(g2.get_state(), ref, "two halves")
    # E: one input stream shared by the batch (instance 1's), against steps with the broadcast samples
    f3, g3 = snap.clone(), snap.clone()
    for k in range(H):
        f3.step(wl.dt[k, 1], wl.om[k, 1], wl.acc[k, 1], wl.pos[k, 1], wl.quat[k, 1], wl.notch[k, 1])
    g3.run_epochs([1] * H, wl.dt[:, 1].contiguous(), wl.om[:, 1].contiguous(), wl.acc[:, 1].contiguous(),
                  wl.pos[:, 1].contiguous(), wl.quat[:, 1].contiguous(), wl.notch[:, 1].contiguous())
    same(g3.get_state(), f3.get_state(), "shared streams")
    print("run child ok", B, dtype)


if __name__ == "__main__":
    main(int(sys.argv[1]), torch.float64 if sys.argv[2].startswith("f64") else torch.float32, pin=sys.argv[2].endswith("pin"),
         H=int(sys.argv[4]) if len(sys.argv) > 4 else 6)
