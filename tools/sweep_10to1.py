"""(GPU box) the reference's 10:1 schedule (ten predicts per camera frame) through dvi_eskf_run_epochs at full
batches; run once per setting of DVI_ESKF_EPOCH_SPLIT_MAX / DVI_ESKF_EPOCH_STREAMS_MAX:
   default                                   two-role pre kernel above B = 40 960, one stream
   DVI_ESKF_EPOCH_SPLIT_MAX=-1               eskf_nominal_kernel + eskf_jac_kernel, one stream
   ... DVI_ESKF_EPOCH_STREAMS_MAX=65536      the same on two streams, half a batch each"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dvi_ekf_amd.workloads import long_horizon
dev = torch.device("cuda:0")
tag = "split_max=%s streams_max=%s" % (os.environ.get("DVI_ESKF_EPOCH_SPLIT_MAX", "-"), os.environ.get("DVI_ESKF_EPOCH_STREAMS_MAX", "-"))
for B, frozen in ((65536, (1,) * 6), (65536, (0,) * 6), (32768, (0,) * 6)):
    wl = long_horizon(batch=B, n_steps=400, imu_hz=300.0, cam_hz=30.0, frozen=frozen, hold_frozen=True)
    for dtype in (torch.float64, torch.float32):
        om, acc = wl.synthesize_imu_on_device(dtype=dtype, device=dev)
        om, acc = om[1:].contiguous(), acc[1:].contiguous()
        if os.environ.get("PER_INSTANCE_IMU") == "1":     # every instance its own stream (same numbers), as in a Monte-Carlo batch
            om = om.unsqueeze(1).expand(-1, B, -1).contiguous()
            acc = acc.unsqueeze(1).expand(-1, B, -1).contiguous()
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)
        dt, pos, quat, notch = t(wl.dt), t(wl.pos), t(wl.quat), t(wl.notch)
        f = wl.make_filter(dtype=dtype, device=dev)
        snap = f.clone()
        times = []
        for _ in range(4):
            f.copy_state_from(snap)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f.run_epochs(wl.n_pred, dt, om, acc, pos, quat, notch); e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e-3)
        tt = min(times[1:])
        st, _ = f.run_epochs(wl.n_pred, dt, om, acc, pos, quat, notch)
        print("om", tuple(om.shape), "frozen", frozen[0], "skipped", int((st & 1).sum()), "finite", bool(torch.isfinite(f.x).all()), "max|x| %.1f" % float(f.x.abs().max()), end="  ")
        print(tag, B, str(dtype)[-7:], "predicts/frame", sorted(set(int(n) for n in wl.n_pred)), "us/frame %.1f" % (tt / len(wl.n_pred) * 1e6), flush=True)
        del f, snap, om, acc
