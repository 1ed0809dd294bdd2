import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dvi_ekf_amd.workloads import long_horizon
dev = torch.device("cuda:0")
for B in (1024, 2048, 4096, 8192, 16384):
    wl = long_horizon(batch=B, n_steps=2000)
    for dtype in (torch.float32, torch.float64):
        om, acc = wl.synthesize_imu_on_device(dtype=dtype, device=dev)
        om, acc = om[1:].contiguous(), acc[1:].contiguous()
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
        print(os.environ.get("DVI_ESKF_EPOCH_STREAMS", "1"), B, str(dtype)[-7:], "us/frame %.1f" % (tt / len(wl.n_pred) * 1e6), "IMU steps/s %.3e" % (B * wl.n_steps / tt), flush=True)
