"""epoch launch time over (batch, predicts per epoch, dtype): two-role pre kernel (DVI_ESKF_PRE_LEGACY=1) against
eskf_nominal_kernel + eskf_jac_kernel -- run once per setting of the environment variable"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd.workloads import monte_carlo_pairs
from tools.bench_configs import _time
dev = torch.device("cuda:0")
for dtype in (torch.float64, torch.float32):
    for B in (1024, 8192, 16384, 32768, 65536):
        wl = monte_carlo_pairs(B, horizon=4, seed=4, device=dev)
        f = wl.make_filter(dtype=dtype)
        snap = f.clone()
        for N in (10, 33):
            f.reserve(N)
            dt = torch.full((N,), 0.01, device=dev, dtype=dtype)
            om = wl.om[0].to(dtype).unsqueeze(0).repeat(N, 1, 1).contiguous()
            acc = wl.acc[0].to(dtype).unsqueeze(0).repeat(N, 1, 1).contiguous()
            k = [0]
            def step():
                if k[0] % 4 == 0:
                    f.copy_state_from(snap)
                f.epoch(dt, om, acc, wl.pos[k[0] % 4].to(dtype), wl.quat[k[0] % 4].to(dtype), wl.notch[k[0] % 4].to(dtype))
                k[0] += 1
            t = _time(step, 16, warm=4)
            print(json.dumps({"legacy": os.environ.get("DVI_ESKF_PRE_LEGACY", "0"), "dtype": str(dtype)[-7:], "B": B, "N": N, "us_per_epoch": round(t * 1e6, 1)}), flush=True)
