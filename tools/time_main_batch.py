"""`main.py --batch B` on the full 139-frame run of the default trajectory (golden-compat plumbing: one IMU sample per frame)
against the Monte-Carlo driver's one-call run of the same frames: device time of Filter.run's run against run_epochs.
usage: time_main_batch.py [batch]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd import Config, Simulator
from dvi_ekf_amd.workloads import monte_carlo_pairs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config(os.path.join(root, "config.yaml"), base_dir=root)
cfg.golden_compat = True
cfg.max_vals = None
cfg.sim.do_fast_sim = False if hasattr(cfg.sim, "do_fast_sim") else None
cfg.batch = B
sim = Simulator(cfg)
frames = len(sim.camera.t) - 1
out = {"batch": B, "frames": frames}
for fast in (True, False):
    sim.reset_kf()
    sim.kf.fast_run = fast
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.kf.run(sim.camera, 0, "")
    torch.cuda.synchronize()
    out["filter_run_fast" if fast else "filter_run_stepwise"] = time.perf_counter() - t0
    out["rows"] = sim.kf.traj.num_values
    if fast:
        out["fast_run_split"] = sim.kf.last_run_timing
wl = monte_carlo_pairs(B, horizon=min(frames, 139), seed=4, device="cuda")
f = wl.make_filter()
H = wl.horizon
for _ in range(2):
    g = f.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run_epochs(np.ones(H, dtype=np.int32), wl.dt, wl.om, wl.acc, wl.pos, wl.quat, wl.notch)
    torch.cuda.synchronize()
    out["montecarlo_run_epochs"] = time.perf_counter() - t0
out["montecarlo_frames"] = H
out["device_run_over_montecarlo"] = out["fast_run_split"]["device_run_s"] / (out["montecarlo_run_epochs"] * (frames - 1) / H)
print(json.dumps(out))
