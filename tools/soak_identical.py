"""Soak check for races: B identical instances must stay bitwise identical through many steps, for every
covariance-kernel selector.  usage: soak_identical.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd.workloads import monte_carlo_pairs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 160
H = 16
wl = monte_carlo_pairs(B, horizon=H, seed=4, device="cuda:0")
for sel, pin in ((1, False), (2, False), (32, False), (1, True), (2, True)):
    f = wl.make_filter()
    f.set_lanes_per_instance(sel)
    x0 = wl.x0[0:1].repeat(B, 1)
    f.set_state(x=x0, P=wl.P0, om_old=wl.om0, acc_old=wl.acc0)
    if pin:
        f.set_noise_per_instance(np.tile(np.diag(wl.Q), (B, 1)), np.tile(np.diag(wl.R), (B, 1)))
    snap = f.clone()
    bad = 0
    for k in range(steps):
        i = k % H
        if i == 0:
            f.copy_state_from(snap)
        f.step(wl.dt[i, 0], wl.om[i, 0], wl.acc[i, 0], wl.pos[i, 0], wl.quat[i, 0], wl.notch[i, 0])
        if i == H - 1 or k == steps - 1:
            st = f.get_state()
            for key in ("x", "P"):
                ref = st[key][0:1]
                bad += int((st[key] != ref).any(dim=tuple(range(1, st[key].dim()))).sum())
    print(f"selector {sel} per-instance-noise {pin}: {steps} steps x {B} instances, mismatching instances: {bad}")
    assert bad == 0
print("soak ok")
