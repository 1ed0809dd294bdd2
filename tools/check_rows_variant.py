"""Compares the row-resident covariance kernel (selector 1) with the LDS-image kernel (32 lanes per
instance) on the ragged perturbed batch of tests/test_gpu_parity.py; prints max relative deviations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import default_run
from tests.test_gpu_parity import _make, _perturbed_x0

def run(tune):
    B = 23
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=(0,) * 6)
    if tune:
        Q = Q.copy(); Q[3, 3] = Q[4, 4] = Q[5, 5] = 1e-4
    x0 = _perturbed_x0(ri, B)
    fa = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=32)
    fb = _make(cfg, ri, Q, R, sa, so, B, False, x0, lpi=1)
    ip = 0
    for k in range(3):
        for _ in range(ri.n_pred[k]):
            fa.predict(ri.dt[ip], ri.om[ip], ri.acc[ip]); fb.predict(ri.dt[ip], ri.om[ip], ri.acc[ip]); ip += 1
        sa_, Ka = fa.update(ri.pos[k], ri.quat[k], ri.notch[k], want_gain=True)
        sb_, _ = fb.update(ri.pos[k], ri.quat[k], ri.notch[k])   # (a gain request would route to the LDS-image kernel)
        Kb = Ka
        a, b = fa.get_state(), fb.get_state()
        out = {key: float(((a[key] - b[key]).abs().max() / (a[key].abs().max() + 1e-300)).cpu()) for key in a}
        out["K"] = float(((Ka - Kb).abs().max() / Ka.abs().max()).cpu())
        print("tune" if tune else "std", "frame", k, {k_: f"{v:.2e}" for k_, v in out.items()}, "status", sa_.cpu().numpy().max(), sb_.cpu().numpy().max())

run(False)
run(True)
