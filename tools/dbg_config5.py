import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvi_ekf_amd.workloads import long_horizon
B = 8192
dev = torch.device("cuda", 0)
wl = long_horizon(batch=B)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
for dtype in (torch.float64, torch.float32):
    f = wl.make_filter(dtype=dtype, device=dev)
    st, _ = f.run_epochs(wl.n_pred, t(wl.dt, dtype), t(wl.om, dtype), t(wl.acc, dtype), t(wl.pos, dtype), t(wl.quat, dtype), t(wl.notch, dtype))
    st = st.cpu().numpy()
    bad = np.argwhere(st != 0)
    x = f.x.double().cpu().numpy()
    print(dtype, "flagged (epoch, instance) pairs:", len(bad), "first:", bad[:5].tolist(), "instances:", len(set(bad[:, 1].tolist())))
    print("  |x| max per instance: max", np.abs(x).max(), "quantiles", np.quantile(np.abs(x).max(1), [0.5, 0.9, 0.99, 0.999, 1.0]))
    if len(bad):
        i = bad[0, 1]
        print("  x0 perturbation of first flagged instance (sigma units):", ((wl.x0[i, :6] - wl.x0[0, :6]) / np.sqrt(np.diag(wl.P0)[:6])).round(2))
        print("  final x of it", x[i].round(3))
