"""diagnostic: which outputs of eskf_pre4_kernel differ from the two-role pre kernel's, and by how much"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scen = sys.argv[1] if len(sys.argv) > 1 else "unfrozen"
d = tempfile.mkdtemp()
outs = []
for legacy in (True, False):
    env = dict(os.environ); env.pop("DVI_ESKF_PRE_LEGACY", None)
    if legacy: env["DVI_ESKF_PRE_LEGACY"] = "1"
    out = os.path.join(d, f"{legacy}.npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_pre4_child.py"), scen, out, ROOT], check=True, env=env, cwd=ROOT)
    outs.append(np.load(out))
a, b = outs
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    neq = ~((x == y) | (np.isnan(x) & np.isnan(y)))
    if neq.any():
        ax = tuple(range(x.ndim - 1)) if x.ndim > 1 else None
        idx = np.argwhere(neq)
        rel = np.abs(x - y)[neq] / np.maximum(1e-300, np.abs(x[neq]))
        print(k, "differs at", int(neq.sum()), "of", x.size, "max rel", rel.max(), "last-axis cols", sorted(set(idx[:, -1].tolist()))[:30], "first rows", sorted(set(idx[:, 0].tolist()))[:10])
    else:
        print(k, "equal")
