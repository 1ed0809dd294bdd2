"""Debug aid: the long-horizon kernel against the epoch-by-epoch path, field by field.  usage: dbg_epochs.py B dtype sched"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd.workloads import monte_carlo_pairs

B = int(sys.argv[1]); dtype = torch.float64 if sys.argv[2] == "f64" else torch.float32
n_pred = [int(v) for v in sys.argv[3].split(",")]
dev = torch.device("cuda", 0)
E, total, H = len(n_pred), sum(n_pred), 8
wl = monte_carlo_pairs(B, horizon=H, seed=4, device=dev, dtype=dtype)
f = wl.make_filter()
if len(sys.argv) > 4 and sys.argv[4] == 'bigR':
    f.set_noise(wl.Q, np.eye(7) * 1e30)
snap = f.clone()
idx = torch.arange(total, device=dev) % H
om, acc = wl.om[idx].contiguous(), wl.acc[idx].contiguous()
dt = torch.full((total, B), 0.1, device=dev, dtype=dtype)
ide = torch.arange(E, device=dev) % H
pos, quat, notch = wl.pos[ide].contiguous(), wl.quat[ide].contiguous(), wl.notch[ide].contiguous()
off = 0
for e, n in enumerate(n_pred):
    if n: f.epoch(dt[off:off+n], om[off:off+n], acc[off:off+n], pos[e], quat[e], notch[e])
    else: f.update(pos[e], quat[e], notch[e])
    off += n
g = snap.clone()
st, _ = g.run_epochs(n_pred, dt, om, acc, pos, quat, notch)
torch.cuda.synchronize()
a, b = f.get_state(), g.get_state()
names = ["p"]*3 + ["v"]*3 + ["q"]*4 + ["dofs"]*6 + ["notch"]*3 + ["pc"]*3 + ["qc"]*4
for k in a:
    x, y = a[k].cpu().numpy().astype(np.float64), b[k].cpu().numpy().astype(np.float64)
    d = np.abs(x - y)
    print(k, "max diff", float(np.nanmax(d)), "n diff", int((x != y).sum()), "of", x.size, "nan", int(np.isnan(y).sum()))
    if k == "x":
        for j in range(26):
            print("   ", j, names[j], float(d[:, j].max()), "ref", x[0, j], "got", y[0, j])
print("status", st.cpu().numpy().tolist()[:2])
# the C oracle on the same inputs: which side is right?
from oracle import c_oracle
R_ = np.eye(7) * 1e30 if (len(sys.argv) > 4 and sys.argv[4] == 'bigR') else wl.R
prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, True, wl.sigma_a, wl.sigma_om, wl.Q, R_)
o = c_oracle.OracleBatch(prm, wl.x0.cpu().numpy(), wl.P0.cpu().numpy(), wl.om0.cpu().numpy(), wl.acc0.cpu().numpy())
off = 0
for e, n in enumerate(n_pred):
    for k in range(n):
        o.predict(dt[off + k].cpu().numpy(), om[off + k].cpu().numpy(), acc[off + k].cpu().numpy())
    o.update(pos[e].cpu().numpy(), quat[e].cpu().numpy(), notch[e].cpu().numpy())
    off += n
for nm, s in (("epoch-by-epoch", a), ("one launch", b)):
    print(nm, "vs oracle: x", float(np.abs(s["x"].cpu().numpy() - o.x).max()), "P", float(np.abs(s["P"].cpu().numpy() - o.P).max() / np.abs(o.P).max()))
