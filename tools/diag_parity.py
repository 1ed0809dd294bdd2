"""Developer diagnostic (GPU box): per-step forced parity of the HIP path vs the C
oracle, printing where the largest deviation sits."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import default_run
from tests.test_gpu_parity import _make, _oracle, _perturbed_x0

golden = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 16
cfg, ri, Q, R, sa, so = default_run(golden_compat=golden, fast=False, frozen=(0,) * 6, interframe=(1 if golden else None))
x0 = _perturbed_x0(ri, B)
f = _make(cfg, ri, Q, R, sa, so, B, golden, x0)
o = _oracle(cfg, ri, Q, R, sa, so, B, golden, x0)
q_old = o.x[:, 6:10].copy()
ip = 0
names = ['p'] * 3 + ['v'] * 3 + ['q'] * 4 + ['dofs'] * 6 + ['notch'] * 3 + ['pc'] * 3 + ['qc'] * 4
for k in range(nframes):
    for _ in range(ri.n_pred[k]):
        f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
        f.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
        o.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
        q_old = o.x[:, 6:10].copy(); ip += 1
        s = {k_: v.cpu().numpy() for k_, v in f.get_state().items()}
        dx = np.abs(s['x'] - o.x); b, c = np.unravel_index(dx.argmax(), dx.shape)
        dP = np.abs(s['P'] - o.P).reshape(B, -1); bp = dP.max(axis=1).argmax(); e = dP[bp].argmax()
        print(f"frame {k+1:2d} predict: dx {dx.max():.1e} (inst {b} {names[c]}[{c}] |x|={np.abs(o.x[b]).max():.1e})  dP {dP.max():.1e} (inst {bp} ({e//24},{e%24}) |P|={np.abs(o.P[bp]).max():.1e} P_el={o.P[bp].reshape(-1)[e]:.2e})")
    f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=q_old)
    st, K = f.update(ri.pos[k], ri.quat[k], ri.notch[k], want_gain=True)
    Ko = np.empty((B, 24, 7))
    for bb in range(B):
        _, Ko[bb] = o.update_one(bb, ri.pos[k], ri.quat[k], ri.notch[k])
    s = {k_: v.cpu().numpy() for k_, v in f.get_state().items()}
    dx = np.abs(s['x'] - o.x); b, c = np.unravel_index(dx.argmax(), dx.shape)
    dP = np.abs(s['P'] - o.P).reshape(B, -1); bp = dP.max(axis=1).argmax(); e = dP[bp].argmax()
    dK = np.abs(K.cpu().numpy() - Ko).reshape(B, -1); bk = dK.max(axis=1).argmax()
    print(f"frame {k+1:2d} update : dx {dx.max():.1e} (inst {b} {names[c]}[{c}] |x|={np.abs(o.x[b]).max():.1e})  dP {dP.max():.1e} (inst {bp} ({e//24},{e%24}) |P|={np.abs(o.P[bp]).max():.1e})  dK {dK.max():.1e} (inst {bk} |K|={np.abs(Ko[bk]).max():.1e})")
