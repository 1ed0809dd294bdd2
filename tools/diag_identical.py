"""Developer diagnostic: identical instances must give identical results."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import default_run
from tests.test_gpu_scale import _filter
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=False, frozen=(1,) * 6, interframe=1)
f = _filter(cfg, Q, R, sa, so, B, True)
if lanes: f.set_lanes_per_instance(lanes)
f.set_state(x=ri.x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
for k in range(4):
    mode = sys.argv[3] if len(sys.argv) > 3 else 'step'
    if mode == 'step':
        f.step(ri.dt[k], ri.om[k], ri.acc[k], ri.pos[k], ri.quat[k], ri.notch[k])
    else:
        f.predict(ri.dt[k], ri.om[k], ri.acc[k])
    st = f.get_state(("x", "P"))
    x, P = st["x"], st["P"].reshape(B, -1)
    bx = (x != x[0:1]).any(dim=1); bP = (P != P[0:1]).any(dim=1)
    idx = torch.nonzero(bx | bP).flatten().cpu().numpy()
    print(f"step {k+1}: differing instances x:{int(bx.sum())} P:{int(bP.sum())}", 'first', idx[:12], 'mod64', np.unique(idx % 64)[:16] if len(idx) else '', 
          'max|dx|', float((x - x[0:1]).abs().max()), 'max|dP|', float((P - P[0:1]).abs().max()))
