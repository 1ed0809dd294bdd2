"""Developer diagnostic (GPU box): gain K of the HIP path vs oracle vs a NumPy LDL^T on
the same covariance, for the worst instance of a forced update."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import default_run
from tests.test_gpu_parity import _make, _oracle, _perturbed_x0
np.set_printoptions(linewidth=200, precision=3)
B = 16
frame = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=False, frozen=(0,) * 6, interframe=1)
x0 = _perturbed_x0(ri, B)
f = _make(cfg, ri, Q, R, sa, so, B, True, x0)
o = _oracle(cfg, ri, Q, R, sa, so, B, True, x0)
sel = [18, 19, 20, 21, 22, 23, 15]
def ldl(S):
    n = S.shape[0]; L = np.eye(n); D = np.zeros(n)
    for j in range(n):
        D[j] = S[j, j] - (L[j, :j] ** 2 * D[:j]).sum()
        for i in range(j + 1, n):
            L[i, j] = (S[i, j] - (L[i, :j] * L[j, :j] * D[:j]).sum()) / D[j]
    return L, D
for k in range(frame):
    o.predict(ri.dt[k], ri.om[k], ri.acc[k])
    if k == frame - 1:
        f.set_state(x=o.x, P=o.P, om_old=o.om_old, acc_old=o.acc_old, q_old=o.x[:, 6:10].copy())
        Pin = o.P.copy()
        st, K = f.update(ri.pos[k], ri.quat[k], ri.notch[k], want_gain=True)
        K = K.cpu().numpy()
        Ko = np.empty((B, 24, 7))
        for b in range(B):
            _, Ko[b] = o.update_one(b, ri.pos[k], ri.quat[k], ri.notch[k])
        d = np.abs(K - Ko).reshape(B, -1).max(axis=1)
        b = d.argmax()
        P = np.tril(Pin[b]) + np.tril(Pin[b], -1).T
        PH = P[:, sel]; S = PH[sel, :] + R
        L, D = ldl(S)
        Z = np.linalg.solve(L, PH.T).T
        Kl = np.linalg.solve(L.T, (Z / D).T).T
        Ki = PH @ np.linalg.inv(S)
        print('inst', b, 'status', st.cpu().numpy()[b], 'D', D)
        print('eig S', np.linalg.eigvalsh(S))
        print('eig P min/max', np.linalg.eigvalsh(P)[[0, -1]])
        print('|K|', np.abs(Ko[b]).max(), ' gpu-oracle', np.abs(K[b] - Ko[b]).max(), ' npLDL-oracle', np.abs(Kl - Ko[b]).max(),
              ' npinv-oracle', np.abs(Ki - Ko[b]).max(), ' gpu-npLDL', np.abs(K[b] - Kl).max())
        print('per-column max |gpu-oracle|:', np.abs(K[b] - Ko[b]).max(axis=0))
        print('per-column max |K|        :', np.abs(Ko[b]).max(axis=0))
        print('residual of K S = PH:  gpu', np.abs(K[b] @ S - PH).max(), ' oracle', np.abs(Ko[b] @ S - PH).max(), ' npLDL', np.abs(Kl @ S - PH).max())
    else:
        o.update(ri.pos[k], ri.quat[k], ri.notch[k])
