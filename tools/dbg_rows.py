import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd.workloads import monte_carlo_pairs
B=int(sys.argv[1]); n_pred=[int(v) for v in sys.argv[2].split(",")]
frozen=tuple(int(c) for c in (sys.argv[3] if len(sys.argv)>3 else "000000"))
dev=torch.device("cuda",0); dtype=torch.float64
E,total,H=len(n_pred),sum(n_pred),8
wl=monte_carlo_pairs(B,horizon=H,seed=4,device=dev,dtype=dtype,frozen=frozen)
snap=wl.make_filter()
idx=torch.arange(total,device=dev)%H
om,acc=wl.om[idx].contiguous(),wl.acc[idx].contiguous()
dt=torch.full((total,B),0.1,device=dev,dtype=dtype)
ide=torch.arange(E,device=dev)%H
pos,quat,notch=wl.pos[ide].contiguous(),wl.quat[ide].contiguous(),wl.notch[ide].contiguous()
t_pred=np.arange(total)*0.1+200
g4=snap.clone(); _,rows=g4.run_epochs(n_pred,dt,om,acc,pos,quat,notch,t_pred=t_pred,rows_instances=min(B,3))
f4=snap.clone(); f4.keep_jacobians(True); _,rr=f4.run_epochs(n_pred,dt,om,acc,pos,quat,notch,t_pred=t_pred)
r=rows.cpu().numpy(); rr=rr[:,:r.shape[1]].cpu().numpy()
d=np.abs(r-rr)
print("max diff per row:", d.max(axis=(1,2)))
print("max diff per column:", d.max(axis=(0,1)))
