import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from dvi_ekf_amd import Config, Simulator
root="/root/repo"
cfg = Config(os.path.join(root, "config.yaml"), base_dir=root)
cfg.golden_compat = True; cfg.max_vals = None; cfg.batch = 65536
sim = Simulator(cfg)
kf = sim.kf
def T(label, fn):
    torch.cuda.synchronize(); t0=time.perf_counter(); r=fn(); torch.cuda.synchronize(); print("%-28s %.2f ms"%(label,(time.perf_counter()-t0)*1e3)); return r
# replicate: build args via the filter's own prep by monkeypatching run_epochs
calls=[]
orig=type(kf._dev).run_epochs
def spy(self,*a,**k):
    calls.append((self,a,k)); return orig(self,*a,**k)
type(kf._dev).run_epochs=spy
T("Filter.run total", lambda: kf.run(sim.camera,0,""))
type(kf._dev).run_epochs=orig
print(kf.last_run_timing)
for (obj,a,k) in calls:
    for rep in range(2):
        T("run_epochs B=%d rows=%s (rep %d)"%(obj.batch, "t_pred" in k, rep), lambda: orig(obj,*a,**k))
T("log filter create+copy", lambda: kf._log_filter())
kf._log_dev=None
T("log filter create+copy (again)", lambda: kf._log_filter())
st=torch.zeros((138,65536),dtype=torch.int32,device="cuda")
T("status[:,0].cpu()", lambda: st[:,0].cpu().numpy())
T("zeros status", lambda: torch.zeros((138,65536),dtype=torch.int32,device="cuda"))
