"""Debug aid (library built with EXTRA=-DESKF_X_EPDUMP, selected with DVI_ESKF_LIB): the Jacobian records of the long-horizon
kernel against those the epoch-by-epoch path leaves in its workspace.  usage: dbg_epochs_records.py B dtype n"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd.workloads import monte_carlo_pairs
from dvi_ekf_amd import _capi

B = int(sys.argv[1]); dtype = torch.float64 if sys.argv[2] == "f64" else torch.float32
n = int(sys.argv[3])
dev = torch.device("cuda", 0)
H = 8
wl = monte_carlo_pairs(B, horizon=H, seed=4, device=dev, dtype=dtype)
f = wl.make_filter(); snap = f.clone()
idx = torch.arange(n, device=dev) % H
om, acc = wl.om[idx].contiguous(), wl.acc[idx].contiguous()
dt = torch.full((n, B), 0.1, device=dev, dtype=dtype)
pos, quat, notch = wl.pos[:1].contiguous(), wl.quat[:1].contiguous(), wl.notch[:1].contiguous()
f.epoch(dt, om, acc, pos[0], quat[0], notch[0])
torch.cuda.synchronize()
lib = _capi.lib()
lib.dvi_eskf_x_raw_J.restype = C.c_void_p; lib.dvi_eskf_x_raw_J.argtypes = [C.c_void_p]
lib.dvi_eskf_x_ep_buffer.restype = C.c_int; lib.dvi_eskf_x_ep_buffer.argtypes = [C.c_void_p]
pJ = lib.dvi_eskf_x_raw_J(f._h)
es = 8 if dtype == torch.float64 else 4
Jold = np.empty((n, B, 96), dtype=np.float64 if es == 8 else np.float32)
import torch.cuda
# copy the workspace out through a tensor alias
buf = torch.empty(n * B * 96, dtype=dtype, device=dev)
C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(pJ), C.c_size_t(n * B * 96 * es), C.c_int(3))
Jold = buf.cpu().numpy().reshape(n, B, 96)
dump = torch.zeros(n * B * 67, dtype=dtype, device=dev)
lib.dvi_eskf_x_ep_buffer(C.c_void_p(dump.data_ptr()))
g = snap.clone()
g.run_epochs([n], dt, om, acc, pos, quat, notch)
torch.cuda.synchronize()
Jnew = dump.cpu().numpy().reshape(n, B, 67)
names = [("A", 0, 9), ("Th", 9, 18), ("Bm", 18, 27), ("Cm", 27, 45), ("Dm", 45, 57), ("Em", 57, 66), ("dt", 66, 67)]
for k in range(n):
    for nm, a, b in names:
        d = np.abs(Jold[k, :, 8 + a:8 + b] - Jnew[k, :, a:b]).max()
        print("predict", k, nm, "max diff", d, "old", Jold[k, 0, 8 + a:8 + min(b, a + 3)], "new", Jnew[k, 0, a:min(b, a + 3)])
