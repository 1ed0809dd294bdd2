"""diagnostic (library built with -DESKF_X_STAMPS): where a block of eskf_pre4_kernel spends its life.  Runs the bench
workload for a few steps and prints, per stamp, the median / min / max over the blocks of the last launch, in
microseconds after the first block's start."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dvi_ekf_amd import _capi
from dvi_ekf_amd.workloads import monte_carlo_pairs
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = monte_carlo_pairs(B, horizon=8, seed=4, device=dev)
f = wl.make_filter()
for rep in range(3):
    for k in range(8):
        f.step(wl.dt[k], wl.om[k], wl.acc[k], wl.pos[k], wl.quat[k], wl.notch[k])
torch.cuda.synchronize()
L = _capi.lib()
buf = np.zeros(8 * 4096, dtype=np.uint64)
fn = L.dvi_eskf_x_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
nb = min(4096, (B + 63) // 64)
st = buf.reshape(8, 4096)[:, :nb].astype(np.int64)
t0 = st[0].min()
names = ["start", "loads arrived", "phase A done", "past barrier 1", "phase B done", "past barrier 2", "stores issued", "stores done"]
for k in range(8):
    v = (st[k] - t0) / 100.0
    print(f"{names[k]:16s} median {np.median(v):7.2f}  min {v.min():7.2f}  max {v.max():7.2f}  p10 {np.percentile(v,10):7.2f} p90 {np.percentile(v,90):7.2f} us")

buf = np.zeros(8 * 4096, dtype=np.uint64)
fn = L.dvi_eskf_x_cstamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data) == 0
nc = min(4096, (B // 2 + 7) // 8)
st = buf.reshape(8, 4096)[:, :nc].astype(np.int64)
t0 = st[0].min()
print("covariance kernel (every eighth wavefront): per-wavefront durations of the phases, us")
names = ["load (rows + record)", "predict", "update", "reset", "store issue", "store ack"]
for k in range(6):
    v = (st[k + 1] - st[k]) / 100.0
    print(f"{names[k]:22s} median {np.median(v):6.2f}  p10 {np.percentile(v,10):6.2f}  p90 {np.percentile(v,90):6.2f}")
life = (st[6] - st[0]) / 100.0
print(f"wavefront life          median {np.median(life):6.2f}  p10 {np.percentile(life,10):6.2f}  p90 {np.percentile(life,90):6.2f};  kernel span {(st[6].max() - t0) / 100.0:.1f} us; starts: median {np.median((st[0]-t0)/100.0):.1f}")
