"""diagnostic (GPU box, library built with EXTRA=-DESKF_X_RUNSTAMPS): where a block of eskf_run_kernel spends a step.
Replays one horizon of the bench workload as one run_epochs_from call and reads the 100 MHz stamps of every block:
step start | records complete | hand-over done | rounds 0..3 of wavefront 0 | rounds 0..3 of wavefront 4 | end-of-step barrier.
usage: DVI_ESKF_LIB=variants_tmp/lib_runstamps.so python tools/exp_run_stamps.py [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd import _capi
from dvi_ekf_amd.workloads import monte_carlo_pairs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
H = 16
wl = monte_carlo_pairs(B, horizon=H, seed=4, device=dev)
f = wl.make_filter(); snap = f.clone()
ones = np.ones(H, dtype=np.int32)
for _ in range(20):
    f.run_epochs(ones, wl.dt, wl.om, wl.acc, wl.pos, wl.quat, wl.notch, src=snap)
torch.cuda.synchronize()
L = _capi.lib()
nb = min(1024, (B + 63) // 64)
buf = np.zeros(1024 * 16 * 12, dtype=np.uint64)
L.dvi_eskf_x_run_stamps.argtypes = [C.c_void_p]
assert L.dvi_eskf_x_run_stamps(buf.ctypes.data) == 0
s = buf.reshape(1024, 16, 12)[:nb].astype(np.float64) / 100.0      # us
t0 = s[:, 0, 0].min()
print("blocks", nb, "first-step start spread (us): min 0 max %.1f; blocks starting within 5 us: %d" % (s[:, 0, 0].max() - t0, int((s[:, 0, 0] - t0 < 5).sum())))
d = lambda a, b: np.median(s[:, :, a] - s[:, :, b])
print("per step, median over blocks and steps (us): pre phase %.2f | hand-over %.2f | wave 0 rounds %s | wave 4 rounds %s | wave0 last round -> barrier %.2f | step %.2f"
      % (d(1, 0), d(2, 1), ["%.2f" % d(3 + r, 2 + r) for r in range(4)], ["%.2f" % (np.median(s[:, :, 7] - s[:, :, 2]) if r == 0 else d(7 + r, 6 + r)) for r in range(4)], d(11, 6), np.median(s[:, 1:, 0] - s[:, :-1, 0])))
first = s[s[:, 0, 0] - t0 < 5]
print("total run of a first-generation block: %.1f us; whole launch %.1f us" % (np.median(first[:, 15, 11] - first[:, 0, 0]), s[:, 15, 11].max() - t0))
for k in (0, 1, 8):
    print("step %d: pre %.2f, rounds(w0) %s" % (k, np.median(s[:, k, 1] - s[:, k, 0]), ["%.2f" % np.median(s[:, k, 3 + r] - s[:, k, 2 + r]) for r in range(4)]))

# the pre phase of every block's last step from the inside (wavefront 0: position / velocity / attitude chain)
pb = np.zeros(1024 * 8, dtype=np.uint64)
L.dvi_eskf_x_pre_stamps.argtypes = [C.c_void_p]
if L.dvi_eskf_x_pre_stamps(pb.ctypes.data) == 0:
    q = pb.reshape(1024, 8)[:nb].astype(np.float64) / 100.0
    m = lambda a, b: np.median(q[:, a] - q[:, b])
    print("pre phase, wavefront 0, last step (us): loads arrived %.2f | phase A %.2f | barrier 1 %.2f | 1b + phase B + state stores %.2f | barrier 2 %.2f"
          % (m(1, 0), m(2, 1), m(3, 2), m(4, 3), m(5, 4)))
