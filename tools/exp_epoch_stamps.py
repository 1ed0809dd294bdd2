"""Where a frame of the long-horizon kernel goes: per wavefront role the cycles of the whole run and the cycles spent
waiting, by cause.  Library built with `make -C dvi_ekf_amd/csrc -B EXTRA=-DESKF_X_EPSTAMPS OUT=../../variants_tmp/lib_epstamps.so`
and selected with DVI_ESKF_LIB.  usage: exp_epoch_stamps.py [batch] [f32|f64] [n_steps] [imu_hz]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvi_ekf_amd import _capi
from dvi_ekf_amd.workloads import long_horizon

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dtype = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
imu_hz = float(sys.argv[4]) if len(sys.argv) > 4 else 1000.0
dev = torch.device("cuda", 0)
wl = long_horizon(batch=B, n_steps=n_steps, imu_hz=imu_hz)
om, acc = wl.synthesize_imu_on_device(dtype=dtype, device=dev)
om, acc = om[1:].contiguous(), acc[1:].contiguous()
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)
dt, pos, quat, notch = t(wl.dt), t(wl.pos), t(wl.quat), t(wl.notch)
f = wl.make_filter(dtype=dtype, device=dev)
snap = f.clone()
lib = _capi.lib()
lib.dvi_eskf_x_ep_buffer.restype = C.c_int
lib.dvi_eskf_x_ep_buffer.argtypes = [C.c_void_p]
nblk_max = (B + 3) // 4
buf = torch.zeros(nblk_max * 16 * 8, dtype=torch.int64, device=dev)
lib.dvi_eskf_x_ep_buffer(C.c_void_p(buf.data_ptr()))
times = []
for _ in range(3):
    f.copy_state_from(snap)
    buf.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f.run_epochs(wl.n_pred, dt, om, acc, pos, quat, notch)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) * 1e-3)
s = buf.cpu().numpy().reshape(nblk_max, 16, 8)
used = s[:, 0, 0] > 0
nblk = int(used.sum())
E = len(wl.n_pred)
tt = min(times[1:])
out = {"batch": B, "dtype": str(dtype), "imu_steps": wl.n_steps, "frames": E, "seconds": tt, "us_per_frame": tt / E * 1e6,
       "imu_steps_per_s": B * wl.n_steps / tt, "blocks": nblk}
# (s_memrealtime at the end of every wavefront: 100 MHz; first to last = the launch)
ends = s[used][:, :5, 5].astype(np.float64)
out["launch_us_by_stamps"] = float(ends.max() - ends[ends > 0].min()) / 100.0
clk = 100e6 if False else None
names = {0: ("RA", ["update", "aux room"]), 1: ("RB", ["update", "aux room", "RA constants", "RA p_cam"]), 2: ("JA", ["aux entry", "ring room"]),
         3: ("JB", ["aux entry", "ring room"]), 4: ("C0", ["record", "residual"])}
# s_memtime ticks per microsecond: the consumer's total against the kernel's duration (one generation of blocks: a block lives for the launch)
ticks_per_us = float(np.median(s[used, 4, 0])) / (tt * 1e6) if nblk <= 512 else None
out["memtime_ticks_per_us_estimate"] = ticks_per_us
for w, (nm, slots) in names.items():
    tot = float(np.median(s[used, w, 0]))
    row = {"total_frac_of_C0": tot / float(np.median(s[used, 4, 0]))}
    waits = 0.0
    for k, sl in enumerate(slots):
        wv = float(np.median(s[used, w, 1 + k]))
        row["wait " + sl + " frac"] = wv / tot
        waits += wv
    row["busy frac"] = 1.0 - waits / tot
    if ticks_per_us:
        row["busy_us_per_frame"] = (tot - waits) / ticks_per_us / E
    out[nm] = row
if os.environ.get("STAMPS_BRIEF"):
    print("B %d %s: %.1f us/frame, %.3e IMU steps/s, %d blocks" % (B, str(dtype)[-7:], out["us_per_frame"], out["imu_steps_per_s"], nblk))
    for nm in ("RA", "RB", "JA", "JB", "C0"):
        r = out[nm]
        print("  %s busy %.2f (%.1f us/frame) | " % (nm, r["busy frac"], r.get("busy_us_per_frame", float("nan"))) +
              ", ".join("%s %.2f" % (k[5:-5], v) for k, v in r.items() if k.startswith("wait ")))
else:
    print(json.dumps(out, indent=1))
