"""Experiment: how well does a thin latency-bound kernel (tools/exp/dummy_chain.hip: the shape a register-lean
pre kernel would have) overlap with the covariance kernel when the two run on different streams?"""
import ctypes as C, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvi_ekf_amd.workloads import monte_carlo_pairs
so = "/tmp/libdummy.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/exp/dummy_chain.hip")])
lib = C.CDLL(so)
dev = torch.device("cuda:0")
B, H = 32768, 16
wl = monte_carlo_pairs(B, horizon=H, seed=4, device=dev)
f = wl.make_filter(); snap = f.clone(); st = torch.zeros(B, dtype=torch.int32, device=dev)
x = torch.rand(65536, dtype=torch.float64, device=dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def filt():
    f.copy_state_from(snap)
    for k in range(H):
        f.step(wl.dt[k], wl.om[k], wl.acc[k], wl.pos[k], wl.quat[k], wl.notch[k], status=st)
def dummy(n):
    for _ in range(n):
        lib.launch_dummy(C.c_void_p(x.data_ptr()), 65536, iters, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def graph_of(fn):
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    torch.cuda.synchronize()
    return g
def both():
    main = torch.cuda.current_stream(dev)
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1):
        filt()
    with torch.cuda.stream(s2):
        dummy(H)
    main.wait_stream(s1); main.wait_stream(s2)
gf, gd, gb = graph_of(filt), graph_of(lambda: dummy(H)), graph_of(both)
tf, td, tb = timed(gf.replay), timed(gd.replay), timed(gb.replay)
print(f"half batch (B={B}) {H} pair-steps alone: {tf/H*1e6:.1f} us/step; {H} dummy kernels (1024 waves, {iters} sincos each) alone: {td/H*1e6:.1f} us each; "
      f"both on two streams: {tb/H*1e6:.1f} us per (step + dummy)  [serial would be {(tf+td)/H*1e6:.1f}]")
