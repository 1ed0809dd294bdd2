"""Experiment: the batch split into S independent sub-batches on S streams (fill/drain of one
sub-batch's kernels overlaps the steady state of the others).  usage: exp_split.py [S ...]"""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dvi_ekf_amd.workloads import monte_carlo_pairs

import os
B, H, K = int(os.environ.get("EXP_B", 65536)), 16, 1024
dev = torch.device("cuda:0")


def run(S):
    sub = B // S
    wls = [monte_carlo_pairs(sub, horizon=H, seed=4, device=dev, first_instance=i * sub) for i in range(S)]
    fs = [w.make_filter() for w in wls]
    snaps = [f.clone() for f in fs]
    sts = [torch.zeros(sub, dtype=torch.int32, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

    def horizon():
        main = torch.cuda.current_stream(dev)
        for i in range(S):
            streams[i].wait_stream(main)
            with torch.cuda.stream(streams[i]):
                fs[i].copy_state_from(snaps[i])
                w = wls[i]
                for k in range(H):
                    fs[i].step(w.dt[k], w.om[k], w.acc[k], w.pos[k], w.quat[k], w.notch[k], status=sts[i])
        for i in range(S):
            main.wait_stream(streams[i])

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        horizon(); horizon()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        horizon()
    torch.cuda.synchronize()
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // H):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    ok = all(int(s.abs().max()) == 0 for s in sts)
    print(f"S={S}: {B * K / t / 1e8:.3f}e8 pair-steps/s, {t / K * 1e3:.4f} ms/step, valid={ok}", flush=True)


for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 1, 2, 4]:
    run(S)
