"""Secondary measurements of SURVEY 8(d) that are not the bench line:
  config #3  predict-only, B = 1 048 576, fp32, 256 consecutive predict launches (P round-trips HBM)
  10:1       v1-literal schedule: epochs of 10 predicts (dt = 0.1) + 1 update in one launch, fp64
Prints one JSON object per measurement.  usage: bench_configs.py [config3] [epoch10] [--batch3 N]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def _time(fn, n, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def config3(B):
    from dvi_ekf_amd.batch import BatchedESKF
    from dvi_ekf_amd.workloads import default_run_inputs
    cfg, ri, Q, R, sa, so = default_run_inputs(golden_compat=True, all_frames=False, frozen=(0,) * 6)
    dev = torch.device("cuda:0")
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=(0,) * 6, sigma_a=sa, sigma_om=so,
                    hold_frozen=True, dtype=torch.float32, device=dev)
    f.set_noise(Q, R)
    f.set_state(x=np.tile(ri.x0, (B, 1)), P=ri.P0, om_old=np.tile(ri.om0, (B, 1)), acc_old=np.tile(ri.acc0, (B, 1)))
    from dvi_ekf_amd import noise
    z = noise.randn(3 << 32, 0, B, 0, 1, 6, dtype=torch.float32, device=dev)[0]     # seed 3
    om = (0.05 * z[:, 0:3]).contiguous()
    acc = (1.0 * z[:, 3:6]).contiguous()
    dt = torch.full((B,), 1e-3, device=dev, dtype=torch.float32)
    t = _time(lambda: f.predict(dt, om, acc), 256)
    ok = bool(torch.isfinite(f.x).all())
    bytes_per = 2716
    return {"config": "#3 predict-only fp32", "batch": B, "launches": 256, "ms_per_launch": t * 1e3,
            "value": B / t, "unit": "predict-steps/s", "algorithmic_GBps": B * bytes_per / t / 1e9,
            "frac_of_8TBps": B * bytes_per / t / 8e12, "finite": ok}


def epoch10(B=65536, N=10):
    from dvi_ekf_amd.workloads import monte_carlo_pairs
    dev = torch.device("cuda:0")
    wl = monte_carlo_pairs(B, horizon=4, seed=4, device=dev)
    f = wl.make_filter()
    snap = f.clone()
    dt = torch.full((N,), 0.1, device=dev, dtype=torch.float64)
    om = wl.om[0].unsqueeze(0).repeat(N, 1, 1).contiguous()
    acc = wl.acc[0].unsqueeze(0).repeat(N, 1, 1).contiguous()
    k = [0]

    def step():
        if k[0] % 4 == 0:
            f.copy_state_from(snap)
        f.epoch(dt, om, acc, wl.pos[k[0] % 4], wl.quat[k[0] % 4], wl.notch[k[0] % 4])
        k[0] += 1

    t = _time(step, 32, warm=4)
    bytes_per = 5440 + 56 * N
    return {"config": "v1-literal 10:1 epochs, fp64", "batch": B, "n_pred": N, "ms_per_epoch": t * 1e3, "value": B / t,
            "unit": "epochs/s", "predict_steps_per_s": B * N / t, "algorithmic_GBps": B * bytes_per / t / 1e9,
            "frac_of_8TBps": B * bytes_per / t / 8e12, "state_restore_every": 4}


def config5(B=8192, frames=64):
    """Config #5 shape: 1 kHz IMU / 30 Hz vision, fp32 -- one launch per camera frame = 33/33/34 predicts
    (dt = 1e-3) + 1 update, P resident on chip across the predicts.  Synthetic IMU samples (throughput only)."""
    from dvi_ekf_amd.workloads import monte_carlo_pairs
    dev = torch.device("cuda:0")
    wl = monte_carlo_pairs(B, horizon=4, seed=5, device=dev)
    f = wl.make_filter(dtype=torch.float32)
    snap = f.clone()
    from dvi_ekf_amd import noise
    counts = [33, 33, 34]
    dts = [torch.full((n,), 1e-3, device=dev, dtype=torch.float32) for n in counts]
    z = noise.randn(5 << 32, 0, 1, 0, sum(counts), 6, dtype=torch.float32, device=dev)[:, 0]   # one shared IMU stream
    zs = torch.split(z, counts)
    oms = [(0.05 * c[:, 0:3]).contiguous() for c in zs]
    accs = [(1.0 * c[:, 3:6]).contiguous() for c in zs]
    pos, quat, notch = wl.pos[0].float(), wl.quat[0].float(), wl.notch[0].float()
    k = [0]

    def frame():
        if k[0] % 8 == 0:
            f.copy_state_from(snap)
        i = k[0] % 3
        f.epoch(dts[i], oms[i], accs[i], pos, quat, notch)
        k[0] += 1

    t = _time(frame, frames, warm=6)
    steps_per_frame = 100.0 / 3
    return {"config": "#5 shape: 1 kHz IMU / 30 Hz vision, fp32, one launch per frame", "batch": B, "ms_per_frame": t * 1e3,
            "imu_steps_per_s": B * steps_per_frame / t, "frames_per_s": B / t, "unit": "instance-steps/s",
            "finite": bool(torch.isfinite(f.x).all())}


if __name__ == "__main__":
    what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["config3", "epoch10", "config5"]
    B3 = 1048576
    if "--batch3" in sys.argv:
        B3 = int(sys.argv[sys.argv.index("--batch3") + 1])
    if "config3" in what:
        print(json.dumps(config3(B3)))
    if "epoch10" in what:
        print(json.dumps(epoch10()))
    if "config5" in what:
        print(json.dumps(config5()))
