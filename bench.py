#!/usr/bin/env python3
"""bench.py -- ESKF predict+update pair-steps/s on MI355X (BASELINE.json metric).

One "step" = one predict+update pair of every instance of the rank's batch (B = 65 536 filter
instances per GPU, fp64, 26-state / 24 error-state), inputs and state resident in HBM.  By default a
horizon of steps is ONE call of dvi_eskf_run_epochs_from = one launch of eskf_run_kernel (the time loop of
Filter.run in the kernel): a block steps its own 64 instances, and the covariances of the sixteen it works on
last in a step -- first in the next -- stay in LDS in between; those of the other 48 round-trip memory every
step (so the measured traffic of a step, `roofline.traffic`, is BELOW the algorithmic bytes `roofline.achieved`
is computed from: a quarter of the covariance traffic is not there).  `--loop-in-kernel 0` enqueues one
dvi_eskf_step call per step instead (one launch per step; every covariance round-trips memory once per step).
Workload: BASELINE config #2/#4 -- the reference's default run (mandala0_mono, default
config.yaml) on the 1:1 schedule (dt = 1.0), per-instance perturbed initial extrinsic
DOFs (unfrozen) and per-instance Gaussian noise on IMU and camera measurements.  The
reference filter is dynamically unstable on long horizons, so the state is
restored from a device snapshot every `--horizon` steps; the restore is INSIDE the timed
region (the first step of a horizon reads the snapshot handle).

Single GPU:  python bench.py --steps K --warmup W
N GPUs:      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                 --master-port P bench.py --gpus N --steps K --warmup W
             or, from a plain shell, `python bench.py --gpus N ...`: without WORLD_SIZE in the environment the
             script starts that launcher itself as a child process and leaves with its exit code
Ranks own disjoint instance blocks; the only collective is the final RCCL all-gather of the
calibration estimates, inside the timed region.  Default: weak scaling, 65 536 instances per GPU
(what the driver's scaling run expects).  `--strong` is BASELINE config #4 as written: ONE batch of
`--batch` instances sharded over the ranks (65 536 -> 8 x 8 192); thanks to the counter-based noise
generator both modes compute, for every global instance, exactly what a single GPU computes for it
(tests/test_gpu_00_two_rank_rehearsal.py).

Timing: W warm-up steps, graph capture and `--sustain-seconds` of untimed replay (clocks), then a
barrier + synchronize, EXACTLY K steps replayed from HIP graphs + the calibration read (+ the
gather), synchronize + barrier; MAX over ranks.  `roofline.frac` is from HIP events around the K
steps on the launch stream, `roofline.frac_wall` from the host clock (ms_per_step);
`roofline.streaming` repeats the measurement at a batch that does not fit the Infinity Cache;
`cpu_baseline` times the C oracle on the host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_PAIR_F64 = 5496      # 2*300w + 2*36w + 7w + 8w, w = 8
LOOP_IN_KERNEL_DEFAULT = True   # see --loop-in-kernel
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(wl, seconds_target=12.0, threads=None):
    """Times the C oracle (oracle/eskf_oracle.c, OpenMP over instances) on a bounded sample
    of the same workload: the first n instances, the first `horizon` pair-steps.  threads: OpenMP threads
    (None = all host cores)."""
    from oracle import c_oracle
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, True, wl.sigma_a,
                               wl.sigma_om, wl.Q, wl.R)
    all_cores = c_oracle.num_threads()
    cores = c_oracle.set_num_threads(threads) if threads else all_cores
    try:
        return _cpu_baseline_run(c_oracle, prm, wl, seconds_target, cores)
    finally:
        if threads:
            c_oracle.set_num_threads(all_cores)


def _cpu_baseline_run(c_oracle, prm, wl, seconds_target, cores):
    H = wl.horizon

    def run(n, steps):
        o = c_oracle.OracleBatch(prm, wl.x0[:n].cpu().numpy(), wl.P0.cpu().numpy(), wl.om0.cpu().numpy(),
                                 wl.acc0.cpu().numpy())
        host = [(wl.dt[k, :n].cpu().numpy(), wl.om[k, :n].cpu().numpy(), wl.acc[k, :n].cpu().numpy(),
                 wl.pos[k, :n].cpu().numpy(), wl.quat[k, :n].cpu().numpy(), wl.notch[k, :n].cpu().numpy())
                for k in range(steps)]
        t0 = time.perf_counter()
        for dt, om, acc, pos, quat, notch in host:
            o.predict(dt, om, acc)
            o.update(pos, quat, notch)
        return time.perf_counter() - t0

    n_probe = min(64 * cores, wl.x0.shape[0])
    t_probe = run(n_probe, 2)
    rate = n_probe * 2 / max(t_probe, 1e-9)
    steps = min(H, 8)
    n = int(min(wl.x0.shape[0], max(cores, rate * seconds_target / steps)))
    t = run(n, steps)
    return {"value": n * steps / t, "unit": "pair-steps/s", "cores": cores, "kind": "port",
            "sample": f"C oracle (oracle/eskf_oracle.c, OpenMP), first {n} instances x {steps} pair-steps of the "
                      f"same workload, {t:.1f} s"}


# ---- the `secondary` block: BASELINE's other configurations in the driver-run line ------------------------------------
# algorithmic bytes and floating-point operations of the covariance arithmetic as the kernels perform it
# (structure of Fx exploited; counted from the instruction lists of fx_apply_store_dpp / fx_apply_col_dpp / the update
# of cov_rows_body, per instance: 24 rows x (74 + 71 FMA + 10 adds) + 13 noise adds per predict; 7 measurements x
# (24 row FMAs + pivot, gain and error-estimate arithmetic) + the reset per update).  The lane-per-instance chains
# (nominal state, Jacobian blocks: ~1.2k operations per predict) are not counted.
BYTES_PREDICT = {"f64": 5432, "f32": 2716}
FLOP_PER_PREDICT = 7213
FLOP_PER_UPDATE = 12492
VALU_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}      # MI355X_MICROARCH.md: vector fp64 78.6 TF, fp32 157.3 TF -- the rate of a plain v_fmac_f32
                                                    # (measured, tools/pk_micro.hip -> profiles/r05_packed_fp32_micro.txt: 128 multiply-adds per clock and CU,
                                                    # with or without the DPP operand; v_pk_fma_f32 doubles it only with both operands in the lane)


def _valu_counters():
    """profiles/valu_counters.json: vector-ALU issue share per regime from the committed rocprofv3 SQ-counter passes
    (tools/prof_round.sh, tools/summarize_sq.py) -- constants of the committed profile, NOT measured in this run."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_counters.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def _valu_issue(key):
    c = _valu_counters()
    e = c.get(key)
    if not e:
        return None
    return {"valu_busy": e["valu_busy"], "formula": c.get("formula"), "source": "static: profiles/valu_counters.json <- " + str(c.get("source")),
            "kernels": [{k: kk[k] for k in ("kernel", "launches", "valu_busy", "SQ_INSTS_VALU_per_wave")} for kk in e.get("kernels", [])]}


def _flop_roofline(predicts_per_s, updates_per_s, dtype, counters_key=None):
    tf = (predicts_per_s * FLOP_PER_PREDICT + updates_per_s * FLOP_PER_UPDATE) / 1e12
    return {"bound": "valu", "achieved": tf, "peak": VALU_PEAK_TFLOPS[dtype], "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS[dtype],
            "flop_per_predict": FLOP_PER_PREDICT, "flop_per_update": FLOP_PER_UPDATE,
            "valu_issue": _valu_issue(counters_key) if counters_key else None,
            "note": "frac: covariance arithmetic only, as performed (Fx block structure exploited), against the vector peak of the dtype; "
                    "resident regime: P stays on chip across the predicts, so no HBM fraction is claimed (SURVEY 8(d)).  valu_issue: the "
                    "share of the chip's vector-ALU issue time the regime's kernels used, by the SQ counters of the committed profile -- "
                    "the rest of the distance to `frac` is instructions that are not multiply-adds of the covariance (lane exchanges, "
                    "addressing, the lane-per-instance chains) and idle lanes (24 of 32 per instance)"}


def secondary_block(dev, measure_pairs):
    """Configs #3 and #5, the reference's 10:1 schedule and one GPU's share of config #4 at 8 GPUs, each with its formula
    and its fraction of the roof that bounds it -- measured here so that they are in the line the driver records, not in a
    file of the builder's.  Bounded: about a minute and a half on top of the headline measurement."""
    from dvi_ekf_amd.workloads import long_horizon, monte_carlo_pairs
    from dvi_ekf_amd.batch import BatchedESKF
    from dvi_ekf_amd.workloads import default_run_inputs
    from dvi_ekf_amd import noise
    out = {}
    t_start = time.perf_counter()

    def events(fn, n, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / n

    # -- config #3: predict-only, B = 1 048 576, fp32, 256 consecutive launches (P round-trips HBM every launch)
    B3 = 1048576
    cfg, ri, Q, R, sa, so = default_run_inputs(golden_compat=True, all_frames=False, frozen=(0,) * 6)
    f = BatchedESKF(B3, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=(0,) * 6, sigma_a=sa, sigma_om=so, hold_frozen=True,
                    dtype=torch.float32, device=dev)
    f.set_noise(Q, R)
    f.set_state(x=np.tile(ri.x0, (B3, 1)), P=ri.P0, om_old=np.tile(ri.om0, (B3, 1)), acc_old=np.tile(ri.acc0, (B3, 1)))
    z = noise.randn(3 << 32, 0, B3, 0, 1, 6, dtype=torch.float32, device=dev)[0]                # seed 3
    om3, acc3 = (0.05 * z[:, 0:3]).contiguous(), (1.0 * z[:, 3:6]).contiguous()
    dt3 = torch.full((B3,), 1e-3, device=dev, dtype=torch.float32)
    t3 = events(lambda: f.predict(dt3, om3, acc3), 256, warm=8)
    out["config3_predict_only_fp32"] = {
        "batch": B3, "launches": 256, "ms_per_launch": t3 * 1e3, "value": B3 / t3, "unit": "predict-steps/s",
        "roofline": {"bound": "hbm", "achieved": B3 * BYTES_PREDICT["f32"] / t3 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": B3 * BYTES_PREDICT["f32"] / t3 / 1e9 / HBM_PEAK_GBS, "bytes_per_predict": BYTES_PREDICT["f32"],
                     "formula": "2*300w + 2*36w + 7w, w = 4"},
        "finite": bool(torch.isfinite(f.x).all())}
    del f, z, om3, acc3, dt3
    torch.cuda.empty_cache()

    # -- the reference's own 10:1 schedule (do_fast_sim: false), fp64, B = 65 536: ten predicts (dt = 0.1) + one update per epoch
    B10, N10 = 65536, 10
    wl = monte_carlo_pairs(B10, horizon=4, seed=4, device=dev)
    f = wl.make_filter()
    snap = f.clone()
    dt10 = torch.full((N10,), 0.1, device=dev, dtype=torch.float64)
    om10 = wl.om[0].unsqueeze(0).repeat(N10, 1, 1).contiguous()
    acc10 = wl.acc[0].unsqueeze(0).repeat(N10, 1, 1).contiguous()
    k = [0]

    def one_epoch():
        f.epoch(dt10, om10, acc10, wl.pos[k[0] % 4], wl.quat[k[0] % 4], wl.notch[k[0] % 4])
        k[0] += 1
    best = None
    for _ in range(2):
        f.copy_state_from(snap)
        k[0] = 0
        t = events(one_epoch, 24, warm=0)
        best = t if best is None else min(best, t)
    byt = 5440 + 56 * N10
    out["epochs_10to1_fp64"] = {
        "batch": B10, "predicts_per_epoch": N10, "ms_per_epoch": best * 1e3, "value": B10 / best, "unit": "epochs/s",
        "predict_steps_per_s": B10 * N10 / best, "launch_pattern": "one dvi_eskf_epoch call per epoch",
        "roofline": _flop_roofline(B10 * N10 / best, B10 / best, "f64", "epochs_10to1_fp64"),
        "hbm_epoch_formula": {"bytes_per_epoch": byt, "formula": "2*300w + 2*36w + 7N*w + 8w, w = 8, N = 10", "achieved": B10 * byt / best / 1e9,
                              "unit": "GB/s", "frac": B10 * byt / best / 1e9 / HBM_PEAK_GBS},
        "finite": bool(torch.isfinite(f.x).all())}
    del f, snap, wl, om10, acc10
    torch.cuda.empty_cache()

    # -- config #5: 1 kHz IMU / 30 Hz vision, 10 000 IMU steps, 33 / 33 / 34 predicts per update; B = 8 192 as specified and
    #    B = 1 024, one GPU's share of it on 8 GPUs; the whole run is ONE dvi_eskf_run_epochs call
    wl5 = long_horizon(batch=8192)
    for name, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        om, acc = wl5.synthesize_imu_on_device(dtype=dtype, device=dev)
        om, acc = om[1:].contiguous(), acc[1:].contiguous()
        tt = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)
        dt5, pos5, quat5, notch5 = tt(wl5.dt), tt(wl5.pos), tt(wl5.quat), tt(wl5.notch)
        for B5 in (8192, 1024):
            f = wl5.make_filter(dtype=dtype, device=dev, batch=B5)
            snap = f.clone()
            times, st = [], None
            for _ in range(3):
                f.copy_state_from(snap)
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                st, _ = f.run_epochs(wl5.n_pred, dt5, om, acc, pos5, quat5, notch5)
                e1.record()
                torch.cuda.synchronize(dev)
                times.append(e0.elapsed_time(e1) * 1e-3)
            t5 = min(times[1:])
            E5 = len(wl5.n_pred)
            out["config5_%s_B%d" % (name, B5)] = {
                "batch": B5, "imu_steps": wl5.n_steps, "updates": E5, "seconds_per_run": t5, "us_per_frame": t5 / E5 * 1e6,
                "value": B5 * wl5.n_steps / t5, "unit": "IMU steps/s (instance-steps)",
                "launch_pattern": "one dvi_eskf_run_epochs call for the whole run",
                "regime": "camera noise 0.2 / 1 deg, frozen extrinsics (with the default YAML noise the reference filter stays bounded for ~800 steps only)",
                "roofline": _flop_roofline(B5 * wl5.n_steps / t5, B5 * E5 / t5, name, "config5_%s_B%d" % (name, B5)),
                "skipped_updates": int((st & 1).sum()), "finite": bool(torch.isfinite(f.x).all())}
            del f, snap
        del om, acc
    del wl5
    torch.cuda.empty_cache()

    # -- pair steps at B = 8 192: one GPU's share of config #4 (65 536 instances sharded over 8 GPUs)
    m8 = measure_pairs(8192)
    ach = BYTES_PER_PAIR_F64 * 8192 / (m8["kern_ms"] * 1e-3) / 1e9
    out["pairs_fp64_B8192"] = {"batch": 8192, "value": 8192 / (m8["kern_ms"] * 1e-3), "unit": "pair-steps/s", "kernel_ms": m8["kern_ms"],
                               "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                            "bytes_per_pair": BYTES_PER_PAIR_F64, "valu_issue": _valu_issue("pairs_fp64_B8192"),
                                            "note": "in a run no covariance leaves the chip between the first and the last step at this batch: "
                                                    "the algorithmic-bytes fraction is a throughput scale, not a traffic claim"},
                               "valid": m8["ok"]}
    out["seconds"] = time.perf_counter() - t_start
    return out


def parity_check(wl, steps=5, n=16):
    """State-trajectory max-abs error of the first n instances against the C oracle over the first
    `steps` pair-steps of the bench workload (outside the timed region; the oracle is the checker)."""
    from oracle import c_oracle
    prm = c_oracle.make_params(wl.cfg.model.length, wl.cfg.model.angle, wl.cfg.frozen_dofs, True, wl.sigma_a,
                               wl.sigma_om, wl.Q, wl.R)
    o = c_oracle.OracleBatch(prm, wl.x0[:n].cpu().numpy(), wl.P0.cpu().numpy(), wl.om0.cpu().numpy(), wl.acc0.cpu().numpy())
    f = wl.make_filter()
    worst0 = worst_rel = 0.0
    for k in range(min(steps, wl.horizon)):
        f.step(wl.dt[k], wl.om[k], wl.acc[k], wl.pos[k], wl.quat[k], wl.notch[k])
        o.predict(wl.dt[k, :n].cpu().numpy(), wl.om[k, :n].cpu().numpy(), wl.acc[k, :n].cpu().numpy())
        o.update(wl.pos[k, :n].cpu().numpy(), wl.quat[k, :n].cpu().numpy(), wl.notch[k, :n].cpu().numpy())
        x = f.x[:n].cpu().numpy()
        worst0 = max(worst0, float(np.abs(x[0] - o.x[0]).max()))
        scale = np.maximum(1.0, np.abs(o.x).max(axis=1, keepdims=True))
        worst_rel = max(worst_rel, float((np.abs(x - o.x) / scale).max()))
    # instance 0 is the unperturbed golden-compat run; the perturbed, unfrozen instances are in the regime
    # where the reference filter itself amplifies rounding differences (|x| grows to 1e2..1e3 within the
    # horizon), hence the error relative to the instance's state magnitude
    return {"instance0_max_abs_state_error_vs_oracle": worst0, "max_state_error_rel_to_state_magnitude": worst_rel,
            "instances": n, "pair_steps": int(min(steps, wl.horizon)), "tolerance": 1e-9}


def cpu_baseline_numpy(wl, steps=16):
    """the NumPy twin of the reference's arithmetic (oracle/eskf_ref.py, complex-step
    Jacobians), ONE instance on ONE core -- the closest analogue of the reference's own CasADi/NumPy path;
    optimistic, since the reference also rebuilds its symbolic graph every step."""
    from oracle.eskf_ref import RefFilter, RefParams
    prm = RefParams(L=wl.cfg.model.length, theta_cam=wl.cfg.model.angle, frozen=tuple(wl.cfg.frozen_dofs), hold_frozen=True,
                    sigma_a=np.asarray(wl.sigma_a, dtype=float), sigma_om=np.asarray(wl.sigma_om, dtype=float),
                    Q=np.asarray(wl.Q, dtype=float), R=np.asarray(wl.R, dtype=float))
    f = RefFilter(prm, wl.x0[0].cpu().numpy(), wl.P0.cpu().numpy(), wl.om0.cpu().numpy(), wl.acc0.cpu().numpy())
    n = min(steps, wl.horizon)
    host = [(float(wl.dt[k, 0]), wl.om[k, 0].cpu().numpy(), wl.acc[k, 0].cpu().numpy(), wl.pos[k, 0].cpu().numpy(),
             wl.quat[k, 0].cpu().numpy(), float(wl.notch[k, 0])) for k in range(n)]
    t0 = time.perf_counter()
    for dt, om, acc, pos, quat, notch in host:
        f.predict(dt, om, acc)
        f.update(pos, quat, notch)
    t = time.perf_counter() - t0
    return {"value": n / t, "unit": "pair-steps/s", "cores": 1, "kind": "port",
            "sample": f"NumPy twin (oracle/eskf_ref.py), instance 0 x {n} pair-steps, {t:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the GPU to reach its sustained clocks (64 timed steps = 9 ms read 4.5e8,
    # 1024 steps = 0.13 s read 5.2e8 on the same box)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=65536, help="filter instances per GPU")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE config #4 as written: --batch is the TOTAL, sharded over the ranks (strong scaling); "
                         "the default keeps --batch instances per GPU (weak scaling, what the driver's scaling run expects)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling only: --batch instances per GPU.  With neither --weak nor --strong a job of N > 1 ranks measures "
                         "BOTH in one run -- `value` is the strong-scaling figure (BASELINE's metric and config #4 read 'batch = 65 536 ... "
                         "sharded'), the weak-scaling one is carried in `weak`")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (configs #3 and #5, the 10:1 schedule, the small batch, the 1-thread oracle)")
    ap.add_argument("--horizon", type=int, default=16, help="pair-steps between state restores")
    ap.add_argument("--lanes", type=int, default=0, help="covariance kernel selector (0 = library default: row-resident, 32 lanes per instance; 2 = row-resident, 24 lanes; 32 = LDS-image kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--restore-copy", action="store_true",
                    help="restart a horizon by copying the snapshot into the filter (176 MB device-to-device) and stepping in "
                         "place, instead of letting the horizon's first step read the snapshot (dvi_eskf_step_from)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from Python instead of replaying a HIP graph")
    ap.add_argument("--per-instance-noise", action="store_true",
                    help="diagnostic: every instance carries its own (identical) noise record -- the tuning-population path")
    ap.add_argument("--sustain-seconds", type=float, default=0.3,
                    help="untimed load (replays of the captured horizon) between the warm-up steps and the timed region, so "
                         "that a short timed region is measured at the GPU's sustained clocks; 0 = none")
    ap.add_argument("--no-streaming", action="store_true", help="skip the secondary measurement at --streaming-batch")
    ap.add_argument("--streaming-batch", type=int, default=262144)
    ap.add_argument("--dump-dofs", default=None, help="rank 0 saves the gathered calibration estimates [B_total,6] (.npy)")
    ap.add_argument("--loop-in-kernel", type=int, default=-1, choices=[-1, 0, 1],
                    help="1: every horizon of the timed region is ONE call of dvi_eskf_run_epochs_from (one predict per "
                         "epoch: one launch of eskf_run_kernel, the time loop in the kernel); 0: one dvi_eskf_step call (one "
                         "or two launches) per step; -1: the default of this build")
    ap.add_argument("--mode", default="pair", choices=["pair", "predict", "update"],
                    help="diagnostic: time only the predict or only the update launch (not the metric)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` from a plain shell: start the N ranks ourselves, one process per GPU, exactly as
        # the driver's launcher would.  A CHILD process (this one has not touched the GPU and never will): rank 0 of
        # the child job prints the JSON line on the stdout it inherits; we leave with the job's exit code.
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if os.environ.get("DVI_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())   # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # (DVI_BENCH_FORCE_DIST=1: take the multi-rank code path -- RCCL communicator, barriers, the final all-gather, the
    #  MAX over ranks -- also with ONE rank; that is how the one-GPU test box exercises the RCCL calls of the N > 1 path)
    if world > 1 or os.environ.get("DVI_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when a communicator is created; stdout of this
        # program is the one JSON line, so the file descriptor points at stderr meanwhile
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if os.environ.get("DVI_BENCH_BACKEND", "nccl") == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier(device_ids=[local_rank])
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from dvi_ekf_amd.workloads import monte_carlo_pairs
    B, H, K, W = args.batch, args.horizon, args.steps, args.warmup
    LOOP_IN_KERNEL = (args.loop_in_kernel == 1) if args.loop_in_kernel >= 0 else LOOP_IN_KERNEL_DEFAULT
    if args.strong and args.weak:
        raise SystemExit("--strong and --weak exclude each other (neither: both are measured when N > 1)")
    # which scaling modes this job measures: the one that was asked for; with N > 1 and no flag both -- first weak
    # (--batch instances per GPU), then strong (--batch instances sharded over the ranks), which becomes `value`
    modes = ["strong"] if args.strong else ["weak"] if (args.weak or world == 1) else ["weak", "strong"]
    if "strong" in modes and args.batch % world:
        raise SystemExit("strong scaling: the total batch must divide evenly over the ranks")

    backend = os.environ.get("DVI_BENCH_BACKEND", "nccl")   # "gloo" only for rehearsing the N>1 path on one GPU

    def barrier():
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    class Run:
        """One filter batch + its snapshot + the captured horizon(s)."""

        def __init__(self, batch, first_instance):
            self.B = batch
            self.wl = monte_carlo_pairs(batch, horizon=H, seed=4, device=dev, first_instance=first_instance)
            self.f = self.wl.make_filter()
            if args.lanes:
                self.f.set_lanes_per_instance(args.lanes)
            if args.per_instance_noise:
                self.f.set_noise_per_instance(np.tile(np.diag(self.wl.Q), (batch, 1)), np.tile(np.diag(self.wl.R), (batch, 1)))
            self.snap = self.f.clone()
            self.status = torch.zeros(batch, dtype=torch.int32, device=dev)
            self.status_run = torch.zeros((H, batch), dtype=torch.int32, device=dev)     # (--loop-in-kernel: per epoch)
            self.ones = np.ones(H, dtype=np.int32)
            self.dofs = torch.empty((batch, 6), dtype=torch.float64, device=dev)
            self.graphs = {}
            self.cal_graph = None

        def pair(self, i):
            """pair-step i of the horizon; the first one starts from the snapshot"""
            wl, f = self.wl, self.f
            if i == 0 and not args.restore_copy:
                f.step_from(self.snap, wl.dt[i], wl.om[i], wl.acc[i], wl.pos[i], wl.quat[i], wl.notch[i], status=self.status)
                return
            if i == 0:
                f.copy_state_from(self.snap)
            f.step(wl.dt[i], wl.om[i], wl.acc[i], wl.pos[i], wl.quat[i], wl.notch[i], status=self.status)

        def eager(self, k0, n):
            if args.mode == "pair" and LOOP_IN_KERNEL:
                # horizon after horizon: steps i .. i + m - 1 of a horizon are ONE call (the first one reads the snapshot)
                k, end = k0, k0 + n
                wl, f = self.wl, self.f
                while k < end:
                    i = k % H
                    m = min(H - i, end - k)
                    f.run_epochs(self.ones[:m], wl.dt[i:i + m], wl.om[i:i + m], wl.acc[i:i + m], wl.pos[i:i + m],
                                 wl.quat[i:i + m], wl.notch[i:i + m], src=self.snap if i == 0 else None,
                                 status=self.status_run[i:i + m])
                    k += m
                return
            for k in range(k0, k0 + n):
                i = k % H
                if args.mode == "pair":
                    self.pair(i)
                else:
                    if i == 0:
                        self.f.copy_state_from(self.snap)
                    if args.mode == "predict":
                        self.f.predict(self.wl.dt[i], self.wl.om[i], self.wl.acc[i])
                    else:
                        self.f.update(self.wl.pos[i], self.wl.quat[i], self.wl.notch[i])

        def capture(self, n):
            """steps 0..n-1 of the step sequence (horizon after horizon) as one HIP graph: at ~0.1 ms of kernel time
            per step the Python/ctypes launch path would otherwise show up in the step time"""
            if n not in self.graphs:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self.eager(0, min(n, H))                 # warm-up on the capture stream
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self.eager(0, n)
                torch.cuda.synchronize(dev)
                self.graphs[n] = g
            return self.graphs[n]

        def read_calibration(self):
            """state.dofs of every instance into self.dofs (the payload of the final gather): flush + one small kernel,
            replayed from a graph like the steps"""
            if args.no_graph:
                return self.f.calibration(out=self.dofs)
            if self.cal_graph is None:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self.f.calibration(out=self.dofs)
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self.f.calibration(out=self.dofs)
                torch.cuda.synchronize(dev)
                self.cal_graph = g
            self.cal_graph.replay()
            return self.dofs

        def plan(self, nsteps):
            """the launches of `nsteps` consecutive steps, each a graph replay -- every step of the timed region is
            replayed, none is launched from Python: a short region (<= 16 horizons) is ONE graph (one launch from the
            host), a long one whole blocks of 16 horizons, then the rest"""
            if not (args.mode == "pair" and not args.no_graph):
                return [("eager", 0, nsteps)]
            blk = 16 * H
            full, rest = divmod(nsteps, blk)
            ops = [("graph", self.capture(blk), blk)] * full
            if rest:
                ops.append(("graph", self.capture(rest), rest))
            return ops

        def execute(self, ops):
            k = 0
            for kind, g, n in ops:
                if kind == "graph":
                    g.replay()
                else:
                    self.eager(k, n)
                k += n

        def sustain(self, seconds):
            """untimed: keeps the GPU under this very load until its clocks have settled (a cold GPU reads
            ~15 % low over the first tens of milliseconds: 4.5e8 over 9 ms against 5.2e8 over 0.13 s, same build)"""
            if seconds <= 0 or args.mode != "pair" or args.no_graph:
                return 0
            g, n = self.capture(H), 0
            t_end = time.perf_counter() + seconds
            while time.perf_counter() < t_end:
                for _ in range(8):
                    g.replay()
                torch.cuda.synchronize(dev)
                n += 8 * H
            return n

    def measure(B, dump=True):
        """the contract's timed region for B instances on this rank: warm-up, EXACTLY K timed steps between barriers, MAX over ranks"""
        run = Run(B, rank * B)
        wl, f, status = run.wl, run.f, run.status

        def all_gather_dofs(dofs):
            if backend == "nccl":
                dist.all_gather(gathered, dofs)                  # RCCL: the only collective on the path
            else:
                host = [torch.empty((B, 6), dtype=torch.float64) for _ in range(world)]
                dist.all_gather(host, dofs.cpu())
                for g_, h_ in zip(gathered, host):
                    g_.copy_(h_)

        run.eager(0, W)                                          # the W warm-up steps
        gathered = None
        run.read_calibration()                                   # (captures its graph: outside the timed region)
        if dist is not None:   # warm the collective up as well
            gathered = [torch.empty((B, 6), dtype=torch.float64, device=dev) for _ in range(world)]
            all_gather_dofs(run.dofs)
        torch.cuda.synchronize(dev)
        ops = run.plan(K)                                        # (captures the graphs: outside the timed region)
        # the first launch of an instantiated graph uploads it to the device (0.1-0.2 ms on these boxes, a fifth of a
        # 20-step region): every graph of the timed region is launched once here, like the warm-up steps
        seen = set()
        for kind, g_, _n in ops:
            if kind == "graph" and id(g_) not in seen:
                seen.add(id(g_))
                g_.replay()
        torch.cuda.synchronize(dev)
        sustained_steps = run.sustain(args.sustain_seconds)

        barrier()
        torch.cuda.synchronize(dev)
        ev_t = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        t0 = time.perf_counter()
        ev_t[0].record()                                         # HIP events on the launch stream around the K steps
        run.execute(ops)
        ev_t[1].record()
        dofs = run.read_calibration()
        if dist is not None:
            all_gather_dofs(dofs)
        torch.cuda.synchronize(dev)
        barrier()
        elapsed = time.perf_counter() - t0
        graph = any(o[0] == "graph" for o in ops)
        kern_ms = ev_t[0].elapsed_time(ev_t[1]) / K

        if args.dump_dofs and rank == 0 and dump:
            allv = torch.cat(gathered, dim=0) if gathered is not None else dofs
            np.save(args.dump_dofs, allv.cpu().numpy())

        # kernel time of one step from eagerly launched steps (diagnostic; contains the host's launch gaps).  With the time
        # loop in the kernel an eager launch is a whole horizon (every launch of a run is: the profile summaries rely on it).
        if args.mode == "pair" and LOOP_IN_KERNEL:
            n_ev, per_ev = 2, H
        else:
            n_ev, per_ev = min(K, 2 * H), 1
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for k in range(n_ev):
            ev[k][0].record()
            run.eager(k * per_ev, per_ev)
            ev[k][1].record()
        torch.cuda.synchronize(dev)

        # every instance must have done the full work: no singular / skipped updates, finite state
        st = (run.status_run if (args.mode == "pair" and LOOP_IN_KERNEL) else status).cpu().numpy()
        x_fin = f.x
        ok = bool((st == 0).all()) and bool(torch.isfinite(x_fin).all())
        kern_ms_eager = float(np.mean([a.elapsed_time(b) for a, b in ev])) / per_ev

        # the same measurement at a batch whose working set does not fit the 256 MiB Infinity Cache (the headline
        # batch's ~225 MB does, EXPERIMENTS.md "Round 4"): what the step sustains when it has to stream from HBM
        streaming = None
        if rank == 0 and world == 1 and dist is None and args.mode == "pair" and not args.no_streaming and not args.no_graph:
            Bs = args.streaming_batch
            del run.graphs
            big = Run(Bs, 0)
            nS = 4 * H
            big.eager(0, H)
            ops_s = big.plan(nS)
            big.sustain(min(args.sustain_seconds, 0.2))
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            big.execute(ops_s)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / nS
            ok_s = bool(((big.status_run if LOOP_IN_KERNEL else big.status).cpu().numpy() == 0).all())
            ach = BYTES_PER_PAIR_F64 * Bs / (ms * 1e-3) / 1e9
            streaming = {"batch": Bs, "steps": nS, "kernel_ms": ms, "pair_steps_per_s": Bs / (ms * 1e-3), "achieved": ach,
                         "frac": ach / HBM_PEAK_GBS, "valid": ok_s,
                         "note": "working set ~%d MB: streams from HBM (Infinity Cache 256 MiB)" % (Bs * 3450 // 1000000)}
            del big

        t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if dist is not None:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return dict(B=B, elapsed=float(t_max.item()), kern_ms=kern_ms, kern_ms_eager=kern_ms_eager, ok=ok, graph=graph, n_ops=len(ops),
                    sustained_steps=sustained_steps, streaming=streaming, wl=wl)

    # the timed region(s): weak scaling keeps --batch instances per GPU, strong scaling shards --batch over the ranks
    results = {}
    for mode in modes:
        results[mode] = measure(args.batch if mode == "weak" else args.batch // world)
        if len(modes) > 1:
            torch.cuda.empty_cache()
    main_mode = modes[-1]                                    # (both measured: the strong-scaling figure is the line's `value`)
    m = results[main_mode]
    B, elapsed, kern_ms, kern_ms_eager, ok, graph, streaming, wl = (m["B"], m["elapsed"], m["kern_ms"], m["kern_ms_eager"], m["ok"],
                                                                   m["graph"], m["streaming"], m["wl"])
    ok = all(r["ok"] for r in results.values())
    sustained_steps, n_ops = m["sustained_steps"], m["n_ops"]

    if rank == 0:
        total_pairs = float(B) * world * K
        value = total_pairs / elapsed
        achieved = BYTES_PER_PAIR_F64 * B / (kern_ms * 1e-3) / 1e9
        achieved_wall = BYTES_PER_PAIR_F64 * B / (elapsed / K) / 1e9
        traffic, traffic_source = None, None
        measured_cfg = False
        tf = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        if os.path.exists(tf):
            try:
                # (the PMC pass measured exactly one configuration: the default fp64 pair step at the headline batch through
                #  the library's default kernels, shared noise model, restore by reading the snapshot -- nothing else gets it)
                measured_cfg = (B == 65536 and args.lanes == 0 and args.mode == "pair" and not args.per_instance_noise
                                and not args.restore_copy and not os.environ.get("DVI_ESKF_LIB")
                                and LOOP_IN_KERNEL == LOOP_IN_KERNEL_DEFAULT and H == 16)
                tfj = json.load(open(tf))
                traffic = tfj.get("hbm_bytes_per_step", tfj.get("hbm_bytes_per_launch")) if measured_cfg else None
            except Exception:
                traffic = None
            if traffic is not None:
                traffic_source = ("static: profiles/traffic_per_launch.json -- a constant taken from the committed rocprofv3 "
                                  "PMC pass of this kernel pair (FETCH_SIZE x2 + WRITE_SIZE per step), NOT measured in this run")
        # A yardstick next to the 8 TB/s figure: the rate at which this box copies a buffer of the step's own measured
        # traffic (half read, half written) -- what its memory system gives the simplest kernel with the same mix.
        yardstick = None
        if traffic and world == 1 and not args.no_streaming:
            try:
                n8 = int(traffic) // 16
                src = torch.empty(n8, dtype=torch.float64, device=dev).normal_()
                dst = torch.empty_like(src)
                for _ in range(5):
                    dst.copy_(src)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(100):
                    dst.copy_(src)
                c1.record()
                torch.cuda.synchronize(dev)
                copy_gbs = 16.0 * n8 * 100 / (c0.elapsed_time(c1) * 1e-3) / 1e9
                step_gbs = float(traffic) / (kern_ms * 1e-3) / 1e9
                yardstick = {"copy_GBps": copy_gbs, "step_traffic_GBps": step_gbs, "step_over_copy": step_gbs / copy_gbs,
                             "note": "device-to-device copy of traffic/2 bytes (read + write counted), measured in this run; "
                                     "step_traffic = roofline.traffic / kernel_ms"}
                del src, dst
                if streaming:       # the same comparison at the streaming batch (traffic scaled with the batch)
                    ns = int(traffic * streaming["batch"] / B) // 16
                    src = torch.empty(ns, dtype=torch.float64, device=dev).normal_()
                    dst = torch.empty_like(src)
                    for _ in range(2):
                        dst.copy_(src)
                    c0.record()
                    for _ in range(20):
                        dst.copy_(src)
                    c1.record()
                    torch.cuda.synchronize(dev)
                    sc = 16.0 * ns * 20 / (c0.elapsed_time(c1) * 1e-3) / 1e9
                    st = float(traffic) * streaming["batch"] / B / (streaming["kernel_ms"] * 1e-3) / 1e9
                    yardstick.update({"streaming_copy_GBps": sc, "streaming_step_traffic_GBps": st, "streaming_step_over_copy": st / sc})
                    del src, dst
            except Exception as ex:     # (never the reason a bench line is missing)
                yardstick = {"error": repr(ex)}
        out = {
            "metric": "ESKF predict+update steps/sec, 26-state, batch=65536 at 1/2/4/8 MI355X",
            "value": value,
            "unit": "pair-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": main_mode,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            # (a line measured with an experiment build of the library -- DVI_ESKF_LIB -- is never a valid bench line)
            "valid": ok and args.mode == "pair" and not os.environ.get("DVI_ESKF_LIB"),
            "all_instances_ok": ok,
            "config": {
                "workload": "BASELINE config #2/#4: Monte-Carlo batch on mandala0_mono, 1:1 predict+update "
                            "pairs (dt=1.0), unfrozen DOFs, per-instance IMU/camera noise and perturbed ICs",
                "batch_per_gpu": B, "global_batch": B * world, "horizon_between_state_restores": H,
                "state_restore_in_timed_region": True,
                "state_restore": "device-to-device copy of the snapshot" if args.restore_copy else
                                 "the first pair-step of a horizon reads the snapshot handle (dvi_eskf_step_from)",
                "covariance_kernel": "row-resident" if (args.lanes or 1) == 1 else "lds-image, %d lanes/instance" % args.lanes,
                "hip_graph": graph,
                "time_loop": "in the kernel: a horizon is one dvi_eskf_run_epochs_from call = one launch of eskf_run_kernel" if LOOP_IN_KERNEL
                             else "on the host: one dvi_eskf_step call per step",
                "timed_region": "every one of the %d steps is replayed from a HIP graph (%d graph launch%s)" % (K, n_ops, "" if n_ops == 1 else "es")
                                if graph else "steps launched from Python",
                "untimed_preparation": "%d warm-up steps, graph capture and one launch of every captured graph, then %d replayed steps (%.2f s) to reach sustained clocks"
                                       % (W, sustained_steps, args.sustain_seconds),
                **({"library_override": os.environ["DVI_ESKF_LIB"]} if os.environ.get("DVI_ESKF_LIB") else {}),
                "final_collective": ("RCCL all_gather of dofs[B,6]" if backend == "nccl" else "gloo all_gather of dofs[B,6] (rehearsal)")
                                    if dist is not None else "none (single GPU)",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_wall": achieved_wall / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source,
                "traffic_over_algorithmic": (traffic / (BYTES_PER_PAIR_F64 * B)) if traffic else None,
                # what actually crossed the fabric, against the same peak: `frac` prices the ALGORITHMIC bytes (a quarter of the
                # covariances never leaves LDS between two steps of a run, so the fabric moves fewer)
                "frac_traffic": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "valu_issue": _valu_issue("headline_pairs_fp64_B65536") if measured_cfg else None,
                "bound_note": "priced against HBM as SURVEY 8(d) defines the metric; by the counters of the committed profile neither pipe "
                              "is saturated at this batch -- the fabric moves `traffic` at `frac_traffic` of peak, the vector ALUs are "
                              "issue-busy for `valu_issue.valu_busy` of the kernel's time: what the step lacks is overlap of the two",
                "kernel": ("eskf_run_kernel: ONE launch per horizon (%d steps; the time loop, every step's pre phase and its "
                           "row-resident covariance rounds inside; between two steps of a launch the covariances of a quarter of "
                           "the instances stay in LDS); achieved = algorithmic bytes of a step / (launch duration / steps "
                           "of the launch)" % H) if LOOP_IN_KERNEL and not args.lanes else
                          "one step = eskf_pre4_kernel + eskf_cov_rows_kernel (dominant; eskf_cov_kernel<LPI> with --lanes)",
                "timing": "frac: HIP events on the launch stream around the K steps of the timed region; frac_wall: from "
                          "ms_per_step (host clock, includes the final calibration read, the synchronisation and barriers)",
                "kernel_ms": kern_ms, "kernel_ms_eager_launches": kern_ms_eager,
                "algorithmic_bytes_per_launch": BYTES_PER_PAIR_F64 * B,
                "streaming": streaming,
                "copy_yardstick": yardstick,
            },
        }
        if len(modes) > 1:      # N > 1 without --strong / --weak: the weak-scaling measurement of the same job, next to `value`
            w_ = results["weak"]
            out["weak"] = {"value": float(w_["B"]) * world * K / w_["elapsed"], "unit": "pair-steps/s", "scaling": "weak",
                           "batch_per_gpu": w_["B"], "global_batch": w_["B"] * world, "ms_per_step": w_["elapsed"] / K * 1e3,
                           "kernel_ms": w_["kern_ms"], "all_instances_ok": w_["ok"],
                           "note": "the same timed region (barrier, EXACTLY K steps, final all-gather, MAX over ranks) with --batch instances "
                                   "PER GPU; `value` of this line is the strong-scaling figure (--batch instances sharded over the ranks)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
            out["cpu_baseline_numpy"] = cpu_baseline_numpy(wl)
            out["cpu_baseline_one_thread"] = cpu_baseline(wl, seconds_target=4.0, threads=1)
            out["parity"] = parity_check(wl)
        del results, m, wl
        torch.cuda.empty_cache()
        if world == 1 and dist is None and not args.no_secondary and args.mode == "pair":
            try:
                out["secondary"] = secondary_block(dev, lambda b_: measure(b_, dump=False))
            except Exception as ex:     # (never the reason a bench line is missing)
                out["secondary"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if not ok and args.mode == "pair":
        raise SystemExit("benchmark invalid: an instance skipped its update or went non-finite")


if __name__ == "__main__":
    main()
