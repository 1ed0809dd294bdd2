"""Child process of tests/test_gpu_pre4.py: runs one deterministic scenario through the C ABI and saves what the
kernels left behind.  The caller runs it twice -- once with DVI_ESKF_PRE_LEGACY=1 in the environment (single-predict
launches through the two-role pre kernel), once without (eskf_pre4_kernel) -- and compares the files bit for bit."""
import sys

import numpy as np
import torch

sys.path.insert(0, sys.argv[3])
from tests.helpers import default_run                      # noqa: E402
from tests.test_gpu_parity import _make, _random_configuration, _random_unit_quats   # noqa: E402


def main(scenario, out):
    frozen, hold, dtype, keep, B = {
        "unfrozen": ((0,) * 6, False, torch.float64, False, 777),
        "mixed_zeroed": ((1, 0, 1, 0, 0, 1), False, torch.float64, False, 300),
        "frozen_hold": ((1,) * 6, True, torch.float64, False, 130),
        "keep_jac": ((0,) * 6, False, torch.float64, True, 200),
        "fp32": ((0, 1, 0, 1, 0, 0), False, torch.float32, False, 515),
        "pin": ((0,) * 6, False, torch.float64, False, 333),      # per-instance noise records (a tuning population)
    }[scenario]
    rng = np.random.default_rng(99)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True, frozen=frozen)
    x0 = _random_configuration(rng, B)
    A = rng.standard_normal((B, 24, 24)) * 10.0 ** rng.uniform(-2, 0, (B, 24, 1))
    P0 = A @ A.transpose(0, 2, 1) / 24 + 1e-4 * np.eye(24)
    Rl = np.diag([1e-2] * 3 + [1e-4] * 3 + [1e-4])
    f = _make(cfg, ri, Q, Rl, sa, so, B, hold, x0, dtype=dtype)
    if keep:
        f.keep_jacobians(True)
    if scenario == "pin":
        q13 = np.tile(np.diag(Q), (B, 1)) * rng.uniform(0.5, 2.0, (B, 13))
        q13[:, 3:6] = 0.0                                       # (no gyro block in Q: the run-mode model)
        r7 = np.tile(np.diag(Rl), (B, 1)) * rng.uniform(0.5, 2.0, (B, 7))
        f.set_noise_per_instance(q13, r7, np.tile(so, (B, 1)) * rng.uniform(0.5, 2.0, (B, 3)))
    f.set_state(x=x0, P=P0, om_old=rng.normal(0, 0.5, (B, 3)), acc_old=rng.normal(0, 20.0, (B, 3)))
    res = {}
    g = f.clone()               # a single predict from the initial state (nothing pending), looked at in isolation
    if keep:
        g.keep_jacobians(True)
    g.predict(0.1 * rng.uniform(0.5, 1.5, B), rng.normal(0, 0.5, (B, 3)), rng.normal(0, 20.0, (B, 3)))
    for key, v in g.get_state().items():
        res["p_" + key] = v.cpu().numpy()
    if keep:
        Fx, Fi = g.jacobians()
        res["p_Fx"], res["p_Fi"] = Fx.cpu().numpy(), Fi.cpu().numpy()
    st, _ = g.update(x0[:, 19:22] + 0.3, x0[:, 22:26], x0[:, 16])    # (update-only launch: the two-role kernel on both sides)
    g.predict(0.07, rng.normal(0, 0.5, 3), rng.normal(0, 20.0, 3))    # ... and a predict that injects it
    for key, v in g.get_state().items():
        res["pu_" + key] = v.cpu().numpy()
    for k in range(3):          # step 0 has nothing pending, steps 1 and 2 inject the previous update on the fly
        dt = 0.1 * rng.uniform(0.5, 1.5, B)
        om, acc = rng.normal(0, 0.5, (B, 3)), rng.normal(0, 20.0, (B, 3))
        pos = x0[:, 19:22] + rng.normal(0, 1.0, (B, 3))
        quat = _random_unit_quats(rng, B) * rng.uniform(0.5, 2.0, (B, 1))
        near = rng.random(B) < 0.7
        quat[near] = x0[near, 22:26] + rng.normal(0, 0.05, (near.sum(), 4))
        notch = x0[:, 16] + rng.normal(0, 0.01, B)
        if k == 1:
            pos[5], quat[7], notch[9] = np.nan, np.inf, np.nan      # non-finite residual flags
        st, _ = f.step(dt, om, acc, pos, quat, notch)
        res[f"st{k}"] = st.cpu().numpy()
        if keep:
            Fx, Fi = f.jacobians()
            res[f"Fx{k}"], res[f"Fi{k}"] = Fx.cpu().numpy(), Fi.cpu().numpy()
    f.predict(0.05, rng.normal(0, 0.5, 3), rng.normal(0, 20.0, 3))   # a lone predict, broadcast sample, injection pending
    for key, v in f.get_state().items():
        res[key] = v.cpu().numpy()
    # epochs (n predicts + update in one launch): per-instance IMU streams with an injection pending, then a shared stream
    # with moving notch rates (the trigonometry of phi changes from predict to predict)
    h = f.clone()
    n = 5
    st, _ = h.step(0.05, rng.normal(0, 0.5, 3), rng.normal(0, 20.0, 3), x0[:, 19:22], x0[:, 22:26], x0[:, 16])
    st, _ = h.epoch(0.02 * rng.uniform(0.5, 1.5, (n, B)), rng.normal(0, 0.5, (n, B, 3)), rng.normal(0, 20.0, (n, B, 3)),
                    x0[:, 19:22] + 0.2, x0[:, 22:26], x0[:, 16])
    res["ep_st"] = st.cpu().numpy()
    for key, v in h.get_state().items():
        res["ep_" + key] = v.cpu().numpy()
    if keep:
        Fx, Fi = h.jacobians()
        res["ep_Fx"], res["ep_Fi"] = Fx.cpu().numpy(), Fi.cpu().numpy()
    n = 7
    st, _ = h.epoch(np.full(n, 0.01), rng.normal(0, 0.5, (n, 3)), rng.normal(0, 20.0, (n, 3)), x0[:, 19:22] + 0.1, x0[:, 22:26], x0[:, 16])
    res["ep2_st"] = st.cpu().numpy()
    for key, v in h.get_state().items():
        res["ep2_" + key] = v.cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
