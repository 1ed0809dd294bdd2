"""Child process of tests/test_gpu_epochs_kernel.py: a run whose epochs have several predicts, enqueued as ONE launch
(dvi_eskf_run_epochs -> eskf_epochs_kernel: producer wavefronts run the recurrences and the Jacobian blocks ahead of the
covariance wavefronts, the covariances stay in registers for the whole run) against the same epochs enqueued one by one
(dvi_eskf_epoch -> eskf_nominal_kernel + eskf_jac_kernel + eskf_cov_rows_kernel), in the same process: state, covariance,
buffers and every epoch's status words must be the same bits.  The block size of the new kernel comes from the environment
(DVI_ESKF_EPOCHS_NC), hence the child.
usage: _epochs_child.py <batch> <f64|f32> <repo root> <schedule: comma-separated predict counts> [frozen mask, 6 digits] [hold 0|1]"""
import sys

import numpy as np
import torch

sys.path.insert(0, sys.argv[3])
from dvi_ekf_amd.workloads import monte_carlo_pairs      # noqa: E402


def same(a, b, what):
    for k in a:
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        assert np.array_equal(x, y, equal_nan=True), (what, k, float(np.nanmax(np.abs(x - y))), int((x != y).sum()), x.size)


def close(a, b, what, tol):
    for k in a:
        x, y = a[k].cpu().numpy().astype(np.float64), b[k].cpu().numpy().astype(np.float64)
        assert np.array_equal(np.isfinite(x), np.isfinite(y)), (what, k, "non-finite entries differ")       # (the instance with the NaN measurement)
        fin = np.isfinite(y)
        s = max(1.0, float(np.abs(y[fin]).max()))
        assert float(np.abs(x[fin] - y[fin]).max()) <= tol * s, (what, k, float(np.abs(x[fin] - y[fin]).max()), s)


def main(B, dtype, n_pred, frozen, hold, rows_closeness=True):
    dev = torch.device("cuda", 0)
    E = len(n_pred)
    total = int(sum(n_pred))
    H = 8
    wl = monte_carlo_pairs(B, horizon=H, seed=4, device=dev, dtype=dtype, frozen=frozen)
    f = wl.make_filter(dtype=dtype)
    if not hold:
        # (frozen DOFs zeroed after every predict, Filter.py:243-245: the v1-literal plumbing)
        from dvi_ekf_amd.batch import BatchedESKF
        f = BatchedESKF(B, length=wl.cfg.model.length, theta_cam=wl.cfg.model.angle, frozen=wl.cfg.frozen_dofs, sigma_a=wl.sigma_a,
                        sigma_om=wl.sigma_om, hold_frozen=False, dtype=dtype, device=dev)
        f.set_noise(wl.Q, wl.R)
        f.set_state(x=wl.x0.to(dtype), P=wl.P0.to(dtype), om_old=wl.om0.to(dtype), acc_old=wl.acc0.to(dtype))
    snap = f.clone()
    # per-instance IMU streams: the samples of the Monte-Carlo pairs, cycled, at dt = 0.1 (per-instance dt with a little spread)
    idx = torch.arange(total, device=dev) % H
    om = wl.om[idx].contiguous()
    acc = wl.acc[idx].contiguous()
    dt = (0.1 * (1.0 + 0.01 * torch.arange(B, device=dev, dtype=dtype) / max(B, 1))).reshape(1, B).repeat(total, 1).contiguous()
    ide = torch.arange(E, device=dev) % H
    pos, quat, notch = wl.pos[ide].clone(), wl.quat[ide].contiguous(), wl.notch[ide].contiguous()
    if E > 2 and B > 3:
        pos[2, 3, 0] = float("nan")          # a NaN camera position in epoch 2 of instance 3: flagged there, on both paths
    # A: epoch by epoch
    sts, off = [], 0
    for e, n in enumerate(n_pred):
        if n > 0:
            st, _ = f.epoch(dt[off:off + n], om[off:off + n], acc[off:off + n], pos[e], quat[e], notch[e])
        else:
            st, _ = f.update(pos[e], quat[e], notch[e])
        sts.append(st.clone())
        off += n
    ref = f.get_state()
    ref_st = torch.stack(sts)
    if E > 2 and B > 3:
        assert int(ref_st[2, 3]) & 2
    exact = all(n >= 2 for n in n_pred)     # (epochs of one predict go through the four-chain pre kernel on path A: a different
                                            #  rounding sequence in the last bits, see tests/test_gpu_pre4.py)
    # (an update that follows the one before after one predict or none meets an innovation covariance at the level of the
    #  measurement noise, cond(S) ~ 1e6 and more: a last-bit difference of the nominal state in front of it comes out at 1e-8)
    cmp = same if exact else (lambda a, b, w: close(a, b, w, 1e-6 if dtype == torch.float64 else 2e-2))
    # B: one launch
    g = snap.clone()
    st, rows = g.run_epochs(n_pred, dt, om, acc, pos, quat, notch)
    assert rows is None
    torch.cuda.synchronize()
    if exact:
        assert torch.equal(st, ref_st), ("status words", (st != ref_st).nonzero()[:8].tolist())
    cmp(g.get_state(), ref, "run_epochs")
    # C: the same run started from the snapshot handle, over a filter that holds something else (and a pending injection)
    h2 = f.clone()
    h2.step(wl.dt[0], wl.om[0], wl.acc[0], wl.pos[0], wl.quat[0], wl.notch[0])
    st2 = torch.full((E, B), -1, dtype=torch.int32, device=dev)
    h2.run_epochs(n_pred, dt, om, acc, pos, quat, notch, src=snap, status=st2)
    if exact:
        assert torch.equal(st2, ref_st), "status words (from)"
    cmp(h2.get_state(), ref, "run_epochs_from")
    # D: continuing a run: the second part enqueued on its own picks up the injection the first part left pending
    if E >= 2:
        g2 = snap.clone()
        e1 = E // 2
        o1 = int(sum(n_pred[:e1]))
        g2.run_epochs(n_pred[:e1], dt[:o1], om[:o1], acc[:o1], pos[:e1], quat[:e1], notch[:e1])
        g2.run_epochs(n_pred[e1:], dt[o1:], om[o1:], acc[o1:], pos[e1:], quat[e1:], notch[e1:])
        cmp(g2.get_state(), ref, "two parts")
    # E: one input stream shared by the batch (instance 1's), against epochs with the broadcast samples
    i1 = min(1, B - 1)
    f3, g3 = snap.clone(), snap.clone()
    dts, oms, accs = dt[:, i1].contiguous(), om[:, i1].contiguous(), acc[:, i1].contiguous()
    poss, quats, notchs = wl.pos[ide][:, i1].contiguous(), quat[:, i1].contiguous(), notch[:, i1].contiguous()
    off = 0
    for e, n in enumerate(n_pred):
        if n > 0:
            f3.epoch(dts[off:off + n], oms[off:off + n], accs[off:off + n], poss[e], quats[e], notchs[e])
        else:
            f3.update(poss[e], quats[e], notchs[e])
        off += n
    g3.run_epochs(n_pred, dts, oms, accs, poss, quats, notchs)
    cmp(g3.get_state(), f3.get_state(), "shared streams")
    # F: the trajectory log -- a row per IMU sample, the row of an epoch's last sample replaced by the corrected state -- from
    # the run's own state snapshots
    if total > 0:
        t_pred = np.arange(total, dtype=np.float64) * 0.1 + 200.0
        g4 = snap.clone()
        nl = min(B, 3)
        _, rows = g4.run_epochs(n_pred, dt, om, acc, pos, quat, notch, t_pred=t_pred, rows_instances=nl)
        assert rows.shape == (total, nl, 30)
        cmp(g4.get_state(), ref, "run with log rows")
        # (i) the row an update belongs to is the log row of the state a run has when it ends with that update: the run cut after
        #     every epoch in turn (a run gives the same bits wherever it is cut into calls)
        cum = np.cumsum(n_pred)
        for e in range(E):
            if cum[e] == 0:
                continue
            gk = snap.clone()
            o = int(cum[e])
            gk.run_epochs(n_pred[:e + 1], dt[:o], om[:o], acc[:o], pos[:e + 1], quat[:e + 1], notch[:e + 1])
            want = gk.log_rows(float(t_pred[o - 1]))[:nl]
            if e + 1 < E and n_pred[e + 1] == 0:
                continue                            # (the next epoch has no sample of its own: its update replaces this row once more)
            if not (torch.equal(rows[o - 1], want) or bool(torch.isnan(want).any())):
                d = (rows[o - 1].double() - want.double()).abs()
                raise AssertionError(("row of update", e, "max abs", float(d.max()), "where (instance, column)", torch.nonzero(d > 0)[:8].tolist()))
        # (ii) the rows of the samples in between (propagated states) against the reference's call sequence (predict, log row, ...,
        #      update, log row) on a handle that keeps its Jacobians and therefore never takes the one-launch path: another
        #      rounding sequence (P is symmetrised after every predict there), compared where the schedule leaves that benign
        if exact and rows_closeness:      # (tools/soak_epochs.py skips it: long random schedules of unfrozen instances amplify a last bit past any bound)
            f4 = snap.clone()
            f4.keep_jacobians(True)
            _, rows_ref = f4.run_epochs(n_pred, dt, om, acc, pos, quat, notch, t_pred=t_pred)
            assert rows_ref.shape == (total, B, 30)
            r, rr = rows.cpu().numpy().astype(np.float64), rows_ref[:, :nl].cpu().numpy().astype(np.float64)
            assert np.array_equal(r[:, :, 0], rr[:, :, 0])
            fin = np.isfinite(rr) & np.isfinite(r)
            tol = 1e-6 if dtype == torch.float64 else 2e-2
            assert np.abs(r[fin] - rr[fin]).max() <= tol * max(1.0, np.abs(rr[fin]).max()), float(np.abs(r[fin] - rr[fin]).max())
    print("epochs child ok", B, dtype, n_pred)


if __name__ == "__main__":
    sched = [int(v) for v in sys.argv[4].split(",")]
    frozen = tuple(int(c) for c in (sys.argv[5] if len(sys.argv) > 5 else "000000"))
    hold = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
    main(int(sys.argv[1]), torch.float64 if sys.argv[2].startswith("f64") else torch.float32, sched, frozen, hold)
