"""C-ABI behaviour on a real device: argument validation (negative codes, nothing thrown),
deferred-injection bookkeeping, raw buffer views."""
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tests.helpers import default_run


def _make(B=4, hold=True):
    from dvi_ekf_amd.batch import BatchedESKF
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa,
                    sigma_om=so, hold_frozen=hold)
    f.set_noise(Q, R)
    f.set_state(x=ri.x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    return f, ri, Q, R


def test_error_codes_instead_of_exceptions():
    from dvi_ekf_amd import _capi
    f, ri, Q, R = _make()
    L = _capi.lib()
    dp = C.POINTER(C.c_double)
    Qbad = np.ascontiguousarray(Q.copy())
    Qbad[0, 1] = 1e-3                                    # the reference never builds a non-diagonal Q
    assert L.dvi_eskf_set_noise(f._h, Qbad.ctypes.data_as(dp), None) == -1
    assert L.dvi_eskf_set_lanes_per_instance(f._h, 7) == -1
    assert L.dvi_eskf_predict(f._h, None, 0, None, None, 0, None) == -1      # missing inputs
    assert L.dvi_eskf_update(None, None, None, None, 0, None, None, None) == -1
    p = _capi.Params()
    h = C.c_void_p()
    assert L.dvi_eskf_create(C.byref(p), 0, 0, 0, C.byref(h)) == -1           # batch must be positive
    assert L.dvi_eskf_create(C.byref(p), 4, 7, 0, C.byref(h)) == -1           # unknown dtype
    assert L.dvi_eskf_create(C.byref(p), 4, 0, 4096, C.byref(h)) == -2        # no such device
    assert L.dvi_eskf_log_rows(f._h, 0.0, None, None) == -1                   # no output buffer
    assert L.dvi_eskf_set_noise_per_instance(f._h, None, None, None, None) == -1
    qn = np.full((4, 13), np.nan)
    assert L.dvi_eskf_set_noise_per_instance(f._h, qn.ctypes.data_as(dp), np.ones((4, 7)).ctypes.data_as(dp), None, None) == -1
    with pytest.raises(_capi.EskfError):
        f.set_noise(Qbad, None)
    # the handle is still usable
    f.predict(ri.dt[0], ri.om[0], ri.acc[0])
    assert bool(torch.isfinite(f.x).all())


def test_deferred_injection_is_invisible_to_the_caller():
    """After step(), every way of looking at the state shows the injected state."""
    f, ri, Q, R = _make()
    g = f.clone()
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    g.predict(ri.dt[0], ri.om[0], ri.acc[0])
    x_pred = g.x.clone()
    g.update(ri.pos[0], ri.quat[0], ri.notch[0])
    x_upd = g.x
    assert not torch.equal(x_pred, x_upd)
    assert torch.allclose(f.x, x_upd, rtol=0, atol=1e-12)
    assert torch.allclose(f.calibration(), x_upd[:, 10:16], rtol=0, atol=1e-12)
    assert torch.allclose(f.clone().x, x_upd, rtol=0, atol=1e-12)
    # a second step on top of a pending injection equals flush-then-step
    a, b = f.clone(), f.clone()
    a.step(ri.dt[1], ri.om[1], ri.acc[1], ri.pos[1], ri.quat[1], ri.notch[1])
    a.step(ri.dt[2], ri.om[2], ri.acc[2], ri.pos[2], ri.quat[2], ri.notch[2])     # pending from the first
    b.step(ri.dt[1], ri.om[1], ri.acc[1], ri.pos[1], ri.quat[1], ri.notch[1])
    _ = b.x                                                                          # flush in between
    b.step(ri.dt[2], ri.om[2], ri.acc[2], ri.pos[2], ri.quat[2], ri.notch[2])
    sa, sb = a.get_state(), b.get_state()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key


def test_flush_and_raw_views():
    from dvi_ekf_amd import _capi
    f, ri, Q, R = _make(B=3)
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    L = _capi.lib()
    assert L.dvi_eskf_flush(f._h, None) == 0 and L.dvi_eskf_flush(f._h, None) == 0     # idempotent
    assert L.dvi_eskf_raw_x(f._h) and L.dvi_eskf_raw_P(f._h)
    assert L.dvi_eskf_batch(f._h) == 3 and L.dvi_eskf_dtype(f._h) == 0
    st = f.get_state()
    assert st["x"].shape == (3, 26) and st["P"].shape == (3, 24, 24)
    assert torch.equal(st["P"], st["P"].transpose(1, 2))          # stored as its lower triangle


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_run_epochs_equals_epoch_by_epoch(dtype):
    """dvi_eskf_run_epochs (Filter.run in one call) == n calls of dvi_eskf_epoch, bit for bit; status and
    log rows per epoch; ragged schedule (an epoch without IMU samples, epochs of different length)."""
    from tests.helpers import default_run
    from dvi_ekf_amd.batch import BatchedESKF
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False)
    B = 5
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=[0] * 6, sigma_a=sa, sigma_om=so,
                    hold_frozen=False, dtype=getattr(torch, dtype))
    f.set_noise(Q, R)
    rng = np.random.default_rng(7)
    x0 = np.tile(ri.x0, (B, 1))
    x0[:, 10:13] += rng.normal(0, 0.02, (B, 3))
    x0[:, 13:16] += rng.normal(0, 2.0, (B, 3))
    f.set_state(x=x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    g = f.clone()
    n_pred = np.array([10, 0, 3, 17, 10, 1], dtype=np.int32)       # ragged on purpose
    E, total = len(n_pred), int(n_pred.sum())
    # per-instance measurements, shared IMU stream
    pos = ri.pos[:E, None, :] + rng.normal(0, 1e-3, (E, B, 3))
    quat = np.repeat(ri.quat[:E, None, :], B, axis=1)
    notch = rng.normal(0, 1e-3, (E, B))
    t_log = ri.t_meas[:E]
    status, rows = f.run_epochs(n_pred, ri.dt[:total], ri.om[:total], ri.acc[:total], pos, quat, notch, t_log=t_log)
    off = 0
    for e in range(E):
        n = int(n_pred[e])
        if n:
            st, _ = g.epoch(ri.dt[off:off + n], ri.om[off:off + n], ri.acc[off:off + n], pos[e], quat[e], notch[e])
        else:
            st, _ = g.update(pos[e], quat[e], notch[e])
        off += n
        assert torch.equal(status[e], st)
        assert torch.equal(rows[e], g.log_rows(float(t_log[e])))
    sf, sg = f.get_state(), g.get_state()
    for key in sf:
        assert torch.equal(sf[key], sg[key]), key
    assert int(status.abs().sum()) == 0 and rows.shape == (E, B, 30) and bool(torch.isfinite(rows).all())
    assert torch.equal(rows[:, :, 0].cpu(), torch.as_tensor(t_log).to(rows.dtype)[:, None].expand(E, B))

    # broadcast measurements, no log
    h1, h2 = g.clone(), g.clone()
    st2, rows2 = h1.run_epochs([2, 2], ri.dt[:4], ri.om[:4], ri.acc[:4], ri.pos[:2], ri.quat[:2], ri.notch[:2])
    assert rows2 is None and st2.shape == (2, B)
    h2.epoch(ri.dt[:2], ri.om[:2], ri.acc[:2], ri.pos[0], ri.quat[0], ri.notch[0])
    h2.epoch(ri.dt[2:4], ri.om[2:4], ri.acc[2:4], ri.pos[1], ri.quat[1], ri.notch[1])
    assert torch.equal(h1.x, h2.x) and torch.equal(h1.get_state()["P"], h2.get_state()["P"])

    # argument validation: codes, not exceptions
    from dvi_ekf_amd import _capi
    L = _capi.lib()
    sc = _capi.Schedule()
    assert L.dvi_eskf_run_epochs(f._h, None, None) == -1
    sc.n_epochs = 0
    assert L.dvi_eskf_run_epochs(f._h, C.byref(sc), None) == 0               # an empty run is a no-op
    sc.n_epochs = 2
    assert L.dvi_eskf_run_epochs(f._h, C.byref(sc), None) == -1              # no schedule arrays
    bad = np.array([1, -1], dtype=np.int32)
    sc.n_pred = bad.ctypes.data_as(C.POINTER(C.c_int32))
    sc.pos = sc.quat_xyzw = sc.notch = 1                                      # never dereferenced: rejected first
    assert L.dvi_eskf_run_epochs(f._h, C.byref(sc), None) == -1              # negative sample count
    with pytest.raises(ValueError):
        f.run_epochs([2, 2], ri.dt[:3], ri.om[:3], ri.acc[:3], ri.pos[:2], ri.quat[:2], ri.notch[:2])


def test_gather_calibration_through_rccl():
    """dvi_eskf_gather_calibration on a real RCCL communicator (one rank on the one GPU of the test box:
    the collective degenerates to a copy, the binding, dtype and count are what is exercised)."""
    from dvi_ekf_amd import _capi
    f, ri, Q, R = _make(B=6, hold=False)
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    L = _capi.lib()
    assert L.dvi_eskf_gather_calibration(f._h, None, None, None) == -1
    rccl = None
    for name in ("librccl.so.1", "librccl.so"):
        for mode in (getattr(os, "RTLD_NOLOAD", 4) | os.RTLD_LAZY, os.RTLD_LAZY):
            try:
                rccl = C.CDLL(name, mode=mode)
                break
            except OSError:
                continue
        if rccl is not None:
            break
    if rccl is None:
        rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        out = f.gather_calibration(comm.value, world_size=1)
        torch.cuda.synchronize()
        assert torch.equal(out, f.calibration())
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("lpi", [None, 2, 32])
def test_step_from_a_snapshot_equals_copy_then_step(lpi):
    """dvi_eskf_step_from: h <- step(src) reads src's covariance and state record directly; bit for bit the
    result of copying src into h and stepping in place; src is untouched; pending injections on either
    side are handled (src's is applied to src first, h's is dropped with the state it belonged to)."""
    f, ri, Q, R = _make(B=37, hold=False)
    if lpi:
        f.set_lanes_per_instance(lpi)
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])        # leaves an injection pending
    snap = f.clone()                                                                 # (flushed copy)
    snap.step(ri.dt[1], ri.om[1], ri.acc[1], ri.pos[1], ri.quat[1], ri.notch[1])     # snapshot WITH a pending injection
    want_src = {k: v.clone() for k, v in snap.clone().get_state().items()}
    a, b = f.clone(), f.clone()
    if lpi:
        a.set_lanes_per_instance(lpi); b.set_lanes_per_instance(lpi)
    a.step(ri.dt[2], ri.om[2], ri.acc[2], ri.pos[2], ri.quat[2], ri.notch[2])        # a has its own pending injection
    st_a, _ = a.step_from(snap, ri.dt[3], ri.om[3], ri.acc[3], ri.pos[3], ri.quat[3], ri.notch[3])
    b.copy_state_from(snap)
    st_b, _ = b.step(ri.dt[3], ri.om[3], ri.acc[3], ri.pos[3], ri.quat[3], ri.notch[3])
    assert torch.equal(st_a, st_b)
    sa, sb, ss = a.get_state(), b.get_state(), snap.get_state()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key
        assert torch.equal(ss[key], want_src[key]), key
    # and it keeps going: a further in-place step agrees as well
    a.step(ri.dt[4], ri.om[4], ri.acc[4], ri.pos[4], ri.quat[4], ri.notch[4])
    b.step(ri.dt[4], ri.om[4], ri.acc[4], ri.pos[4], ri.quat[4], ri.notch[4])
    assert torch.equal(a.x, b.x) and torch.equal(a.get_state()["P"], b.get_state()["P"])
    # mismatched handles are refused
    from dvi_ekf_amd import _capi
    other, *_ = _make(B=5, hold=False)
    L = _capi.lib()
    assert L.dvi_eskf_step_from(a._h, other._h, 1, None, 0, None, None, 0, None, None, None, 0, None, None, None) == -1
    assert L.dvi_eskf_step_from(a._h, None, 1, None, 0, None, None, 0, None, None, None, 0, None, None, None) == -1


@pytest.mark.parametrize("dtype,B", [("float64", 1), ("float64", 1003), ("float32", 515)])
def test_copy_state_is_a_bitwise_restore(dtype, B):
    """dvi_eskf_copy_state (one launch of 16-byte moves over the packed covariance and the state record): the target
    holds the source's state bit for bit, whatever it held and whatever was pending on either side; the source is
    untouched; a copy onto itself changes nothing; handles of another batch size or dtype are refused."""
    from dvi_ekf_amd import _capi
    from dvi_ekf_amd.batch import BatchedESKF
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True)
    dt_ = getattr(torch, dtype)
    rng = np.random.default_rng(B)

    def make(n=B, d=dt_):
        f = BatchedESKF(n, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=cfg.frozen_dofs, sigma_a=sa,
                        sigma_om=so, hold_frozen=False, dtype=d)
        f.set_noise(Q, R)
        f.set_state(x=np.tile(ri.x0, (n, 1)) + 1e-3 * rng.standard_normal((n, 26)), P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
        return f

    src, dst = make(), make()
    src.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])      # src: an injection pending
    dst.step(ri.dt[1], ri.om[1], ri.acc[1], ri.pos[1], ri.quat[1], ri.notch[1])      # dst: another one, to be dropped
    dst.step(ri.dt[2], ri.om[2], ri.acc[2], ri.pos[2], ri.quat[2], ri.notch[2])
    want = {k: v.clone() for k, v in src.clone().get_state().items()}
    dst.copy_state_from(src)
    got, kept = dst.get_state(), src.get_state()
    for key in want:
        assert torch.equal(got[key], want[key]), key
        assert torch.equal(kept[key], want[key]), key
    dst.copy_state_from(dst)                                                         # onto itself
    for key, v in dst.get_state().items():
        assert torch.equal(v, want[key]), key
    # ... and the copy behaves like the original from here on
    st_a, _ = src.step(ri.dt[3], ri.om[3], ri.acc[3], ri.pos[3], ri.quat[3], ri.notch[3])
    st_b, _ = dst.step(ri.dt[3], ri.om[3], ri.acc[3], ri.pos[3], ri.quat[3], ri.notch[3])
    assert torch.equal(st_a, st_b)
    sa_, sb_ = src.get_state(), dst.get_state()
    for key in sa_:
        assert torch.equal(sa_[key], sb_[key]), key
    L = _capi.lib()
    assert L.dvi_eskf_copy_state(dst._h, make(B + 1)._h, None) == -1
    assert L.dvi_eskf_copy_state(dst._h, make(B, torch.float32 if dt_ == torch.float64 else torch.float64)._h, None) == -1
    assert L.dvi_eskf_copy_state(dst._h, None, None) == -1


def test_run_epochs_error_after_the_fork_still_joins_the_side_streams(tmp_path):
    """An error while the two halves of an under-filled batch are being enqueued on the handle's side streams must not leave
    those streams forked: eagerly the caller's stream has to wait for what was already enqueued on them, and under capture an
    unjoined stream would make hipStreamEndCapture fail (hipErrorStreamCaptureUnjoined) -- the call returns its error and the
    capture still ends.  The failure is injected by a hook that only a TEST BUILD of the library contains
    (-DESKF_TEST_HOOKS, built here into the test's temporary directory); the scenario runs in a child process with that
    build selected (tests/_split_child.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / "libdvi_eskf_testhooks.so"
    subprocess.check_call(["make", "-C", os.path.join(root, "dvi_ekf_amd", "csrc"), "-s", "-B", "EXTRA=-DESKF_TEST_HOOKS", f"OUT={lib}"], timeout=1200)
    env = dict(os.environ)
    env["DVI_ESKF_LIB"] = str(lib)
    env.pop("DVI_ESKF_EPOCHS", None)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "_split_child.py"), root], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "split child ok" in p.stdout


@pytest.mark.parametrize("parts,B,dtype", [("3", 8448, "f64"), ("4", 8448, "f32"), ("8", 9216, "f64")])
def test_run_epochs_on_k_streams_equals_epoch_by_epoch(parts, B, dtype):
    """DVI_ESKF_EPOCH_STREAMS=k: k parts of an under-filled batch (multiples of 64 instances) run their epochs on k streams of
    the handle's own.  Same bits as the epochs called one by one (child process: the switch is read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("DVI_ESKF_EPOCHS", "DVI_ESKF_EPOCHS_NC", "DVI_ESKF_EPOCHS_PP", "DVI_ESKF_EPOCH_STREAMS_MIN", "DVI_ESKF_EPOCH_STREAMS_MAX"):
        env.pop(k, None)
    env["DVI_ESKF_EPOCH_STREAMS"] = parts
    env["DVI_ESKF_EPOCHS"] = "0"        # (the epoch-by-epoch path also where the long-horizon kernel would take the run)
    env["DVI_ESKF_VERBOSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "_streams_child.py"), root, str(B), dtype], cwd=root, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "streams child ok" in p.stdout
    assert "%s parts of the batch" % parts in p.stderr, p.stderr[-2000:]


@pytest.mark.parametrize("dtype,per_instance,B", [("float64", False, 2176), ("float64", True, 8320), ("float32", False, 8320), ("float64", False, 8320)])
def test_run_epochs_on_two_streams_equals_epoch_by_epoch(dtype, per_instance, B):
    """An under-filled batch (B a multiple of 128, 2 048 <= B <= 16 384, every epoch with >= 2 predicts, no log rows):
    dvi_eskf_run_epochs runs the two halves of the batch on two streams of the handle's own -- one half's recurrences
    and Jacobian kernel next to the other half's covariance kernel.  Same bits as the epochs called one by one (which
    cover the whole batch per launch), status words included; also when the call is captured into a HIP graph."""
    from tests.helpers import default_run
    from dvi_ekf_amd.batch import BatchedESKF
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False)
    # (B = 8 320 = 65 x 128: beyond what the long-horizon kernel takes in one generation of blocks -- the two streams; B = 2 176:
    #  that kernel eagerly, the two streams under capture, where the ragged schedule cannot be uploaded)
    dt_ = getattr(torch, dtype)
    f = BatchedESKF(B, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=[0] * 6, sigma_a=sa, sigma_om=so,
                    hold_frozen=False, dtype=dt_)
    f.set_noise(Q, R)
    rng = np.random.default_rng(17)
    x0 = np.tile(ri.x0, (B, 1))
    x0[:, 10:13] += rng.normal(0, 0.02, (B, 3))
    x0[:, 13:16] += rng.normal(0, 2.0, (B, 3))
    x0[:, 17] = rng.normal(0, 0.05, B)                         # moving notch: the trigonometry of phi changes per predict
    f.set_state(x=x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    n_pred = np.array([10, 3, 17, 2, 10], dtype=np.int32)
    E, total = len(n_pred), int(n_pred.sum())
    pos = ri.pos[:E, None, :] + rng.normal(0, 1e-3, (E, B, 3))
    quat = np.repeat(ri.quat[:E, None, :], B, axis=1)
    notch = rng.normal(0, 1e-3, (E, B))
    if per_instance:
        dt = np.repeat(ri.dt[:total, None], B, axis=1) * rng.uniform(0.9, 1.1, (total, B))
        om = ri.om[:total, None, :] + rng.normal(0, 1e-3, (total, B, 3))
        acc = ri.acc[:total, None, :] + rng.normal(0, 1e-2, (total, B, 3))
    else:
        dt, om, acc = ri.dt[:total], ri.om[:total], ri.acc[:total]
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])        # (an injection is pending at the start)
    g, c = f.clone(), f.clone()
    g.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    c.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    f.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])
    status, _ = f.run_epochs(n_pred, dt, om, acc, pos, quat, notch)
    off = 0
    for e in range(E):
        n = int(n_pred[e])
        st, _ = g.epoch(dt[off:off + n], om[off:off + n], acc[off:off + n], pos[e], quat[e], notch[e])
        off += n
        assert torch.equal(status[e], st)
    sf, sg = f.get_state(), g.get_state()
    for key in sf:
        assert torch.equal(sf[key], sg[key]), key
    assert bool(torch.isfinite(sf["x"]).all())
    # captured: the side streams join the capture through the fork / join events
    dev = f.device
    dev_t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt_, device=dev)
    tdt, tom, tacc, tpos, tquat, tnotch = dev_t(dt), dev_t(om), dev_t(acc), dev_t(pos), dev_t(quat), dev_t(notch)
    c.reserve(int(n_pred.max()))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        c.run_epochs(n_pred, tdt, tom, tacc, tpos, tquat, tnotch)
    torch.cuda.synchronize(dev)
    gr.replay()
    torch.cuda.synchronize(dev)
    sc_ = c.get_state()
    for key in sf:
        assert torch.equal(sf[key], sc_[key]), ("captured", key)


def test_copy_instances_takes_a_slice_of_another_batch():
    """dvi_eskf_copy_instances: a small filter <- instances [first, first + b) of a large one, incl. a pending injection of
    the source; bad ranges are refused."""
    from tests.helpers import default_run
    from dvi_ekf_amd.batch import BatchedESKF
    from dvi_ekf_amd import _capi
    cfg, ri, Q, R, sa, so = default_run(golden_compat=True, fast=True)
    mk = lambda b: BatchedESKF(b, length=cfg.model.length, theta_cam=cfg.model.angle, frozen=[0] * 6, sigma_a=sa, sigma_om=so, hold_frozen=True)
    big, small = mk(77), mk(5)
    rng = np.random.default_rng(3)
    x0 = np.tile(ri.x0, (77, 1))
    x0[:, 10:16] += rng.normal(0, 0.01, (77, 6))
    big.set_noise(Q, R)
    big.set_state(x=x0, P=ri.P0, om_old=ri.om0, acc_old=ri.acc0)
    big.step(ri.dt[0], ri.om[0], ri.acc[0], ri.pos[0], ri.quat[0], ri.notch[0])      # (leaves an injection pending)
    small.copy_instances_from(big, 40)
    a, b = big.get_state(), small.get_state()
    for k in a:
        assert torch.equal(a[k][40:45], b[k]), k
    L = _capi.lib()
    assert L.dvi_eskf_copy_instances(small._h, big._h, 73, None) == -1        # 73 + 5 > 77
    assert L.dvi_eskf_copy_instances(small._h, big._h, -1, None) == -1
    assert L.dvi_eskf_copy_instances(small._h, small._h, 0, None) == -1
