"""HIP-graph safety of the step calls (include/dvi_eskf.h, "HIP graphs"): whether an update's error
injection is still due is device state, so a captured sequence of steps replays like the same calls
made eagerly, whatever the host knew at capture time; workspace growth is refused under capture."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from dvi_ekf_amd import _capi
from tests.helpers import default_run
from tests.test_gpu_parity import _make, _perturbed_x0


def _inputs(ri, B, n, dev):
    rng = np.random.default_rng(7)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    out = []
    for k in range(n):
        out.append(dict(dt=t(np.full(B, ri.dt[k])), om=t(ri.om[k] + 1e-3 * rng.standard_normal((B, 3))),
                        acc=t(ri.acc[k] + 1e-2 * rng.standard_normal((B, 3))), pos=t(np.tile(ri.pos[k], (B, 1))),
                        quat=t(np.tile(ri.quat[k], (B, 1))), notch=t(np.full(B, ri.notch[k]))))
    return out


def _capture(fn, dev):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    torch.cuda.synchronize(dev)
    return g


@pytest.mark.parametrize("pending_at_capture", [False, True])
def test_captured_step_pair_replayed_three_times_equals_six_eager_steps(pending_at_capture):
    """graph = [step, step]; three replays must leave, bit for bit, the state of six eager steps -- also
    when the handle had (or had not) an injection pending while the graph was recorded: the first step of
    every replay has to inject what the previous replay's last update left behind."""
    B = 70
    dev = torch.device("cuda", 0)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False, frozen=(0,) * 6)
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, False, x0)
    inp = _inputs(ri, B, 3, dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)

    def step(flt, i):
        a = inp[i]
        flt.step(a["dt"], a["om"], a["acc"], a["pos"], a["quat"], a["notch"], status=status)

    if pending_at_capture:
        step(f, 2)                        # leaves an injection pending on the handle
    e = f.clone()                         # (flushes f: the host now believes nothing is due ...)
    if pending_at_capture:
        step(f, 2)                        # ... and this makes something due again, on both
        step(e, 2)
    # recording does not execute: the graph is built from f's CURRENT state
    g = _capture(lambda: (step(f, 0), step(f, 1)), dev)
    for _ in range(3):
        g.replay()
    for _ in range(3):
        step(e, 0)
        step(e, 1)
    torch.cuda.synchronize(dev)
    a, b = f.get_state(), e.get_state()
    for key in a:
        assert torch.equal(a[key], b[key]), key
    assert bool(torch.isfinite(a["x"]).all())
    # ... and once more after the state has been looked at (flush) in between
    g.replay()
    step(e, 0)
    step(e, 1)
    a, b = f.get_state(), e.get_state()
    for key in a:
        assert torch.equal(a[key], b[key]), key


def test_captured_predict_update_sequence_with_state_reads_in_between():
    """update ; predict ; update captured as separate calls, with the calibration read inside the graph (a
    flush in the middle of the recorded sequence)."""
    B = 33
    dev = torch.device("cuda", 0)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False, frozen=(0,) * 6)
    f = _make(cfg, ri, Q, R, sa, so, B, False, _perturbed_x0(ri, B))
    e = f.clone()
    inp = _inputs(ri, B, 2, dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    dofs_f = torch.empty((B, 6), dtype=torch.float64, device=dev)
    lib = _capi.lib()
    import ctypes as C
    P = lambda t: C.c_void_p(t.data_ptr())

    def seq(flt, out):
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        a, b = inp
        _capi.check(lib.dvi_eskf_update(flt._h, P(a["pos"]), P(a["quat"]), P(a["notch"]), 1, None, P(st), s), "update")
        _capi.check(lib.dvi_eskf_get_calibration(flt._h, P(out), s), "calibration")
        _capi.check(lib.dvi_eskf_predict(flt._h, P(b["dt"]), 1, P(b["om"]), P(b["acc"]), 1, s), "predict")
        _capi.check(lib.dvi_eskf_update(flt._h, P(b["pos"]), P(b["quat"]), P(b["notch"]), 1, None, P(st), s), "update")

    g = _capture(lambda: seq(f, dofs_f), dev)
    dofs_e = torch.empty_like(dofs_f)
    for _ in range(2):
        g.replay()
        seq(e, dofs_e)
        torch.cuda.synchronize(dev)
        assert torch.equal(dofs_f, dofs_e)
    a, b = f.get_state(), e.get_state()
    for key in a:
        assert torch.equal(a[key], b[key]), key


def test_workspace_growth_is_refused_under_capture_and_reserve_fixes_it():
    B = 9
    dev = torch.device("cuda", 0)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False, frozen=(0,) * 6)
    f = _make(cfg, ri, Q, R, sa, so, B, False)
    e = f.clone()
    n = 5
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    dt, om, acc = t(ri.dt[:n]), t(ri.om[:n]), t(ri.acc[:n])
    pos, quat, notch = t(ri.pos[0]), t(ri.quat[0]), t(ri.notch[0])
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    lib = _capi.lib()
    import ctypes as C
    P = lambda x: C.c_void_p(x.data_ptr())
    codes = []

    def epoch(flt):
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        codes.append(lib.dvi_eskf_epoch(flt._h, n, P(dt), 0, P(om), P(acc), 0, P(pos), P(quat), P(notch), 0, None, P(st), s))

    _capture(lambda: epoch(f), dev)          # a 5-predict epoch on a handle that never ran one: needs an allocation
    assert codes[-1] == -5                   # DVI_ESKF_ECAPTURE; nothing was recorded, nothing hung
    f.reserve(n)
    g = _capture(lambda: epoch(f), dev)
    assert codes[-1] == 0
    g.replay()
    epoch(e)
    assert codes[-1] == 0
    a, b = f.get_state(), e.get_state()
    for key in a:
        assert torch.equal(a[key], b[key]), key


def test_buffers_a_graph_refers_to_are_not_replaced_after_capture():
    """Rule (4) of the header: once a step of a handle has been captured, a longer eager epoch / reserve (the
    Jacobian workspace would be re-allocated under the graph), set_noise on a per-instance-noise handle (that
    buffer would be freed) and a change of keep_jacobians (the record format is a recorded argument) are refused
    with DVI_ESKF_ECAPTURE -- and the graph still replays correctly afterwards."""
    B = 12
    dev = torch.device("cuda", 0)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False, frozen=(0,) * 6)
    f = _make(cfg, ri, Q, R, sa, so, B, False)
    f.set_noise_per_instance(np.tile(np.diag(Q), (B, 1)), np.tile(np.diag(R), (B, 1)))
    e = f.clone()
    inp = _inputs(ri, B, 1, dev)[0]
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    lib = _capi.lib()
    import ctypes as C
    P = lambda x: C.c_void_p(x.data_ptr())

    def step(flt):
        flt.step(inp["dt"], inp["om"], inp["acc"], inp["pos"], inp["quat"], inp["notch"], status=st)

    g = _capture(lambda: step(f), dev)
    n = 4
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    dt, om, acc = t(ri.dt[:n]), t(ri.om[:n]), t(ri.acc[:n])
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.dvi_eskf_epoch(f._h, n, P(dt), 0, P(om), P(acc), 0, P(inp["pos"]), P(inp["quat"]), P(inp["notch"]), 1, None, P(st), s)
    assert rc == -5
    assert lib.dvi_eskf_reserve(f._h, n) == -5
    dp = C.POINTER(C.c_double)
    q = np.ascontiguousarray(Q, dtype=np.float64)
    assert lib.dvi_eskf_set_noise(f._h, q.ctypes.data_as(dp), None) == -5
    assert lib.dvi_eskf_keep_jacobians(f._h, 1) == -5
    assert lib.dvi_eskf_keep_jacobians(f._h, 0) == 0          # (no change: fine)
    assert b"captured" in lib.dvi_eskf_strerror(-5)
    g.replay()
    step(e)
    torch.cuda.synchronize(dev)
    a, b = f.get_state(), e.get_state()
    for key in a:
        assert torch.equal(a[key], b[key]), key


@pytest.mark.parametrize("from_snapshot", [False, True])
def test_captured_run_in_one_launch_replays_like_the_eager_calls(from_snapshot):
    """dvi_eskf_run_epochs / run_epochs_from with one predict per epoch -- one launch of eskf_run_kernel, the time loop
    in the kernel -- recorded into a HIP graph: three replays of a 3-epoch run equal nine eager steps (continuing: every
    replay's first step injects what the last step of the replay before left pending, which is device state); started
    from a snapshot handle every replay restarts there and equals one eager run."""
    B = 200
    dev = torch.device("cuda", 0)
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=False, frozen=(0,) * 6)
    x0 = _perturbed_x0(ri, B)
    f = _make(cfg, ri, Q, R, sa, so, B, False, x0)
    E = 3
    inp = _inputs(ri, B, E, dev)
    stack = lambda key: torch.stack([a[key] for a in inp]).contiguous()
    dt, om, acc, pos, quat, notch = (stack(k) for k in ("dt", "om", "acc", "pos", "quat", "notch"))
    ones = np.ones(E, dtype=np.int32)
    st = torch.zeros((E, B), dtype=torch.int32, device=dev)
    snap = f.clone()
    e = f.clone()
    g = _capture(lambda: f.run_epochs(ones, dt, om, acc, pos, quat, notch, src=snap if from_snapshot else None, status=st), dev)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(dev)
    for rep in range(1 if from_snapshot else 3):
        for i in range(E):
            a = inp[i]
            e.step(a["dt"], a["om"], a["acc"], a["pos"], a["quat"], a["notch"])
    sf, se = f.get_state(), e.get_state()
    for key in sf:
        assert torch.equal(sf[key], se[key]), key
    assert int(st.max()) == 0
