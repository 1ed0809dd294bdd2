"""Soak of the long-horizon kernel's hand-over protocol (LDS counters, rings, mailboxes): random batches, schedules (zeros and
ones included where the comparison is bitwise: schedules of >= 2 predicts per epoch, or 0), precisions and frozen masks, each
run compared bit for bit with the same epochs enqueued one by one (tests/_epochs_child.py's scenarios, in-process).  The block
size follows DVI_ESKF_EPOCHS_NC / _PP of the environment.  usage: soak_epochs.py [seconds] [seed]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import importlib.util
import numpy as np
import torch
saved = sys.argv
sys.argv = [saved[0], "0", "f64", root, "2"]
spec = importlib.util.spec_from_file_location("_epochs_child", os.path.join(root, "tests", "_epochs_child.py"))
child = importlib.util.module_from_spec(spec)
spec.loader.exec_module(child)
sys.argv = saved
rng = np.random.default_rng(seed)
t_end, n = time.time() + secs, 0
while time.time() < t_end:
    B = int(rng.choice([1, 3, 5, 64, 130, 515, 1000, 1500, 2500, 4097, 6144, 7424, 8192]))
    dtype = torch.float64 if rng.random() < 0.5 else torch.float32
    if B > 6144 and dtype == torch.float64 and os.environ.get("DVI_ESKF_EPOCHS") != "1":
        continue                # (not the long-horizon kernel's by default: the scenario's bitwise checks are about that kernel)
    E = int(rng.integers(1, 9))
    sched = [int(v) for v in rng.choice([0, 2, 3, 5, 9, 17, 33], size=E)]
    if sum(sched) == 0:
        sched[0] = 2
    frozen = tuple(int(v) for v in rng.integers(0, 2, 6)) if rng.random() < 0.3 else (0,) * 6
    hold = bool(rng.random() < 0.7)
    try:
        child.main(B, dtype, sched, frozen, hold, rows_closeness=False)
    except AssertionError:
        print("soak: FAILED at B %d %s schedule %s frozen %s hold %s" % (B, str(dtype)[-7:], sched, frozen, hold), flush=True)
        raise
    n += 1
    if n % 10 == 0:
        print("soak: %d runs bitwise equal (last: B %d %s %s)" % (n, B, str(dtype)[-7:], sched), flush=True)
print("soak ok: %d runs" % n)
