"""one scenario of tests/_epochs_child.py by hand.  usage: dbg_soak_case.py <B> <f64|f32> <schedule> <frozen mask> <hold 0|1> [repeats]"""
import os, sys, importlib.util
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
args = sys.argv[1:]
sys.argv = [sys.argv[0], "0", "f64", root, "2"]
spec = importlib.util.spec_from_file_location("_epochs_child", os.path.join(root, "tests", "_epochs_child.py"))
child = importlib.util.module_from_spec(spec); spec.loader.exec_module(child)
B, dt, sched = int(args[0]), (torch.float64 if args[1] == "f64" else torch.float32), [int(v) for v in args[2].split(",")]
frozen, hold = tuple(int(c) for c in args[3]), bool(int(args[4]))
for r in range(int(args[5]) if len(args) > 5 else 1):
    try:
        child.main(B, dt, sched, frozen, hold, rows_closeness=False)
    except AssertionError as ex:
        print("FAILED:", ex)
