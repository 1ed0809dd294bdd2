"""Generates tests/golden/leaf_vectors.json from the reference's pure-NumPy/scipy leaf
modules (dvi_ekf/tools/math.py, tools/Quaternion.py, filter/state.py), loaded by file path
with their parent packages stubbed so that dvi_ekf/__init__.py (which needs casadi etc.) is
not executed.  Run in the build container only (needs /root/reference); the JSON it writes
is the committed fixture.  scipy here is 1.15.3 (reference pins 1.10.1): the two differ only
in Rotation.from_matrix on NON-orthogonal input, which is therefore not sampled here (that
path is pinned by the saved trajectory instead).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "leaf_vectors.json")


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


for pkg in ("dvi_ekf", "dvi_ekf.tools", "dvi_ekf.filter"):
    m = types.ModuleType(pkg)
    m.__path__ = []
    sys.modules[pkg] = m
math_mod = _load("dvi_ekf.tools.math", "dvi_ekf/tools/math.py")
quat_mod = _load("dvi_ekf.tools.Quaternion", "dvi_ekf/tools/Quaternion.py")
state_mod = _load("dvi_ekf.filter.state", "dvi_ekf/filter/state.py")
Q, State, ErrorState = quat_mod.Quaternion, state_mod.State, state_mod.ErrorState

rng = np.random.default_rng(20261004)
L = lambda a: np.asarray(a, dtype=float).tolist()
out = {"skew": [], "quat_mul": [], "normalise": [], "angle_axis": [], "about_axis": [], "rot": [], "from_euler_z": [],
       "from_matrix_orthonormal": [], "error_state": [], "state_add": []}

for _ in range(8):
    x = rng.normal(size=3)
    out["skew"].append({"x": L(x), "S": L(math_mod.skew(x))})
    a, b = rng.normal(size=4), rng.normal(size=4)
    qa, qb = Q(val=a), Q(val=b)
    out["quat_mul"].append({"a": L(a), "b": L(b), "ab": L((qa * qb).xyzw)})
    qn = Q(val=a, do_normalise=True)
    out["normalise"].append({"q": L(a), "n": L(qn.xyzw)})
    small = Q(val=np.array([*(rng.normal(size=3) * 0.2), 1.0]), do_normalise=True)
    out["angle_axis"].append({"q": L(small.xyzw), "angle": float(small.angle), "axis": L(small.axis)})
    ang, ax = float(rng.uniform(-2, 2)), rng.normal(size=3) * rng.choice([1.0, 1e-3])
    out["about_axis"].append({"angle": ang, "axis": L(ax), "q": L(Q.about_axis(ang, ax).xyzw)})
    out["rot"].append({"q": L(a), "R": L(Q(val=a).rot)})
    z = float(rng.uniform(-3, 3))
    out["from_euler_z"].append({"z": z, "q": L(Q(val=np.array([0, 0, z]), euler="xyz").xyzw)})
    Rm = Q(val=b, do_normalise=True).rot
    out["from_matrix_orthonormal"].append({"R": L(Rm), "q": L(Q(val=Rm, do_normalise=True).xyzw)})
    e = rng.normal(size=24) * np.r_[np.full(6, 0.1), np.full(3, 0.02), np.full(9, 0.1), np.full(3, 0.1), np.full(3, 0.02)]
    es = ErrorState.from_array(e)
    out["error_state"].append({"e": L(e), "dq": L(es.dq.xyzw), "dqc": L(es.dqc.xyzw)})
    s = State(rng.normal(size=3), rng.normal(size=3), Q(val=rng.normal(size=4), do_normalise=True), rng.normal(size=6),
              rng.normal(size=3), rng.normal(size=3), Q(val=rng.normal(size=4), do_normalise=True))
    s2 = s + es
    flat = lambda st: L(np.concatenate([st.p, st.v, st.q.xyzw, st.dofs, st.notch_dofs, st.p_cam, st.q_cam.xyzw]))
    out["state_add"].append({"x": flat(s), "e": L(e), "x_plus": flat(s2)})
# the vector quoted in SURVEY.md 8(c)
es = ErrorState.from_array(np.arange(24) * 1e-3)
out["error_state"].append({"e": L(np.arange(24) * 1e-3), "dq": L(es.dq.xyzw), "dqc": L(es.dqc.xyzw)})
json.dump(out, open(OUT, "w"), indent=0)
print("wrote", OUT, {k: len(v) for k, v in out.items()})
