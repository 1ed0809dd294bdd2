"""the ctypes stub a maintainer of the reference would put next to
dvi_ekf/filter/Filter.py is EXECUTED here as it is printed there (only the library path is made absolute) and
driven the way the reference's Filter would drive it."""
import os
import re

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle
from tests.helpers import default_run

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "class HipBackend" in b)
    assert 'C.CDLL("libdvi_eskf.so")' in code
    code = code.replace('C.CDLL("libdvi_eskf.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "dvi_ekf_amd", "libdvi_eskf.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    return ns


class _ReferenceFilterLike:
    """the attributes of the reference's Filter that the stub reads"""

    def __init__(self, cfg, ri, Q, R, sa, so):
        from dvi_ekf_amd.state import State
        self._probe_length, self._probe_angle = cfg.model.length, cfg.model.angle
        self._frozen_dofs = list(cfg.frozen_dofs)
        self.stdev_na, self.stdev_nom = np.asarray(sa, dtype=float), np.asarray(so, dtype=float)
        self.Q, self.R = Q, R
        self._states = State.from_vector(ri.x0)
        self._P = ri.P0.copy()
        self.om_old, self.acc_old = ri.om0.copy(), ri.acc0.copy()


def test_integration_stub_b_runs_the_reference_schedule():
    cfg, ri, Q, R, sa, so = default_run(golden_compat=False, fast=True)     # v1-literal: hold_frozen = 0 in the stub
    ns = _stub_namespace()
    kf = _ReferenceFilterLike(cfg, ri, Q, R, sa, so)
    be = ns["HipBackend"](kf)
    prm = c_oracle.make_params(cfg.model.length, cfg.model.angle, cfg.frozen_dofs, False, sa, so, Q, R)
    o = c_oracle.OracleBatch(prm, ri.x0[None, :], ri.P0, ri.om0, ri.acc0)
    ip = 0
    for k in range(4):
        for _ in range(ri.n_pred[k]):
            be.propagate(ri.dt[ip], ri.om[ip], ri.acc[ip])
            o.predict(ri.dt[ip], ri.om[ip], ri.acc[ip])
            ip += 1
        K = be.update(ri.pos[k], ri.quat[k], float(ri.notch[k]))
        _, Ko = o.update_one(0, ri.pos[k], ri.quat[k], ri.notch[k])
        assert K is not None and K.shape == (24, 7)
        assert np.abs(K - Ko).max() <= 1e-8 * max(1.0, np.abs(Ko).max())
    x = torch.empty((1, 26), dtype=torch.float64, device="cuda")
    P = torch.empty((1, 24, 24), dtype=torch.float64, device="cuda")
    lib, vp = ns["_lib"], ns["_vp"]
    assert lib.dvi_eskf_get_state(be.h, vp(x.data_ptr()), vp(P.data_ptr()), None, None, None, ns["_stream"]()) == 0
    assert np.abs(x[0].cpu().numpy() - o.x[0]).max() <= 1e-9
    assert np.abs(P[0].cpu().numpy() - o.P[0]).max() <= 1e-9 * np.abs(o.P[0]).max()
