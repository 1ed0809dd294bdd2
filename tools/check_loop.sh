#!/bin/bash
# (GPU box; for a build with profiles/r03_cov_loop_experiment.patch applied) the looping covariance kernel against the one-pair-per-wave kernel, bit for bit, on the scenarios of
# tests/_pre4_child.py:  tools/check_loop.sh <rounds> [scenario ...]
r=$1; shift
for sc in "${@:-unfrozen}"; do
  python tests/_pre4_child.py $sc /tmp/loop_a.npz $PWD || exit 1
  DVI_ESKF_COV_ROUNDS=$r DVI_ESKF_COV_ROUNDS_MINBLOCKS=1 python tests/_pre4_child.py $sc /tmp/loop_b.npz $PWD || exit 1
  python - <<PY || exit 1
import numpy as np, sys
a, b = np.load("/tmp/loop_a.npz"), np.load("/tmp/loop_b.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print("$sc rounds=$r:", len(a.files), "arrays,", "all bitwise equal" if not bad else "DIFFERENT: %s" % bad)
sys.exit(1 if bad else 0)
PY
done
