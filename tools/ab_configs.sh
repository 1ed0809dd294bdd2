#!/bin/bash
# A/B over environment switches for tools/bench_configs.py (GPU box): tools/ab_configs.sh <reps> "<config args>" "ENV=VAL" ...
reps=$1; cfg=$2; shift; shift
for i in $(seq 1 $reps); do
  for s in "$@"; do
    env $s python tools/bench_configs.py $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$s', d.get('config'), d.get('batch'), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_launch','frac_of_8TBps','value','ms_per_step','pair_steps_per_s','finite')})"
  done
done
