#!/bin/bash
# A/B harness over environment switches of ONE library build (run on the GPU box): alternates bench.py between
# the settings given as arguments, e.g.  tools/ab_env.sh 3 "DVI_ESKF_PRE_LEGACY=1" "DVI_ESKF_PRE_LEGACY=0"
reps=$1; shift
for i in $(seq 1 $reps); do
  for s in "$@"; do
    env $s python bench.py --no-cpu-baseline --no-streaming ${AB_ARGS} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$s',round(d['value']/1e8,3),round(d['roofline']['kernel_ms'],4),round(d['roofline']['frac_wall'],4),d['all_instances_ok'])"
  done
done
