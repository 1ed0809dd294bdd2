#!/bin/bash
# A/B over library variants with common bench.py arguments (GPU box): AB_ARGS="..." tools/ab_lib_args.sh <reps> base|<name> ...
reps=$1; shift
for i in $(seq 1 $reps); do
  for n in "$@"; do
    l=""; [ "$n" != base ] && l="DVI_ESKF_LIB=$PWD/variants_tmp/lib_$n.so"
    env $l python bench.py --no-cpu-baseline --no-streaming ${AB_ARGS} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$n |',round(d['value']/1e8,3),round(d['roofline']['kernel_ms'],4),round(d['roofline']['frac'],4),d['all_instances_ok'])"
  done
done
