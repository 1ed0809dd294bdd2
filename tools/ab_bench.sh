#!/bin/bash
# A/B harness: alternates bench.py between prebuilt library variants variants_tmp/lib_<name>.so (run on the GPU box)
# usage: tools/ab_bench.sh <reps> <name> [<name> ...]      (the variant is selected per process: DVI_ESKF_LIB)
reps=$1; shift
for i in $(seq 1 $reps); do
  for d in "$@"; do
    DVI_ESKF_LIB=$PWD/variants_tmp/lib_$d.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$d',round(d['value']/1e8,3),round(d['roofline']['kernel_ms'],4),d['all_instances_ok'])"
  done
done
