#!/bin/bash
# A/B harness: alternates bench.py between prebuilt library variants variants_tmp/lib_<name>.so (run on the GPU box)
# usage: tools/ab_bench.sh <reps> <name> [<name> ...]
reps=$1; shift
cp dvi_ekf_amd/libdvi_eskf.so /tmp/lib_orig.so
for i in $(seq 1 $reps); do
  for d in "$@"; do
    cp variants_tmp/lib_$d.so dvi_ekf_amd/libdvi_eskf.so
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$d',round(d['value']/1e8,3),round(d['roofline']['kernel_ms'],4),d['valid'])"
  done
done
cp /tmp/lib_orig.so dvi_ekf_amd/libdvi_eskf.so
