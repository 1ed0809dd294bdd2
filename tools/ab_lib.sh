#!/bin/bash
# A/B harness over prebuilt library variants and environment switches (run on the GPU box):
#   tools/ab_lib.sh <reps> "<name>[:ENV=VAL[,ENV=VAL...]]" ...      name = base (the committed library) or variants_tmp/lib_<name>.so
# The variant is selected per process (DVI_ESKF_LIB, dvi_ekf_amd/_capi.py); the committed library is never replaced.
reps=$1; shift
for i in $(seq 1 $reps); do
  for s in "$@"; do
    n=${s%%:*}; e=""; [ "$s" != "$n" ] && e=$(echo "${s#*:}" | tr ',' ' ')
    l=""; [ "$n" != base ] && l="DVI_ESKF_LIB=$PWD/variants_tmp/lib_$n.so"
    env $l $e python bench.py --no-cpu-baseline --no-streaming ${AB_ARGS} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$s',round(d['value']/1e8,3),round(d['roofline']['kernel_ms'],4),round(d['roofline']['frac_wall'],4),d['all_instances_ok'])"
  done
done
