#!/bin/bash
# A/B harness over prebuilt library variants and environment switches (run on the GPU box):
#   tools/ab_lib.sh <reps> "<name>[:ENV=VAL[,ENV=VAL...]]" ...      name = base (the committed library) or variants_tmp/lib_<name>.so
cp dvi_ekf_amd/libdvi_eskf.so /tmp/lib_orig.so
reps=$1; shift
for i in $(seq 1 $reps); do
  for s in "$@"; do
    n=${s%%:*}; e=""; [ "$s" != "$n" ] && e=$(echo "${s#*:}" | tr ',' ' ')
    if [ "$n" = base ]; then cp /tmp/lib_orig.so dvi_ekf_amd/libdvi_eskf.so; else cp variants_tmp/lib_$n.so dvi_ekf_amd/libdvi_eskf.so; fi
    env $e python bench.py --no-cpu-baseline --no-streaming 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$s',round(d['value']/1e8,3),round(d['roofline']['kernel_ms'],4),round(d['roofline']['frac_wall'],4),d['valid'])"
  done
done
cp /tmp/lib_orig.so dvi_ekf_amd/libdvi_eskf.so
