#!/bin/bash
# A/B harness over bench.py argument sets (GPU box): tools/ab_args.sh <reps> "<args A>" "<args B>" ...   (ENV=VAL words in front of the args are exported)
reps=$1; shift
for i in $(seq 1 $reps); do
  for s in "$@"; do
    e=""; a=""
    for w in $s; do case "$w" in [A-Z_]*=*) e="$e $w";; *) a="$a $w";; esac; done
    env $e python bench.py --no-cpu-baseline --no-streaming $a 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$s |',round(d['value']/1e8,3),round(d['roofline']['kernel_ms'],4),round(d['roofline']['frac'],4),round(d['roofline']['frac_wall'],4),d['all_instances_ok'])"
  done
done
