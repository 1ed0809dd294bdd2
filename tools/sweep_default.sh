#!/bin/bash
# the launcher's own block-size choice over the batch size (GPU box), next to tools/sweep_run_rounds.sh: tools/sweep_default.sh [run|step]
what=${1:-run}
for B in 4096 8192 12288 16384 20480 24576 28672 32768 40960 49152 57344 65536 98304 131072; do
  if [ "$what" = run ]; then x=""; else x="--loop-in-kernel 0"; fi
  DVI_ESKF_VERBOSE=1 python bench.py --no-cpu-baseline --no-streaming $x --batch $B 2> gpurun_out/_sd.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print($B, round(d['value']/1e8,3), d['all_instances_ok'], end=' ')"
  grep -o "([0-9]* instances per block)" gpurun_out/_sd.err | sort -u | tr '\n' ' '; echo
done
