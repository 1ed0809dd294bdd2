#!/bin/bash
# usage (on the GPU box): tools/prof_round.sh <tag>  -- the round's committed profile set: kernel trace + FETCH / WRITE PMC passes of
# the bench workload (tools/prof_full.sh), the traffic constant bench.py reports refreshed from it, THEN the bench lines
# (driver's short call and the default), the secondary configurations and the SQ counters of the step kernels
tag=$1
bash tools/prof_full.sh ${tag}_final || exit 1
python - <<PY
import json
d = json.load(open("profiles/${tag}_final_summary.json"))
json.dump({"hbm_bytes_per_launch": d["hbm_bytes_per_launch"], "read_bytes_x2_corrected": d["hbm_read_bytes_per_launch_corrected_x2"],
           "write_bytes": d["hbm_write_bytes_per_launch"], "source": "profiles/${tag}_final_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 per MI355X_MICROARCH.md)"},
          open("profiles/traffic_per_launch.json", "w"), indent=1)
PY
python bench.py --steps 20 --warmup 5 > profiles/${tag}_final_bench_short.json 2> gpurun_out/${tag}_bench_short.err || exit 1
python bench.py > profiles/${tag}_final_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
python tools/bench_configs.py 2> gpurun_out/${tag}_configs.err | python -c "
import sys, json
print(json.dumps([json.loads(l) for l in sys.stdin if l.startswith('{')], indent=1))" > profiles/${tag}_secondary_configs.json
bash tools/pmc.sh ${tag}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" | grep -E "run_kernel|step_kernel|pre4|cov_rows" > profiles/${tag}_sq_counters.txt
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}_* profiles/traffic_per_launch.json gpurun_out/profiles_${tag}/
python -c "
import json
for f in ('profiles/${tag}_final_bench_short.json', 'profiles/${tag}_final_bench.json'):
    d = json.load(open(f)); print(f, '%.3e' % d['value'], d['roofline']['frac'], d['roofline']['frac_wall'], d['roofline']['traffic'], d['roofline']['streaming']['frac'] if d['roofline']['streaming'] else None, d['valid'])
"
