#!/bin/bash
# usage (on the GPU box): tools/prof_round.sh <tag> [parts, default 1234]  -- the round's committed profile set:
#  1. kernel trace + FETCH / WRITE PMC passes of the bench workload (tools/prof_full.sh) and the traffic constant bench.py reports
#  2. (after 3, whose constants they quote) the bench lines: the driver's short call and the default; both carry the `secondary` block
#  3. SQ counters (vector-ALU issue share; condensed into profiles/valu_counters.json, which bench.py quotes) of the bench workload and of the resident regimes: 10:1 epochs at B = 65 536, config #5's
#     schedule at B = 1 024 / 8 192 in both precisions, pair steps at B = 8 192 -- with the kernel stats of the same commands
#  4. where a frame of the long-horizon kernel goes (tools/exp_epoch_stamps.py on the -DESKF_X_EPSTAMPS build, when
#     variants_tmp/lib_epstamps.so is there)
tag=$1
parts=${2:-1234}
has() { case "$parts" in *$1*) return 0;; esac; return 1; }
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
if has 1; then
bash tools/prof_full.sh ${tag}_final || exit 1
python - <<PY
import json
d = json.load(open("profiles/${tag}_final_summary.json"))
json.dump({"hbm_bytes_per_step": d["hbm_bytes_per_launch"], "read_bytes_x2_corrected": d["hbm_read_bytes_per_launch_corrected_x2"],
           "write_bytes": d["hbm_write_bytes_per_launch"], "source": "profiles/${tag}_final_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 per MI355X_MICROARCH.md)"},
          open("profiles/traffic_per_launch.json", "w"), indent=1)
PY
fi
if has 3; then
bash tools/pmc.sh ${tag}_sq "$SQ" | grep -E "run_kernel|step_kernel|pre4|cov_rows" > profiles/${tag}_sq_counters.txt
{
  echo "# SQ counters of the resident regimes (tools/pmc_cmd.sh; valu_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32))"
  bash tools/pmc_cmd.sh ${tag}_sq_epoch10 "$SQ" $PWD/tools/bench_configs.py epoch10 | grep eskf
  bash tools/pmc_cmd.sh ${tag}_sq_small "$SQ" $PWD/tools/bench_configs.py small | grep eskf
  bash tools/pmc_cmd.sh ${tag}_sq_c5 "$SQ" $PWD/tools/sweep_config5.py 1024,8192 2000 | grep eskf
  DVI_ESKF_EPOCHS=0 bash tools/pmc_cmd.sh ${tag}_sq_c5_ebe "$SQ" $PWD/tools/sweep_config5.py 1024 2000 | grep eskf | sed 's/^/[DVI_ESKF_EPOCHS=0] /'
} > profiles/${tag}_sq_resident.txt
{
  bash tools/prof_cmd.sh ${tag}_ks_epoch10 $PWD/tools/bench_configs.py epoch10
  bash tools/prof_cmd.sh ${tag}_ks_small $PWD/tools/bench_configs.py small
  bash tools/prof_cmd.sh ${tag}_ks_c5 $PWD/tools/sweep_config5.py 1024,8192 2000
} > profiles/${tag}_kernel_stats_resident.txt 2>&1
python tools/summarize_sq.py ${tag} > gpurun_out/${tag}_summarize_sq.txt
fi
if has 2; then
python bench.py --steps 20 --warmup 5 > profiles/${tag}_final_bench_short.json 2> gpurun_out/${tag}_bench_short.err || exit 1
python bench.py > profiles/${tag}_final_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
fi
if has 4 && [ -f variants_tmp/lib_epstamps.so ]; then
  {
    for cfg in "1024 f32" "1024 f64" "6144 f32" "6144 f64"; do
      STAMPS_BRIEF=1 DVI_ESKF_LIB=$PWD/variants_tmp/lib_epstamps.so python tools/exp_epoch_stamps.py $cfg 2000 2>&1 | grep -v "^dvi_ekf_amd:\|amdgpu.ids"
    done
    STAMPS_BRIEF=1 DVI_ESKF_LIB=$PWD/variants_tmp/lib_epstamps.so python tools/exp_epoch_stamps.py 8192 f32 2000 2>&1 | grep -v "^dvi_ekf_amd:\|amdgpu.ids"
  } > profiles/${tag}_epochs_kernel_stamps.txt
fi
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}_* profiles/traffic_per_launch.json profiles/valu_counters.json gpurun_out/profiles_${tag}/
has 2 && python -c "
import json
for f in ('profiles/${tag}_final_bench_short.json', 'profiles/${tag}_final_bench.json'):
    d = json.load(open(f)); print(f, '%.3e' % d['value'], d['roofline']['frac'], d['roofline']['frac_wall'], d['roofline']['traffic'], d['roofline']['streaming']['frac'] if d['roofline']['streaming'] else None, d['valid'])
"
