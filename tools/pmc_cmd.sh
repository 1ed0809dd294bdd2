#!/bin/bash
# usage (on the GPU box): tools/pmc_cmd.sh <tag> "<counter> <counter> ..." <python script and args...>  -- one rocprofv3 --pmc pass
# (counters in their own run, no trace domains) over a command; per kernel the mean of every counter, the launches, and -- when
# SQ_BUSY_CYCLES and SQ_ACTIVE_INST_VALU are among them -- the share of the chip's vector-ALU issue time the kernel used:
#   valu_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines)
# (SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts' SIMDs, SQ_BUSY_CYCLES cycles summed over the 32 shader engines:
#  MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units")
tag=$1; ctr=$2; shift; shift
R=$PWD
mkdir -p gpurun_out/$tag
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/$tag/pmc -- python3 "$@" > $R/gpurun_out/$tag/pmc_out.txt 2> $R/gpurun_out/$tag/pmc.err) || exit 1
python - <<PY | tee gpurun_out/$tag/pmc.txt
import collections, csv, glob
fn = glob.glob("gpurun_out/$tag/pmc/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fn)):
    acc[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "eskf" not in k: continue
    m = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
    extra = {}
    if m.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in m:
        extra["valu_busy"] = round(m["SQ_ACTIVE_INST_VALU"] / (8.0 * m["SQ_BUSY_CYCLES"]), 3)
    if m.get("SQ_WAVES") and "SQ_INSTS_VALU" in m:
        extra["valu_insts_per_wave"] = round(m["SQ_INSTS_VALU"] / m["SQ_WAVES"], 1)
    print("$tag", k, m, extra, "launches", len(next(iter(d.values()))))
PY
rm -rf gpurun_out/$tag/pmc
