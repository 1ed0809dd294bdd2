#!/bin/bash
# usage (on the GPU box): tools/prof_cmd.sh <tag> <python script and args...>  -- rocprofv3 kernel stats of a command
tag=$1; shift
R=$PWD
mkdir -p gpurun_out/$tag
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 "$@" > $R/gpurun_out/$tag/out.txt 2> $R/gpurun_out/$tag/err.txt)
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/$tag/prof/**/*_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("$tag", r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us", r["Percentage"], "%")
PY
cat gpurun_out/$tag/out.txt | cut -c1-600
rm -f gpurun_out/$tag/prof/*/*kernel_trace.csv
