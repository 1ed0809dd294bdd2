#!/bin/bash
# usage (on the GPU box): tools/prof_variant.sh <tag> [variants_tmp/lib_X.so]  -- kernel stats of bench.py with a prebuilt library variant
tag=$1; lib=$2
# (the variant is selected per process -- DVI_ESKF_LIB, dvi_ekf_amd/_capi.py -- the committed library is never replaced)
[ -n "$lib" ] && export DVI_ESKF_LIB=$PWD/$lib
R=$PWD
mkdir -p gpurun_out/$tag
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-streaming > $R/gpurun_out/$tag/bench.json 2> $R/gpurun_out/$tag/bench.err)
python - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/$tag/prof/**/*_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("$tag", r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
d = json.load(open("gpurun_out/$tag/bench.json")); print("$tag", "value %.3e frac %.4f kernel_ms %.4f valid %s" % (d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["valid"]))
PY
rm -f gpurun_out/$tag/prof/*/*kernel_trace.csv
