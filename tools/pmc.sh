#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> "<counter> <counter> ..." [env settings...]  -- one rocprofv3 --pmc pass over
# the bench workload (counters in their own run, no trace domains), mean per kernel printed and kept in gpurun_out/<tag>/pmc.txt
tag=$1; ctr=$2; shift; shift
R=$PWD
mkdir -p gpurun_out/$tag
for s in "$@"; do export "$s"; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/$tag/pmc -- python3 $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-streaming --no-secondary --sustain-seconds 0.05 > /dev/null 2> $R/gpurun_out/$tag/pmc.err) || exit 1
python - <<PY | tee gpurun_out/$tag/pmc.txt
import collections, csv, glob
fn = glob.glob("gpurun_out/$tag/pmc/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fn)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "eskf" not in k: continue
    print("$tag", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
rm -rf gpurun_out/$tag/pmc
