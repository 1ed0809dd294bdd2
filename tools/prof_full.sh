#!/bin/bash
# usage (on the GPU box): tools/prof_full.sh <tag>   -- kernel trace + the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate
# runs, MI355X_MICROARCH.md) of the bench workload; summaries go to profiles/<tag>_* and gpurun_out/<tag>/
tag=$1
R=$PWD
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 128 --warmup 16 --no-cpu-baseline --no-streaming --no-secondary --sustain-seconds 0.05"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/trace -- python3 $R/bench.py $ARGS > $R/gpurun_out/$tag/bench_trace.json 2> $R/gpurun_out/$tag/trace.err || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$tag/pmc_fetch -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/$tag/fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$tag/pmc_write -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/$tag/write.err || exit 1
cd $R
python tools/summarize_prof.py $tag gpurun_out/$tag/trace gpurun_out/$tag/pmc_fetch gpurun_out/$tag/pmc_write > gpurun_out/$tag/summary.txt
cp profiles/${tag}_* gpurun_out/$tag/
rm -f gpurun_out/$tag/*/*/*kernel_trace.csv gpurun_out/$tag/*/*/*counter_collection.csv
tail -25 gpurun_out/$tag/summary.txt
