#!/bin/bash
# block size of the one-launch kernels over the batch size (GPU box): tools/sweep_run_rounds.sh [run|step] [batches...]
#   run:  eskf_run_kernel (bench.py default), DVI_ESKF_RUN_ROUNDS = 1..4 (16..64 instances per block)
#   step: eskf_step_kernel (bench.py --loop-in-kernel 0), DVI_ESKF_FUSED_ROUNDS = 1..4
what=${1:-run}; shift
bs=${@:-"4096 8192 12288 16384 20480 24576 28672 32768 40960 49152 57344 65536 98304 131072"}
for B in $bs; do
  if [ "$what" = run ]; then
    bash tools/ab_args.sh 1 "DVI_ESKF_RUN_ROUNDS=1 --batch $B" "DVI_ESKF_RUN_ROUNDS=2 --batch $B" "DVI_ESKF_RUN_ROUNDS=3 --batch $B" "DVI_ESKF_RUN_ROUNDS=4 --batch $B"
  else
    bash tools/ab_args.sh 1 "DVI_ESKF_FUSED_ROUNDS=1 --loop-in-kernel 0 --batch $B" "DVI_ESKF_FUSED_ROUNDS=2 --loop-in-kernel 0 --batch $B" "DVI_ESKF_FUSED_ROUNDS=3 --loop-in-kernel 0 --batch $B" "DVI_ESKF_FUSED_ROUNDS=4 --loop-in-kernel 0 --batch $B"
  fi
done
