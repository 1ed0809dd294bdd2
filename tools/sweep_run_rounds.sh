for B in 4096 8192 12288 16384 24576 32768 49152 65536 131072; do
  bash tools/ab_args.sh 2 "DVI_ESKF_RUN_ROUNDS=1 --batch $B" "DVI_ESKF_RUN_ROUNDS=2 --batch $B" "DVI_ESKF_RUN_ROUNDS=4 --batch $B"
done
