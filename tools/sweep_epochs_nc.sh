#!/bin/bash
# the long-horizon kernel over batch and block size (DVI_ESKF_EPOCHS_NC covariance wavefronts per block) on config #5's schedule,
# and the epoch-by-epoch path (DVI_ESKF_EPOCHS=0) next to it.  usage: tools/sweep_epochs_nc.sh "<batches>" "<nc values>" [n_steps]
B=${1:-1024,2048,4096,8192}
NCS=${2:-"2 4 8 12"}
N=${3:-2000}
DVI_ESKF_EPOCHS=0 python tools/sweep_config5.py $B $N 2>&1 | grep -v amdgpu
for nc in $NCS; do DVI_ESKF_EPOCHS_NC=$nc python tools/sweep_config5.py $B $N 2>&1 | grep -v amdgpu; done
