#!/bin/bash
# usage (on the GPU box): tools/prof_variants.sh <name> [<name> ...] -- kernel stats of bench.py for each prebuilt
# library variant variants_tmp/lib_<name>.so (built with `make -C dvi_ekf_amd/csrc -B EXTRA=... OUT=../../variants_tmp/lib_<name>.so`)
for n in "$@"; do bash tools/prof_variant.sh v_$n variants_tmp/lib_$n.so 2>&1 | grep -E "pre|cov_rows|value"; done
