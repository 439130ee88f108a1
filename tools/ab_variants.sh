#!/bin/bash
# A/B of prebuilt development variants (tools/dev_variant.sh) on the GPU box: for each name, the default iteration at full size
# with residuals every iteration and every 10th, and admm_profile of the alternating pair with / without residuals.
#   tools/ab_variants.sh <name> ...        (ALT_ARGS="<segments>" to force a segment count)
cd "$(dirname "$0")/.."
for name in "$@"; do
  lib="$PWD/variants/libadmm_hip_${name}.so"
  [ -f "$lib" ] || { echo "$name: no such variant"; continue; }
  echo "== $name"
  ADMM_HIP_LIB="$lib" timeout -k 10 200 python tools/alt_time.py 0 1 ${ALT_ARGS} 2>&1 | tail -3 || exit 1
  ADMM_HIP_LIB="$lib" timeout -k 10 200 python tools/alt_time.py 0 10 ${ALT_ARGS} 2>&1 | head -1 || exit 1
done
