#!/bin/bash
# Build libadmm_hip.so variants on the GPU box and time the alternating iteration with each.
# usage: tools/alt_sweep.sh "<name>:<-D flags>" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  lib="gpurun_out/variants/libadmm_hip_${name}.so"
  python __graft_entry__.py variant "$PWD/$lib" $flags > "gpurun_out/variants/${name}.build.log" 2>&1 || { echo "$name: build failed"; continue; }
  echo "== $name ($flags)"
  ADMM_HIP_LIB="$PWD/$lib" timeout -k 10 200 python tools/alt_time.py 0 1 2>&1 | tail -3
  rm -rf "$lib" "$lib.obj"
done
