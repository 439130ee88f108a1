#!/bin/bash
# Build libadmm_hip.so variants on the GPU box and bench each (A/B of tuning constants).
# usage: tools/variant_sweep.sh "<name>:<-D flags>" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  lib="gpurun_out/variants/libadmm_hip_${name}.so"
  python __graft_entry__.py variant "$PWD/$lib" $flags > "gpurun_out/variants/${name}.build.log" 2>&1
  ADMM_HIP_LIB="$PWD/$lib" timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline ${BENCH_ARGS} > "gpurun_out/variants/${name}.json" 2> "gpurun_out/variants/${name}.err" || echo "variant $name failed"
  tail -n1 "gpurun_out/variants/${name}.json" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['batch_iterations_per_s']), d['kernels_ms']['fused_resid'], 'plain', d['kernels_ms']['fused_plain']['xb_ms'], d['kernels_ms']['fused_plain']['xfz_ms'], 'ci10', round(d['check_interval_10']['batch_iterations_per_s']), 'zdual_ms', d['kernels_ms']['unfused_resid']['zdual_ms'])"
  rm -rf "$lib" "$lib.obj"
done
