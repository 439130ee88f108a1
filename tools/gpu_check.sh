#!/bin/bash
# One gpurun call that re-validates everything measured on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- tools/gpu_check.sh [profile-tag]
# 1. -m gpu parity tests  2. smoke()  3. default bench.py  4. (with a tag) the three rocprofv3 passes.
set -o pipefail
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p "$O"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$O/pytest_gpu.log"; tail -3 "$O/pytest_gpu.log"
grep -q "pytest exit 0" "$O/pytest_gpu.log" || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 600 python bench.py > "$O/bench.log" 2>&1 || { tail -5 "$O/bench.log"; exit 1; }
tail -n1 "$O/bench.log" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.3e %s | %d batch-it/s | roofline %.1f%% (%s) | zdual standalone %.1f%% | cpu %.3e on %d cores' % (
    d['value'], d['unit'], d['batch_iterations_per_s'], 100 * d['roofline']['frac'], d['roofline']['kernel'].split(' ')[0],
    100 * d['roofline_zdual_standalone']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']))"
if [ -n "$1" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf "$O/prof_stats" "$O/prof_fetch" "$O/prof_write"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_stats" -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$O/prof_stats.log" 2>&1 &&
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/prof_fetch" -- python3 "$R/bench.py" --steps 20 --warmup 2 --repeats 2 --profile-launches 40 --no-cpu-baseline > "$O/prof_fetch.log" 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/prof_write" -- python3 "$R/bench.py" --steps 20 --warmup 2 --repeats 2 --profile-launches 40 --no-cpu-baseline > "$O/prof_write.log" 2>&1 &&
  echo "profiles collected: python profiles/summarize.py $1 gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write"
fi
