#!/bin/bash
# End-of-round measurements on one GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- bash tools/round_end.sh <tag>
# smoke, the default bench.py (timed), the driver-style short run, then the three rocprofv3 passes of the headline and of
# configs[4]; bench JSON lines and profile summaries land in gpurun_out/ (copy what is kept into profiles/).
set -o pipefail
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; tag=${1:-rXX}; mkdir -p "$O/profiles_out"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:warnings > "$O/${tag}_pytest_gpu.log" 2>&1; prc=$?; tail -1 "$O/${tag}_pytest_gpu.log"
[ $prc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
t0=$(date +%s)
timeout -k 10 600 python bench.py > "$O/${tag}_bench.log" 2>&1 || { tail -5 "$O/${tag}_bench.log"; exit 1; }
echo "default bench.py: $(( $(date +%s) - t0 )) s wall"
tail -n1 "$O/${tag}_bench.log" > "$O/profiles_out/${tag}_bench.json"
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > "$O/${tag}_bench_driver.log" 2>&1 || { tail -5 "$O/${tag}_bench_driver.log"; exit 1; }
echo "driver-style bench.py --steps 20 --warmup 5: $(( $(date +%s) - t0 )) s wall"
tail -n1 "$O/${tag}_bench_driver.log" > "$O/profiles_out/${tag}_bench_steps20.json"
python - "$O/profiles_out/${tag}_bench.json" "$O/profiles_out/${tag}_bench_steps20.json" <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    r = d["roofline"]
    print("%s: %.0f batch-it/s (%.4f ms/step), roofline %.3f (%s %.1f / %.1f us), ci10 %.0f, cpu %s" % (
        f.split("/")[-1], d["batch_iterations_per_s"], d["ms_per_step"], r["frac"], r["unit"],
        1e3 * r["per_kernel"]["xfze"]["avg_launch_ms"], 1e3 * r["per_kernel"]["xbze"]["avg_launch_ms"],
        d["check_interval_10"]["batch_iterations_per_s"], d.get("cpu_baseline", {}).get("value")))
PY
bash tools/gpu_profile.sh "$tag" || exit 1
timeout -k 10 600 python bench.py --workload cw_formation --precision mixed --no-cpu-baseline > "$O/${tag}_formation_mixed_bench.log" 2>&1 || { tail -5 "$O/${tag}_formation_mixed_bench.log"; exit 1; }
tail -n1 "$O/${tag}_formation_mixed_bench.log" > "$O/profiles_out/${tag}_formation_mixed_bench.json"
MFMA=1 bash tools/gpu_profile.sh "${tag}_formation_mixed" --workload cw_formation --precision mixed || exit 1
for cfg in "4096 1000" "64 200"; do set -- $cfg
  timeout -k 10 300 python bench.py --workload cw_perinstance --batch $1 --horizon $2 --no-cpu-baseline 2>/dev/null | tail -n1 > "$O/profiles_out/${tag}_perinstance_${1}x${2}_bench.json" || exit 1
done
# per-instance dynamics at n = 12 (the wide shapes' kernels): bench lines, then kernel stats + HBM PMC passes of the full-size case
for cfg in "4096 1000" "1024 1000" "64 200"; do set -- $cfg
  timeout -k 10 300 python bench.py --workload cw_formation_perinstance --batch $1 --horizon $2 --no-cpu-baseline 2>/dev/null | tail -n1 > "$O/profiles_out/${tag}_perinstance_n12_${1}x${2}_bench.json" || exit 1
done
bash tools/gpu_profile.sh "${tag}_perinstance_n12" --workload cw_formation_perinstance --batch 4096 --horizon 1000 || exit 1
python - "$O/profiles_out" "$tag" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(f"{sys.argv[1]}/{sys.argv[2]}_perinstance_*bench.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], "%.1f batch-it/s, roofline %.3f, S %d" % (d["batch_iterations_per_s"], d["roofline"]["frac"], d["config"]["segments"]), d["roofline"]["avg_launch_ms"])
PY
ls -la "$O/profiles_out"
