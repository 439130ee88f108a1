#!/bin/bash
# A/B of wide per-instance variants on the GPU box: tools/pinst_ab.sh "<batch> <horizon>" <variant|-> [bench args]  (variant "-" = in-tree library)
cd "$(dirname "$0")/.."
set -- $1 "${@:2}"
b=$1; n=$2; v=$3; shift 3
[ "$v" != "-" ] && export ADMM_HIP_LIB=$PWD/variants/libadmm_hip_$v.so
timeout -k 10 280 python bench.py --workload cw_formation_perinstance --batch $b --horizon $n --no-cpu-baseline "$@" 2> gpurun_out/ab_$v.err | tail -n1 > gpurun_out/ab_$v.json
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab_$v.json").read())
    print("$v $*", "%dx%d" % ($b, $n), round(d["batch_iterations_per_s"], 1), "it/s", "frac", round(d["roofline"]["frac"], 3), {k: round(x, 3) for k, x in d["roofline"]["avg_launch_ms"].items()}, "S", d["config"]["segments"], "scan", round(d["roofline"]["xscan_ms"] or 0, 4))
except Exception as e:
    print("$v FAILED", e); print(open("gpurun_out/ab_$v.err").read()[-800:])
PY
