#!/bin/bash
# The rocprofv3 passes of a bench.py configuration (gpurun refuses --pmc together with the trace domains; counters that do not fit one
# pass go in separate passes): kernel stats, FETCH_SIZE, WRITE_SIZE, and -- MFMA=1 -- the matrix-pipe counters (busy cycles, MFMA ops).
#   [MFMA=1] tools/gpu_profile.sh <tag> [bench.py args...]   -> gpurun_out/profiles_out/<tag>_{kernel_stats,hbm_traffic[,mfma]}.csv
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_${tag}_stats" "$O/prof_${tag}_fetch" "$O/prof_${tag}_write" "$O/prof_${tag}_mbusy" "$O/prof_${tag}_mops"
PMCARGS="--steps 20 --warmup 2 --repeats 2 --min-timed-seconds 0 --profile-launches 40 --no-cpu-baseline --no-configs4"
MF=()
if [ -n "$MFMA" ]; then
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE MfmaUtil --kernel-trace --output-format csv -d "$O/prof_${tag}_mbusy" -- python3 "$R/bench.py" $PMCARGS "$@" > "$O/prof_${tag}_mbusy.log" 2>&1 &&
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d "$O/prof_${tag}_mops" -- python3 "$R/bench.py" $PMCARGS "$@" > "$O/prof_${tag}_mops.log" 2>&1 || { tail -5 "$O/prof_${tag}_mbusy.log" "$O/prof_${tag}_mops.log"; exit 1; }
  MF=("$O/prof_${tag}_mbusy" "$O/prof_${tag}_mops")
fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_${tag}_stats" -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --no-configs4 "$@" > "$O/prof_${tag}_stats.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/prof_${tag}_fetch" -- python3 "$R/bench.py" $PMCARGS "$@" > "$O/prof_${tag}_fetch.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/prof_${tag}_write" -- python3 "$R/bench.py" $PMCARGS "$@" > "$O/prof_${tag}_write.log" 2>&1 &&
{ mkdir -p "$O/profiles_out" && PROFILES_OUT="$O/profiles_out" python3 "$R/profiles/summarize.py" "$tag" "$O/prof_${tag}_stats" "$O/prof_${tag}_fetch" "$O/prof_${tag}_write" "${MF[@]}" &&
  rm -rf "$O/prof_${tag}_stats" "$O/prof_${tag}_fetch" "$O/prof_${tag}_write" "$O/prof_${tag}_mbusy" "$O/prof_${tag}_mops" && echo "summaries in gpurun_out/profiles_out/${tag}_*.csv (raw rocprofv3 output deleted: > 64 MiB)"; }
