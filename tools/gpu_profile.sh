#!/bin/bash
# The three rocprofv3 passes of a bench.py configuration (gpurun refuses --pmc together with the trace domains):
#   tools/gpu_profile.sh <tag> [bench.py args...]       then:  python profiles/summarize.py <tag> gpurun_out/prof_<tag>_{stats,fetch,write}
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_${tag}_stats" "$O/prof_${tag}_fetch" "$O/prof_${tag}_write"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_${tag}_stats" -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline "$@" > "$O/prof_${tag}_stats.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/prof_${tag}_fetch" -- python3 "$R/bench.py" --steps 20 --warmup 2 --repeats 2 --profile-launches 40 --no-cpu-baseline "$@" > "$O/prof_${tag}_fetch.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/prof_${tag}_write" -- python3 "$R/bench.py" --steps 20 --warmup 2 --repeats 2 --profile-launches 40 --no-cpu-baseline "$@" > "$O/prof_${tag}_write.log" 2>&1 &&
{ mkdir -p "$O/profiles_out" && PROFILES_OUT="$O/profiles_out" python3 "$R/profiles/summarize.py" "$tag" "$O/prof_${tag}_stats" "$O/prof_${tag}_fetch" "$O/prof_${tag}_write" &&
  rm -rf "$O/prof_${tag}_stats" "$O/prof_${tag}_fetch" "$O/prof_${tag}_write" && echo "summaries in gpurun_out/profiles_out/${tag}_*.csv (raw rocprofv3 output deleted: > 64 MiB)"; }
