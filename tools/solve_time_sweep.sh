#!/bin/bash
# GPU box: -m gpu tests, then tools/solve_time.py over thread counts (synchronous refactors) and with the background refactors.
set -o pipefail
cd /root/repo; O=gpurun_out; 
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
grep -q "pytest exit 0" $O/pytest_gpu.log || exit 1
nproc > $O/solve_time.log
for w in "cw_rendezvous fp64" "cw_formation fp64_mfma" "cw_formation mixed"; do set -- $w
  for th in 1 4 16; do ADMM_NO_SPECULATE=1 ADMM_FACTOR_THREADS=$th timeout -k 10 200 python tools/solve_time.py --workload $1 --precision $2 --repeats 1 >> $O/solve_time.log 2>&1 || exit 1; done
  ADMM_SPEC_DEBUG=1 timeout -k 10 200 python tools/solve_time.py --workload $1 --precision $2 --repeats 3 >> $O/solve_time.log 2>&1 || exit 1
done
cat $O/solve_time.log
