#!/bin/bash
set -o pipefail
cd /root/repo; O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
grep -q "pytest exit 0" $O/pytest_gpu.log || exit 1
{
for b in 1 4 8; do
  for e in 0 1; do
    if [ $e == 1 ]; then export ADMM_NO_GEMV_SCAN=1; echo "== batch $b MFMA scan"; else unset ADMM_NO_GEMV_SCAN; echo "== batch $b GEMV scan"; fi
    ALT_BATCH=$b timeout -k 10 100 python tools/mfma_time.py 0 cw_rendezvous 1 || exit 1
    ALT_BATCH=$b timeout -k 10 100 python tools/mfma_time.py 0 cw_rendezvous 0 || exit 1
  done
done
} > $O/adhoc.log 2>&1
cat $O/adhoc.log
