#!/bin/bash
# Developer loop on the GPU box: parity tests once, then the stale-state stress probe in several fresh processes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
timeout -k 10 400 python -u -m pytest $R/tests -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/r2_t0.log 2>&1 || { tail -30 $R/gpurun_out/r2_t0.log; exit 1; }
tail -2 $R/gpurun_out/r2_t0.log
for i in 1 2 3 4 5 6; do
  timeout -k 10 150 python -u $R/tools/stress_stale.py ${1:-12} >> $R/gpurun_out/r2_stress.log 2>&1 || { tail -20 $R/gpurun_out/r2_stress.log; exit 1; }
  tail -1 $R/gpurun_out/r2_stress.log
done
grep -c "differing" $R/gpurun_out/r2_stress.log
