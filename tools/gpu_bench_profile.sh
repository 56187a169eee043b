#!/bin/bash
# The default bench command under rocprofv3 --kernel-trace --stats (summary goes to profiles/), without the CPU baseline:
# its sixteen worker processes would be profiled too and fill the log with their shutdown messages; the plain bench
# run times it.  Usage: tools/gpu_bench_profile.sh TAG
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/benchprof_$TAG -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/benchprof_$TAG.log 2>&1 || { tail -20 $R/gpurun_out/benchprof_$TAG.log; exit 1; }
grep '"metric"' $R/gpurun_out/benchprof_$TAG.log
rm -f $R/gpurun_out/benchprof_$TAG/*/*kernel_trace.csv
