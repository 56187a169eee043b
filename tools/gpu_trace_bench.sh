#!/bin/bash
# Developer aid: kernel timeline of a short multi-stream bench run.  Usage: tools/gpu_trace_bench.sh TAG STREAMS
set -o pipefail
TAG=${1:-x}; S=${2:-8}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/bench.py --frames 32 --steps 1 --warmup 1 --no-cpu-baseline --no-square --streams $S $3 > $R/gpurun_out/trace_$TAG.log 2>&1 || { tail -20 $R/gpurun_out/trace_$TAG.log; exit 1; }
tail -1 $R/gpurun_out/trace_$TAG.log | cut -c1-400
