#!/bin/bash
# Developer loop on the GPU box: FETCH_SIZE and WRITE_SIZE in separate passes.  Usage: tools/gpu_traffic.sh TAG
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic_rd_$TAG -- python3 $R/tools/probe_traffic.py > $R/gpurun_out/traffic_rd_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/traffic_rd_$TAG.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic_wr_$TAG -- python3 $R/tools/probe_traffic.py > $R/gpurun_out/traffic_wr_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/traffic_wr_$TAG.log; exit 1; }
echo done
