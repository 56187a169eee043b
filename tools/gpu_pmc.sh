#!/bin/bash
# Developer loop on the GPU box: two counter passes over one 8K encode+decode.  Usage: tools/gpu_pmc.sh TAG
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmcA_$TAG -- python3 $R/tools/probe.py 4320 7680 1 > $R/gpurun_out/pmcA_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/pmcA_$TAG.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmcB_$TAG -- python3 $R/tools/probe.py 4320 7680 1 > $R/gpurun_out/pmcB_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/pmcB_$TAG.log; exit 1; }
echo done
