#!/bin/bash
# On the GPU box: what the level-1 forward launch costs in VALU terms.
#   1. tools/bin/probe_rates       issue cost of the f64 instructions, with the shader clock held during the probe
#   2. probe_fwd_wgs.py            the clock held inside the forward launch's workgroups (KHG_TIMING build)
#   3. one --pmc pass              wave-instructions of the launch by unit
# summarised into gpurun_out/<TAG>_valu.json by tools/valu_summary.py.  Build beforehand (on the CPU):
#   hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_rates tools/probe_rates.hip
#   hipcc <the library's flags> -DKHG_TIMING -o tools/bin/libkhugle_timing.so image-compression_amd/csrc/khugle_hip.hip
# Usage: tools/gpu_valu.sh TAG
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/valu_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/bin/probe_rates > $O/rates.log 2>&1 || { tail -5 $O/rates.log; exit 1; }
KHG_LIB=$R/tools/bin/libkhugle_timing.so timeout -k 10 120 python3 $R/tools/probe_fwd_wgs.py > $O/clock.log 2>&1 || { tail -5 $O/clock.log; exit 1; }
timeout -k 10 120 python3 $R/tools/probe_fwd.py > $O/alone.log 2>&1 || { tail -5 $O/alone.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc -- python3 $R/tools/probe_fwd.py > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
python3 $R/tools/valu_summary.py $O $R/gpurun_out/${TAG}_valu.json
