#!/bin/bash
# On the GPU box: per-kernel durations of one-stream 8K encode + decode passes at one preset.
# Usage: tools/gpu_preset_profile.sh PRESET TAG
set -o pipefail
L=${1:-0}; TAG=${2:-preset$L}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 python3 $R/tools/probe_presets.py $L > $O/plain.log 2>&1 || { tail -20 $O/plain.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/probe_presets.py $L > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
S=$(ls $O/trace/*/*kernel_stats.csv | head -1)
cp $S $O/kernel_stats.csv
rm -f $O/trace/*/*kernel_trace.csv
cat $O/plain.log; head -18 $O/kernel_stats.csv
