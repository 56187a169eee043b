#!/bin/bash
# On the GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) and HBM traffic (FETCH_SIZE, WRITE_SIZE, one
# --pmc pass each) of one-stream 8K encode + decode passes, summarised into profiles/<TAG>_kernels.json.
# Usage: tools/gpu_roofline.sh TAG
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/roof_$TAG
mkdir -p $O $R/profiles
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/probe_roofline.py 5 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/probe_roofline.py 2 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/probe_roofline.py 2 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
COMP=$(grep comp_bytes $O/trace.log | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['comp_bytes'])")
S=$(ls $O/trace/*/*kernel_stats.csv | head -1)
cp $S $R/gpurun_out/${TAG}_kernel_stats_one_stream_8k.csv
python3 $R/tools/roofline_kernels.py $S $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $COMP $R/gpurun_out/${TAG}_kernels.json
rm -f $O/trace/*/*kernel_trace.csv
