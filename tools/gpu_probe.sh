#!/bin/bash
# Developer loop on the GPU box: profiled 8K probe only (no parity tests).  Usage: tools/gpu_probe.sh TAG
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/probe.py > $R/gpurun_out/p_$TAG.log 2>&1 || { tail -30 $R/gpurun_out/p_$TAG.log; exit 1; }
grep bytes $R/gpurun_out/p_$TAG.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    if "khg" in r["Name"]:
        print("%-60s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
