#!/bin/bash
# Developer loop on the GPU box: profile the 8K probe with alternative builds of the library.
# Usage: tools/gpu_variant.sh TAG LIB [LIB...]   (libraries built beforehand into tools/bin/)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  KHG_LIB=$R/tools/bin/$L timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_${TAG}_$N -- python3 $R/tools/probe.py > $R/gpurun_out/var_${TAG}_$N.log 2>&1 || { tail -20 $R/gpurun_out/var_${TAG}_$N.log; exit 1; }
  echo "== $N: $(grep bytes $R/gpurun_out/var_${TAG}_$N.log)"
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/var_${TAG}_$N/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if any(k in r["Name"] for k in "${KHG_VAR_FILTER:-fwd,inv}".split(",")):
        print("   %-50s calls %4s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
