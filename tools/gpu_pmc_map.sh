#!/bin/bash
# On the GPU box: instruction counters and durations of the decode kernels for an alternative build.  Usage: tools/gpu_pmc_map.sh TAG LIB.so
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcmap_$1
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export KHG_LIB=$R/tools/bin/$2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p -- python3 $R/tools/probe_presets.py 1 > $O/p.log 2>&1 || { tail -5 $O/p.log; exit 1; }
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/probe_presets.py 1 > $O/t.log 2>&1 || { tail -5 $O/t.log; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
dur = {}
for f in glob.glob("$O/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0][:40]] = float(r["AverageNs"]) / 1e3
for k in sorted(acc):
    if "dec_map_kernel<32" in k or "dec_write" in k or "packscan" in k or "inv_kernel" in k or "fwd_kernel" in k:
        print("$1 %-34s %6.1f us" % (k, dur.get(k, 0)), {c: int(acc[k][c] / n[k][c]) for c in sorted(acc[k])})
PY
