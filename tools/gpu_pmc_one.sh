#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcs_$1
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
[ -n "$2" ] && export KHG_LIB=$R/tools/bin/$2
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/p -- python3 $R/tools/probe_fwd.py > $O/p.log 2>&1 || { tail -5 $O/p.log; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(int)
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "khg_fwd_kernel<0" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("$1", {k: int(acc[k] / n[k]) for k in sorted(acc)})
PY
