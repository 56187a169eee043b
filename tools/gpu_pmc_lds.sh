#!/bin/bash
# On the GPU box: LDS counters of the one-stream 8K pass per kernel (one --pmc pass).  Usage: tools/gpu_pmc_lds.sh TAG
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmclds_$1
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/p -- python3 $R/tools/probe_roofline.py 2 > $O/p.log 2>&1 || { tail -5 $O/p.log; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if "khg" in k:
        print("%-42s" % k, {c: int(acc[k][c] / n[k][c]) for c in sorted(acc[k])})
PY
