#!/bin/bash
# Developer loop on the GPU box: counter passes over the level-1 forward launch alone (tools/probe_fwd.py).
# Usage: tools/gpu_pmc_fwd.sh TAG [library under tools/bin]
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$2" ] && export KHG_LIB=$R/tools/bin/$2
O=$R/gpurun_out/pmcfwd_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  n=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $O/p$n -- python3 $R/tools/probe_fwd.py > $O/p$n.log 2>&1 || { tail -5 $O/p$n.log; exit 1; }
}
pass 1 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR
pass 2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass 3 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass 4 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass 5 TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_EA0_RDREQ_sum
pass 6 SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS
pass 7 TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_WRITE_sum TCC_BUSY_sum
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(int)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "khg_fwd_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print("%-40s %16.0f  (per launch, %d launches)" % (k, acc[k] / n[k], n[k]))
PY
