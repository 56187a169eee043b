"""Developer aid: concurrency summary of a rocprofv3 kernel_trace.csv (per-kernel mean duration, queues, busy union)."""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1], recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
t0 = rows[0]["s"]
# split into phases by kernel family
def fam(n):
    if "khg_dec" in n or "khg_inv" in n or "khg_rgb" in n: return "dec"
    if "khg_fwd" in n or "khg_table" in n or "khg_lens" in n or "khg_scan" in n or "khg_pack" in n: return "enc"
    return "other"
for phase in ("enc", "dec"):
    ks = [r for r in rows if fam(r["Kernel_Name"]) == phase]
    # the last pass = second half of dispatches
    ks = ks[len(ks) // 2:]
    span = max(r["e"] for r in ks) - min(r["s"] for r in ks)
    ev = sorted([(r["s"], 1) for r in ks] + [(r["e"], -1) for r in ks])
    busy = 0; depth = 0; last = None; wsum = 0
    for t, d in ev:
        if depth > 0: busy += t - last; wsum += depth * (t - last)
        depth += d; last = t
    print("%s: span %.2f ms, busy union %.2f ms, mean concurrency while busy %.2f, queues %s" % (
        phase, span / 1e6, busy / 1e6, wsum / max(1, busy), sorted({r["Queue_Id"] for r in ks})))
    per = defaultdict(list)
    for r in ks: per[r["Kernel_Name"][:40]].append(r["e"] - r["s"])
    tot = sum(sum(v) for v in per.values())
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("   %-40s n %4d mean %8.1f us  sum %8.2f ms (%4.1f%%)" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
