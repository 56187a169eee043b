"""Per-kernel averages of the counters in rocprofv3 counter_collection.csv files (developer aid)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:44]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
print("%-44s " % "kernel" + " ".join("%14s" % n[-14:] for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0) / max(1, calls[k].get("SQ_BUSY_CYCLES", 1))):
    if not k.startswith(("khg", "void khg")):
        continue
    print("%-44s " % k + " ".join("%14.0f" % (acc[k][n] / max(1, calls[k][n])) for n in names))
