"""Static instruction mix of a kernel's loops from hipcc -S output (developer aid).
usage: isa_loop_stats.py file.s mangled_kernel_name_substring"""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") or (l.startswith("_Z") and name in l and ": " in l))
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
body = txt[start:end]
def cat(op):
    if op.startswith(("v_readlane", "v_writelane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")): return "vmem"
    return "other"
# regions between loop headers
cur = "prologue"; stats = collections.OrderedDict()
lab = None
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: lab = m.group(1)
    m = re.search(r"(Loop Header|Inner Loop Header): Depth=(\d+)", l)
    if m and lab: cur = "%s depth %s" % (lab, m.group(2))
    m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
    if m: cur = ".L%s depth %s" % (m.group(1), m.group(2))
    s = l.strip()
    if not s or s.startswith((";", ".")) or s.endswith(":"): continue
    op = s.split()[0]
    stats.setdefault(cur, collections.Counter())[cat(op)] += 1
for k, c in stats.items():
    print("%-22s" % k, " ".join("%s %4d" % (n, c[n]) for n in ("valu", "salu", "branch", "lane", "wait", "lds", "vmem")), " total", sum(c.values()))
