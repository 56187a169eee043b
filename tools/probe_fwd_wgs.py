"""Developer probe for a KHG_TIMING build: per-workgroup schedule of the level-1 forward launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import numpy as np
import torch
import khugle_amd.lib
if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
from khugle_amd import Codec

H, W = 4320, 7680
c = Codec(H, W)
f = c.synth_dev(H, W, 2025, 0)
for _ in range(2):
    ms = c.time_luma_forward_dev(f, 1, reps=5)
print("level-1 forward launch: %.1f us" % (ms * 1e3))
lib = C.CDLL(khugle_amd.lib.LIB_PATH)
buf = np.zeros(8 * 8192, dtype=np.uint32)
lib.khg_debug_fwd.argtypes = [C.c_void_p, C.c_size_t]
assert lib.khg_debug_fwd(buf.ctypes.data, buf.size) == 0
w = buf.reshape(-1, 8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 688
w = w[:n]
live = w[:, 0] != 0
print("workgroups with stamps:", int(live.sum()), "of", n)
nl = int(live[:nl].sum())
w = w[live]
n = w.shape[0]
t0 = w[:, 0].min()
st = (w[:, 0] - t0).astype(np.int64) / 100.0  # us
en = (w[:, 1] - t0).astype(np.int64) / 100.0
du = en - st
for name, sl in (("luma", slice(0, nl)), ("chroma", slice(nl, n))):
    print("%-6s wgs %4d  start min/mean/max %.1f %.1f %.1f  end min/mean/max %.1f %.1f %.1f  dur min/mean/max %.1f %.1f %.1f" % (
        name, du[sl].size, st[sl].min(), st[sl].mean(), st[sl].max(), en[sl].min(), en[sl].mean(), en[sl].max(),
        du[sl].min(), du[sl].mean(), du[sl].max()))
hw = w[:, 2]
cu = (hw >> 8) & 15
sh_ = (hw >> 12) & 1
se = (hw >> 13) & 7
xcc = w[:, 3] & 15
key = xcc.astype(np.int64) * 4096 + se * 256 + sh_ * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs", u.size, "wgs per CU histogram", np.bincount(cnt))
# per CU: mix of luma/chroma
mix = {}
for i in range(n):
    mix.setdefault(int(key[i]), []).append(("L" if i < nl else "C", round(float(du[i]), 1)))
for k in list(mix)[:12]:
    print(k, mix[k])
print("span of the launch %.1f us" % (en.max() - st.min()))
pro = (w[:, 6] - w[:, 0]).astype(np.int64) / 100.0
loop = (w[:, 7] - w[:, 6]).astype(np.int64) / 100.0
epi = (w[:, 1] - w[:, 7]).astype(np.int64) / 100.0
print("prologue mean %.1f max %.1f  loop (wave 0) mean %.1f max %.1f  epilogue mean %.1f max %.1f us" % (pro.mean(), pro.max(), loop.mean(), loop.max(), epi.mean(), epi.max()))
for name, sl in (("luma", slice(0, nl)), ("chroma", slice(nl, n))):
    print("  %-6s prologue %.1f  loop %.1f  epilogue %.1f us (means)" % (name, pro[sl].mean(), loop[sl].mean(), epi[sl].mean()))
cyc = (w[:, 5].astype(np.int64) - w[:, 4].astype(np.int64)) & 0xFFFFFFFF
ghz = cyc / np.maximum(du, 1e-3) / 1e3
print("shader clock held inside the workgroups: median %.2f GHz (min %.2f, max %.2f)" % (np.median(ghz), ghz.min(), ghz.max()))
