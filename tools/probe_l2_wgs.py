"""Developer probe for a KHG_TIMING build: per-workgroup schedule of the level-2 forward launch (the last forward launch
of an encode).  usage: probe_l2_wgs.py NWG"""
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
out = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
for _ in range(3):
    c.encode_dev(f, 1, out)
    c.encode_result()
lib = C.CDLL(khugle_amd.lib.LIB_PATH)
buf = np.zeros(8 * 8192, dtype=np.uint32)
lib.khg_debug_fwd.argtypes = [C.c_void_p, C.c_size_t]
assert lib.khg_debug_fwd(buf.ctypes.data, buf.size) == 0
n = int(sys.argv[1])
w = buf.reshape(-1, 8)[:n]
t0 = w[:, 0].min()
st = (w[:, 0] - t0).astype(np.int64) / 100.0
en = (w[:, 1] - t0).astype(np.int64) / 100.0
du = en - st
print("wgs %d start min/mean/max %.1f %.1f %.1f  end min/mean/max %.1f %.1f %.1f  dur min/mean/max %.1f %.1f %.1f" % (
    n, st.min(), st.mean(), st.max(), en.min(), en.mean(), en.max(), du.min(), du.mean(), du.max()))
print("dur percentiles 10/50/90/99: %s" % np.percentile(du, [10, 50, 90, 99]).round(1))
print("end percentiles 10/50/90/99: %s" % np.percentile(en, [10, 50, 90, 99]).round(1))
pro = (w[:, 6] - w[:, 0]).astype(np.int64) / 100.0
loop = (w[:, 7] - w[:, 6]).astype(np.int64) / 100.0
epi = (w[:, 1] - w[:, 7]).astype(np.int64) / 100.0
print("prologue mean %.1f max %.1f  loop (wave 0) mean %.1f max %.1f  epilogue mean %.1f max %.1f us" % (pro.mean(), pro.max(), loop.mean(), loop.max(), epi.mean(), epi.max()))
cyc = (w[:, 5].astype(np.int64) - w[:, 4].astype(np.int64)) & 0xFFFFFFFF
print("clock median %.2f GHz" % np.median(cyc / np.maximum(du, 1e-3) / 1e3))
