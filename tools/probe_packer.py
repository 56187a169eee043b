"""Developer probe for a KHG_TIMING build: per-entry schedule of the single-pass packer on an 8K frame."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import khugle_amd.lib

if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
import torch
from khugle_amd import Codec
from khugle_amd import lib as L

H, W = 4320, 7680
c = Codec(H, W)
frame = c.synth_dev(H, W, 2025, 0)
out = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
for _ in range(3):
    c.encode_dev(frame, 1, out)
    n = c.encode_result()
lib = L.load()
ne = 3042
buf = np.zeros(8 * ne, np.uint64)
lib.khg_debug_packer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert lib.khg_debug_packer(c._ctx, buf.ctypes.data, buf.size) == 0
r = buf.reshape(ne, 8).astype(np.int64)
ok = r[:, 4] > 0
t0 = r[ok, 0].min()
us = (r - t0) / 100.0
print("entries with a whole-path end stamp", ok.sum(), "of", ne, " span %.1f us" % us[ok, 4].max())
names = ["codes+scan", "look-back", "puts+barrier", "write"]
for k in range(4):
    d = us[ok, k + 1] - us[ok, k]
    print("  %-14s p50 %6.1f p90 %6.1f max %6.1f" % (names[k], np.median(d), np.percentile(d, 90), d.max()))
idx = np.nonzero(ok)[0]
for i in idx[:: max(1, idx.size // 40)]:
    print("  entry %5d start %7.1f | %s | end %7.1f" % (i, us[i, 0], " ".join("%6.1f" % (us[i, k + 1] - us[i, k]) for k in range(4)), us[i, 4]))
