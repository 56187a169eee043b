"""Developer probe for a KHG_TIMING build: per-workgroup schedule of the decode map and write kernels."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import khugle_amd.lib

if os.environ.get("KHG_LIB"):  # the KHG_TIMING build of the library
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
import torch
from khugle_amd import Codec
from khugle_amd import lib as L

H, W = 4320, 7680
c = Codec(H, W)
frame = c.synth_dev(H, W, 2025, 0)
out = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
dec = torch.zeros((3, H, W), dtype=torch.uint8, device="cuda")
c.encode_dev(frame, int(os.environ.get("KHG_PRESET", "1")), out)
n = c.encode_result()
for _ in range(2):
    c.decode_dev(out, n, H, W, dec)
    assert c.last_status() == 0
lib = L.load()
blk0 = (C.c_uint32 * 3)()
nb = C.c_uint32()
buf = np.zeros(16 << 20, np.uint8)
lib.khg_debug_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
rc = lib.khg_debug_blocks(c._ctx, buf.ctypes.data, buf.size, blk0, C.byref(nb))
rec = buf[: nb.value * 256].view(np.uint32).reshape(nb.value, 64)
pad = rec[:, 58:64]
act = np.nonzero(pad[:, 1])[0]
t0 = pad[act, 0].min()
print("active", act.size, "blk0", list(blk0), "nblocks", nb.value)
s = (pad[act, 0] - t0) / 100.0
e = (pad[act, 1] - t0) / 100.0
print("map: span %.1f us; start p50 %.1f max %.1f; dur p50 %.1f p90 %.1f max %.1f" % (
    e.max(), np.median(s), s.max(), np.median(e - s), np.percentile(e - s, 90), (e - s).max()))
order = np.argsort(s)
m1 = (pad[act, 4] - t0) / 100.0
m2 = (pad[act, 5] - t0) / 100.0
for i in order[:: max(1, act.size // 40)]:
    print("  blk %5d start %7.1f dur %7.1f | lane0: first+main %7.1f rejoin+resolve %7.1f rest-of-block %7.1f" % (
        act[i], s[i], e[i] - s[i], m1[i] - s[i], m2[i] - m1[i], e[i] - m2[i]))
w0 = pad[act, 2].min()
ws = (pad[act, 2] - w0) / 100.0
wd = pad[act, 3] / 100.0
print("write: span %.1f us; start max %.1f; dur p50 %.1f p90 %.1f max %.1f" % ((ws + wd).max(), ws.max(), np.median(wd), np.percentile(wd, 90), wd.max()))
