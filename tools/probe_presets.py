#!/usr/bin/env python3
"""One 8K frame, one stream: encode and decode time per preset (KHG_LIB selects an alternative build;
arguments: the presets to run, default 0 1 2)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
import torch  # noqa: E402
import khugle_amd.lib  # noqa: E402
if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
from khugle_amd import Codec  # noqa: E402

R, C_ = 4320, 7680
c = Codec(R, C_)
f = c.synth_dev(R, C_, 2025, 0)
out = torch.zeros(c.comp_bound(R, C_), dtype=torch.uint8, device="cuda")
dec = torch.zeros((3, R, C_), dtype=torch.uint8, device="cuda")
for lvl in ([int(x) for x in sys.argv[1:]] or [0, 1, 2]):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            c.encode_dev(f, lvl, out); n = c.encode_result()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(10):
            c.decode_dev(out, n, R, C_, dec); assert c.last_status() == 0 or os.environ.get("KHG_NOCHECK")  # timing-only builds decode wrongly
        torch.cuda.synchronize(); t2 = time.perf_counter()
    hd = khugle_amd.peek(out[:n].cpu().numpy().tobytes())
    print("%s preset %d: encode %.0f us, decode %.0f us, %d bytes, tables %d %d %d" % (
        os.path.basename(khugle_amd.lib.LIB_PATH), lvl, (t1 - t0) * 1e5, (t2 - t1) * 1e5, n, hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len), flush=True)
