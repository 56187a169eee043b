"""Developer probe: the level-1 forward launch alone (HIP events), for builds whose results may be wrong on purpose
(experiments that delete a part of the kernel to price it).  KHG_LIB selects the build."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
import khugle_amd.lib
if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
from khugle_amd import Codec

H, W = 4320, 7680
c = Codec(H, W)
f = c.synth_dev(H, W, 2025, 0)
c.time_luma_forward_dev(f, 1, reps=20)
ts = sorted(c.time_luma_forward_dev(f, 1, reps=20) * 1e3 for _ in range(int(os.environ.get("KHG_ROUNDS", "7"))))
print("level-1 forward launch: min %.1f median %.1f max %.1f us" % (ts[0], ts[len(ts) // 2], ts[-1]))
