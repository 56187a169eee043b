"""Developer probe: the forward launches of an 8K encode alone, repeated (KHG_LIB selects an alternative build whose
later stages may fail: only the transform kernels' durations are of interest, under rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
import khugle_amd.lib
if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
from khugle_amd import Codec

R, C_ = 4320, 7680
c = Codec(R, C_)
f = c.synth_dev(R, C_, 2025, 0)
out = torch.zeros(c.comp_bound(R, C_), dtype=torch.uint8, device="cuda")
for _ in range(10):
    c.encode_dev(f, 1, out)
    try:
        c.encode_result()
    except Exception as e:  # a timing-only build
        pass
torch.cuda.synchronize()
print("done")
