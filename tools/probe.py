"""Developer probe: a few 8K encode+decode passes (run under rocprofv3 for per-kernel times)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
import khugle_amd.lib
if os.environ.get("KHG_LIB"):  # an alternative build of the library (tools/gpu_variant.sh)
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
from khugle_amd import Codec

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4320, 7680)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
c = Codec(H, W)
frames = [c.synth_dev(H, W, 2025, i) for i in range(4)]
out = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
dec = torch.zeros((3, H, W), dtype=torch.uint8, device="cuda")
for i in range(2):
    c.encode_dev(frames[i % 4], 1, out)
    n = c.encode_result()
    c.decode_dev(out, n, H, W, dec)
    assert c.last_status() == 0
torch.cuda.synchronize()
t = time.time()
for i in range(N):
    c.encode_dev(frames[i % 4], 1, out)
    n = c.encode_result()
torch.cuda.synchronize()
te = (time.time() - t) / N
t = time.time()
for i in range(N):
    c.decode_dev(out, n, H, W, dec)
    assert c.last_status() == 0
torch.cuda.synchronize()
td = (time.time() - t) / N
print("bytes %d  encode %.3f ms (%.0f MP/s)  decode %.3f ms (%.0f MP/s)" % (n, te * 1e3, H * W / te / 1e6, td * 1e3, H * W / td / 1e6))
