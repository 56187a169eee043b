"""Developer probe for tools/gpu_roofline.sh: a few 8K encode + decode passes on ONE stream plus the quality
metrics and two kernels of known byte counts (the frame generator writes 3 bytes per pixel, the PSNR kernel reads
6) that calibrate the HBM counters.  Prints the figures the per-kernel byte model needs (.comp length, longest code)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
from khugle_amd import Codec

H, W, N = 4320, 7680, int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = Codec(H, W)
frames = [c.synth_dev(H, W, 2025, i) for i in range(2)]
out = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
dec = torch.zeros((3, H, W), dtype=torch.uint8, device="cuda")
for i in range(N):
    c.encode_dev(frames[i % 2], 1, out)
    n = c.encode_result()
    c.decode_dev(out, n, H, W, dec)
    assert c.last_status() == 0
c.ssim_dev(frames[0][0], dec[0])
c.psnr_dev(frames[0], dec)
torch.cuda.synchronize()
print(json.dumps({"rows": H, "cols": W, "comp_bytes": n}))
