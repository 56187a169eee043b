#!/usr/bin/env python3
"""One 8K frame through the kernels outside the bench's timed path (format-2 index, pictures, thumbnails, metrics), for
a rocprofv3 --kernel-trace --stats run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
import torch  # noqa: E402
from khugle_amd import Codec  # noqa: E402

R, C_, LVL = 4320, 7680, 1
c = Codec(R, C_)
f = c.synth_dev(R, C_, 2025, 0)
out = torch.zeros(c.comp_bound(R, C_) + c.index_bound(R, C_), dtype=torch.uint8, device="cuda")
dec = torch.zeros((3, R, C_), dtype=torch.uint8, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    c.set_format(2)
    c.encode_dev(f, LVL, out)
    n = c.encode_result()
    c.set_format(1)
    c.decode_indexed_dev(out, n, R, C_, dec)
    assert c.last_status() == 0
    c.vis_coeffs_dev(f)
    c.vis_dequant_dev(out, n, R, C_)
    c.vis_steps_dev(R, C_, LVL)
    c.vis_bars_dev(R, C_, (7000000, 1000000, 900000))
    for ch in range(3):
        c.downsample_dev(f[ch])
    c.ssim_dev(f[0], dec[0])
    c.psnr_dev(f, dec)
torch.cuda.synchronize()
print("done", n)
