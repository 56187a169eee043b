"""Developer probe for the HBM-traffic counter passes: the luma forward kernel plus two kernels of known byte
counts (the frame generator writes 3 bytes per pixel, the PSNR kernel reads 6) to calibrate the counters."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
from khugle_amd import Codec

H, W = 4320, 7680
c = Codec(H, W)
a = c.synth_dev(H, W, 2025, 0)
b = c.synth_dev(H, W, 2025, 1)
print("psnr", c.psnr_dev(a, b))
print("luma forward ms", c.time_luma_forward_dev(a, 1, reps=5))
torch.cuda.synchronize()
