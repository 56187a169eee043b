#!/usr/bin/env python3
"""Timing of the ordered-mean kernel (khg_ordered_mean_dev) on inputs that exercise its paths differently."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
import torch  # noqa: E402
from khugle_amd.lib import load  # noqa: E402

L = load()
rng = np.random.default_rng(1)
N = 518400
cases = {
    "ones (integer windows only)": np.ones(N),
    "blocks like a frame's": np.clip(0.85 + 0.2 * rng.random(N), -1, 1),
    "normal(0,1) (sign changes)": rng.standard_normal(N),
    "ones, 8 windows": np.ones(65536),
    "ones, 1 window": np.ones(8192),
}
for name, x in cases.items():
    d = torch.from_numpy(x).cuda()
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    def run():
        assert L.khg_ordered_mean_dev(C.c_void_p(d.data_ptr()), d.numel(), C.c_void_p(out.data_ptr()), None) == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    s = 0.0
    for v in x.tolist():
        s += v
    ok = (s / len(x)) == float(out.cpu()[0])
    print("%-32s n=%7d  %8.1f us  exact=%s" % (name, len(x), e0.elapsed_time(e1) * 1000 / 20, ok), flush=True)
