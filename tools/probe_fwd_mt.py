"""Developer probe: throughput cost of the level-1 forward launch when several streams keep the GPU saturated with it
(no dependencies between launches of different streams): wall time / launches."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import khugle_amd.lib
if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
from khugle_amd import Codec

H, W, REPS = 4320, 7680, 40
for nt in (1, 2, 4, 8):
    cs = [Codec(H, W) for _ in range(nt)]
    fs = [c.synth_dev(H, W, 2025, i) for i, c in enumerate(cs)]
    ss = [torch.cuda.Stream() for _ in range(nt)]
    torch.cuda.synchronize()
    out = [0.0] * nt

    def work(i):
        with torch.cuda.stream(ss[i]):
            out[i] = cs[i].time_luma_forward_dev(fs[i], 1, reps=REPS)

    for rnd in range(2):
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%d streams: %.1f us per launch (wall / launches), per-stream event average %.1f us" % (
        nt, 1e6 * dt / (nt * (REPS + 1)), 1e3 * sum(out) / nt))
    del cs, fs
