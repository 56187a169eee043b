"""Developer probe: host time spent inside khg_encode_dev / khg_decode_dev (submission cost, no waiting)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
from khugle_amd import Codec

H, W = 4320, 7680
S = 8
ctxs = [Codec(H, W) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
frame = ctxs[0].synth_dev(H, W, 2025, 0)
outs = [torch.zeros(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(S)]
decs = [torch.zeros((3, H, W), dtype=torch.uint8, device="cuda") for _ in range(S)]
for rep in range(3):
    torch.cuda.synchronize()
    ts = []
    for c in range(S):
        with torch.cuda.stream(streams[c]):
            a = time.perf_counter()
            ctxs[c].encode_dev(frame, 1, outs[c])
            ts.append(time.perf_counter() - a)
    a = time.perf_counter()
    ns = [ctxs[c].encode_result() for c in range(S)]
    wait = time.perf_counter() - a
    print("encode submit us:", " ".join("%.0f" % (t * 1e6) for t in ts), " wait all %.0f us" % (wait * 1e6))
    torch.cuda.synchronize()
    ts = []
    for c in range(S):
        with torch.cuda.stream(streams[c]):
            a = time.perf_counter()
            ctxs[c].decode_dev(outs[c], ns[c], H, W, decs[c])
            ts.append(time.perf_counter() - a)
    a = time.perf_counter()
    for c in range(S):
        assert ctxs[c].last_status() == 0
    wait = time.perf_counter() - a
    print("decode submit us:", " ".join("%.0f" % (t * 1e6) for t in ts), " wait all %.0f us" % (wait * 1e6))
