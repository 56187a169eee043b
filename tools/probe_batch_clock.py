"""Developer probe for a KHG_TIMING build: the shader clock the level-1 forward workgroups hold while 24 streams keep
the GPU busy (encode passes back to back), against one frame encoded on an idle GPU."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import numpy as np
import torch
import khugle_amd.lib
if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
from khugle_amd import Codec

H, W, S, F = 4320, 7680, 24, int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = C.CDLL(khugle_amd.lib.LIB_PATH)
lib.khg_debug_fwd.argtypes = [C.c_void_p, C.c_size_t]


def clock(tag):
    buf = np.zeros(8 * 8192, dtype=np.uint32)
    assert lib.khg_debug_fwd(buf.ctypes.data, buf.size) == 0
    w = buf.reshape(-1, 8)[:1024]
    w = w[w[:, 0] != 0]
    du = ((w[:, 1] - w[:, 0]) & 0xFFFFFFFF).astype(np.int64) / 100.0
    cyc = (w[:, 5].astype(np.int64) - w[:, 4].astype(np.int64)) & 0xFFFFFFFF
    ok = (du > 20) & (du < 2000)
    ghz = cyc[ok] / du[ok] / 1e3
    ghz = ghz[(ghz > 0.5) & (ghz < 3.0)]  # stamps of two launches mixed in one record fall outside
    print("%s: workgroup duration median %.1f us, shader clock median %.2f GHz (p10 %.2f, p90 %.2f; %d records)" % (
        tag, np.median(du[ok]), np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90), ghz.size), flush=True)


ctxs = [Codec(H, W) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
frames = [ctxs[0].synth_dev(H, W, 2025, i) for i in range(S)]
outs = [torch.zeros(ctxs[0].comp_bound(H, W), dtype=torch.uint8, device="cuda") for _ in range(S)]
torch.cuda.synchronize()
ctxs[0].encode_dev(frames[0], 1, outs[0]); ctxs[0].encode_result()
time.sleep(0.5)
ctxs[0].encode_dev(frames[0], 1, outs[0]); ctxs[0].encode_result()
torch.cuda.synchronize()
clock("idle GPU, one frame")
for rep in range(3):
    t0 = time.perf_counter()
    for f in range(F):
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                if f: ctxs[s].encode_result()
                ctxs[s].encode_dev(frames[s], 1, outs[s])
    for s in range(S):
        ctxs[s].encode_result()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("batch pass %d: %.0f MP/s encode" % (rep, F * S * H * W / dt / 1e6), flush=True)
    clock("after %d frames on %d streams" % (F * S, S))
