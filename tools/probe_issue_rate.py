"""Developer probe: how much of a bench pass the host thread spends inside the enqueue calls (24 streams, 8K frames)."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
import khugle_amd
H, W, F, S = 4320, 7680, 64, 24
ctxs = [khugle_amd.Codec(H, W) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
frames = torch.empty((F, 3, H, W), dtype=torch.uint8, device="cuda")
for f in range(F):
    ctxs[0].synth_dev(H, W, 2025, f, out=frames[f])
cap = (2 * H * W + (1 << 20)) & ~3
comps = torch.zeros((F, cap), dtype=torch.uint8, device="cuda")
decs = torch.empty((S, 3, H, W), dtype=torch.uint8, device="cuda")
lens = [0] * F
torch.cuda.synchronize()
for rep in range(3):
    t_enq = t_wait = 0.0
    t0 = time.perf_counter()
    for f in range(F):
        c = f % S
        if f >= S:
            a = time.perf_counter(); lens[f - S] = ctxs[c].encode_result(); t_wait += time.perf_counter() - a
        a = time.perf_counter()
        with torch.cuda.stream(streams[c]):
            ctxs[c].encode_dev(frames[f], 1, comps[f])
        t_enq += time.perf_counter() - a
    for f in range(max(0, F - S), F):
        lens[f] = ctxs[f % S].encode_result()
    torch.cuda.synchronize()
    te = time.perf_counter() - t0
    d_enq = d_wait = 0.0
    t0 = time.perf_counter()
    for f in range(F):
        c = f % S
        if f >= S:
            a = time.perf_counter(); assert ctxs[c].last_status() == 0; d_wait += time.perf_counter() - a
        a = time.perf_counter()
        with torch.cuda.stream(streams[c]):
            ctxs[c].decode_dev(comps[f], lens[f], H, W, decs[c])
        d_enq += time.perf_counter() - a
    for c in range(S):
        assert ctxs[c].last_status() == 0
    torch.cuda.synchronize()
    td = time.perf_counter() - t0
    print("encode pass %.1f ms: enqueue %.1f ms (%.0f us/frame), waits %.1f ms | decode pass %.1f ms: enqueue %.1f ms (%.0f us/frame), waits %.1f ms"
          % (te * 1e3, t_enq * 1e3, t_enq / F * 1e6, t_wait * 1e3, td * 1e3, d_enq * 1e3, d_enq / F * 1e6, d_wait * 1e3))
