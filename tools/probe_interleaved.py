"""Developer probe: one step as two separate passes (encode all, sync, decode all) against one interleaved
pipeline (a frame's decode follows its encode on the same stream, no drain in between)."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
import khugle_amd
H, W, F = 4320, 7680, 128
S = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = 2
enc = [khugle_amd.Codec(H, W) for _ in range(S)]
dec = [khugle_amd.Codec(H, W) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
frames = torch.empty((F, 3, H, W), dtype=torch.uint8, device="cuda")
for f in range(F):
    enc[0].synth_dev(H, W, 2025, f, out=frames[f])
cap = (2 * H * W + (1 << 20)) & ~3
comps = torch.zeros((F, cap), dtype=torch.uint8, device="cuda")
decs = torch.empty((S, 3, H, W), dtype=torch.uint8, device="cuda")
lens = [0] * F
torch.cuda.synchronize()

def separate():
    for f in range(F):
        c = f % S
        if f >= S: lens[f - S] = enc[c].encode_result()
        with torch.cuda.stream(streams[c]): enc[c].encode_dev(frames[f], 1, comps[f])
    for f in range(max(0, F - S), F): lens[f] = enc[f % S].encode_result()
    torch.cuda.synchronize()
    for f in range(F):
        c = f % S
        if f >= S: assert dec[c].last_status() == 0
        with torch.cuda.stream(streams[c]): dec[c].decode_dev(comps[f], lens[f], H, W, decs[c])
    for c in range(S): assert dec[c].last_status() == 0
    torch.cuda.synchronize()

def interleaved(steps):
    pending = [None] * S
    busy = [False] * S
    def finish(c):
        g = pending[c]
        lens[g] = enc[c].encode_result()
        if busy[c]: assert dec[c].last_status() == 0
        with torch.cuda.stream(streams[c]): dec[c].decode_dev(comps[g], lens[g], H, W, decs[c])
        busy[c] = True
        pending[c] = None
    for _ in range(steps):
        for f in range(F):
            c = f % S
            if pending[c] is not None: finish(c)
            with torch.cuda.stream(streams[c]): enc[c].encode_dev(frames[f], 1, comps[f])
            pending[c] = f
    for c in range(S):
        if pending[c] is not None: finish(c)
    for c in range(S):
        if busy[c]: assert dec[c].last_status() == 0
    torch.cuda.synchronize()

separate(); interleaved(1)
for rep in range(2):
    t = time.perf_counter()
    for _ in range(K): separate()
    ts = time.perf_counter() - t
    t = time.perf_counter()
    interleaved(K)
    ti = time.perf_counter() - t
    mp = K * F * H * W / 1e6
    print("streams %d: separate passes %.1f MP/s, interleaved %.1f MP/s" % (S, mp / ts, mp / ti), flush=True)
