"""Developer probe: N contexts encode / decode concurrently; results are compared with one context working alone."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
from khugle_amd import Codec

H = int(sys.argv[1]); W = int(sys.argv[2]); S = int(sys.argv[3]); NBIG = int(sys.argv[4]) if len(sys.argv) > 4 else 0
big = [Codec(4320, 7680) for _ in range(NBIG)]  # idle bystanders, as in the bench
ctxs = [Codec(H, W) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
cap = (2 * H * W + (1 << 20)) & ~3
frames = [ctxs[0].synth_dev(H, W, 2025, i) for i in range(S)]
ref_comp, ref_dec = [], []
solo = Codec(H, W)
for i in range(S):
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    solo.encode_dev(frames[i], 1, out)
    n = solo.encode_result()
    dec = torch.zeros((3, H, W), dtype=torch.uint8, device="cuda")
    solo.decode_dev(out, n, H, W, dec)
    assert solo.last_status() == 0
    ref_comp.append(out[:n].clone()); ref_dec.append(dec)
torch.cuda.synchronize()
for rep in range(3):
    outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(S)]
    decs = [torch.zeros((3, H, W), dtype=torch.uint8, device="cuda") for _ in range(S)]
    torch.cuda.synchronize()
    for c in range(S):
        with torch.cuda.stream(streams[c]):
            ctxs[c].encode_dev(frames[c], 1, outs[c])
    ns = [ctxs[c].encode_result() for c in range(S)]
    enc_ok = [bool(ns[c] == ref_comp[c].numel() and torch.equal(outs[c][:ns[c]], ref_comp[c])) for c in range(S)]
    for c in range(S):
        with torch.cuda.stream(streams[c]):
            ctxs[c].decode_dev(ref_comp[c], ref_comp[c].numel(), H, W, decs[c])
    sts = [ctxs[c].last_status() for c in range(S)]
    dec_ok = [bool(torch.equal(decs[c], ref_dec[c])) for c in range(S)]
    print("rep", rep, "encode identical", enc_ok, "decode status", sts, "decode identical", dec_ok, flush=True)
