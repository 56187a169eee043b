#!/usr/bin/env python3
"""Decode rate of 8K frames with and without the seek index (format 2), one stream and a pipelined batch; also what
the index costs the encoder and the file."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
import torch  # noqa: E402
from khugle_amd import Codec  # noqa: E402

ROWS, COLS, LVL, NCTX, FRAMES = 4320, 7680, 1, 24, 96
ctxs = [Codec(ROWS, COLS) for _ in range(NCTX)]
streams = [torch.cuda.Stream() for _ in range(NCTX)]
c0 = ctxs[0]
frame = c0.synth_dev(ROWS, COLS, 2025, 0)
cap = c0.comp_bound(ROWS, COLS) + c0.index_bound(ROWS, COLS)
files = {}
for fmt in (1, 2):
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    c0.set_format(fmt)
    c0.encode_dev(frame, LVL, out)
    n = c0.encode_result()
    files[fmt] = (out, n)
    # encoder cost, one stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        c0.encode_dev(frame, LVL, out)
        c0.encode_result()
    torch.cuda.synchronize()
    print("format %d: %9d bytes, encode %.1f us per frame (one stream)" % (fmt, n, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
c0.set_format(1)
assert files[2][0][:files[1][1]].equal(files[1][0][:files[1][1]])
decs = [torch.zeros((3, ROWS, COLS), dtype=torch.uint8, device="cuda") for _ in range(NCTX)]


def run(indexed, nctx, frames):
    out, n = files[2]
    n_use = n if indexed else files[1][1]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        i = f % nctx
        c = ctxs[i]
        with torch.cuda.stream(streams[i]):
            if f >= nctx:
                assert c.last_status() == 0
            (c.decode_indexed_dev if indexed else c.decode_dev)(out, n_use, ROWS, COLS, decs[i])
    for i in range(min(nctx, frames)):
        assert ctxs[i].last_status() == 0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames


for indexed in (False, True):
    run(indexed, NCTX, NCTX)
    one = run(indexed, 1, 20)
    many = run(indexed, NCTX, FRAMES)
    mp = ROWS * COLS / 1e6
    print("%-8s decode: %.1f us per frame alone, %.1f us per frame pipelined (%d contexts) = %.0f MP/s" % (
        "indexed" if indexed else "ordinary", one * 1e6, many * 1e6, NCTX, mp / many), flush=True)
want = torch.zeros_like(decs[0]); got = torch.zeros_like(decs[0])
c0.decode_dev(files[1][0], files[1][1], ROWS, COLS, want); assert c0.last_status() == 0
c0.decode_indexed_dev(files[2][0], files[2][1], ROWS, COLS, got); assert c0.last_status() == 0
print("same pixels:", bool(torch.equal(want, got)))
