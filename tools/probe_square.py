"""Developer probe: encode + decode square frames one at a time and report status and parity of the round trip."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
from khugle_amd import Codec

side = int(sys.argv[1]) if len(sys.argv) > 1 else 5760
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
c = Codec(side, side)
c2 = Codec(side, side)
out = torch.zeros((2 * side * side + (1 << 20)) & ~3, dtype=torch.uint8, device="cuda")
dec = torch.zeros((3, side, side), dtype=torch.uint8, device="cuda")
dec2 = torch.zeros((3, side, side), dtype=torch.uint8, device="cuda")
for i in range(n):
    f = c.synth_dev(side, side, 2025, i)
    c.encode_dev(f, 1, out)
    ln = c.encode_result()
    c.decode_dev(out, ln, side, side, dec)
    st = c.last_status()
    c2.decode_dev(out, ln, side, side, dec2)
    st2 = c2.last_status()
    print(i, "len", ln, "status", st, st2, "same", bool(torch.equal(dec, dec2)), flush=True)
