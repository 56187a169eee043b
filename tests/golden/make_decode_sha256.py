"""Writes tests/golden/decode_sha256.json: SHA-256 of the decoded picture (R plane, G plane, B plane, row-major u8)
of every Run/*.bmp x preset .comp file and of the three .comp files the reference ships.

Run in the build container (needs /root/reference and oracle/_ref): the files are encoded and decoded through the
reference's own Huffman.h / Reader.h / DWT.h (oracle/_ref, kref_encode_file / kref_decode_file); dequantiser and
colour conversion around them are the oracle's restatement, pinned by the screenshots' PSNR / SSIM figures.
"""
import ctypes as C
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "image-compression_amd", "python"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle_py as O
from khugle_amd.bmp import read_bmp

IMAGES = ["baboon", "bird", "camera", "cat", "couple", "girl", "house", "lena", "library"]


def ref_decode(path):
    R = O.ref()
    hd = O.peek(open(path, "rb").read())
    out = np.zeros((3, hd.h, hd.w), np.uint8)
    n = C.c_int(0)
    rc = R.kref_decode_file(path.encode(), out[0], out[1], out[2], C.byref(n))
    assert rc == 0 and n.value == hd.h
    return out


def main():
    O.build()
    R = O.ref()
    assert R is not None, "oracle/_ref is not built (needs the reference checkout)"
    run = os.path.join(HERE, "run")
    out = {}
    tmp = os.path.join(tempfile.gettempdir(), "khg_golden_%d.comp" % os.getpid())
    for name in IMAGES:
        rgb = read_bmp(os.path.join(run, name + ".bmp"))
        for lvl in (0, 1, 2):
            assert R.kref_encode_file(rgb[0], rgb[1], rgb[2], rgb.shape[1], lvl, tmp.encode()) == 0
            out["%s_%d" % (name, lvl)] = hashlib.sha256(ref_decode(tmp).tobytes()).hexdigest()
    os.unlink(tmp)
    for f in ("cat_high", "couple_normal", "library_extreme"):
        out[f + ".comp"] = hashlib.sha256(ref_decode(os.path.join(run, f + ".comp")).tobytes()).hexdigest()
    with open(os.path.join(HERE, "decode_sha256.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote %d hashes" % len(out))


main()
