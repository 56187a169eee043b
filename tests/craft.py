"""Hand-built .comp files for decoder tests (layout of Writer.h:9-69)."""
import struct

import numpy as np


def pack_codes(symbols, table_syms, codes, lens):
    """Concatenate the codes of `symbols` MSB-first; returns (bytes, bit count, pad)."""
    slot = {int(s): i for i, s in enumerate(table_syms)}
    bits = []
    for s in symbols:
        i = slot[int(s)]
        c, n = int(codes[i]), int(lens[i])
        bits.extend((c >> (n - 1 - k)) & 1 for k in range(n))
    nbits = len(bits)
    pad = (8 - nbits % 8) % 8
    bits.extend([0] * pad)
    arr = np.packbits(np.array(bits, np.uint8)) if bits else np.zeros(0, np.uint8)
    return arr.tobytes(), nbits + pad, pad


def make_comp(rows, cols, lvl, planes):
    """planes: three (counts, syms, symbol sequence) tuples -> .comp bytes (codes from the oracle)."""
    import oracle_py as O
    streams, tables, lens, pads, ks = [], [], [], [], []
    for counts, syms, seq in planes:
        codes, ln = O.build_codes(counts, syms)
        data, nbits, pad = pack_codes(seq, syms, codes, ln)
        streams.append(data)
        lens.append(nbits)
        pads.append(pad)
        ks.append(len(counts))
        tables.append(b"".join(struct.pack("<ii", int(c), int(s)) for c, s in zip(counts, syms)))
    header = struct.pack("<8i4B", lens[0], lens[1], lens[2], ks[0], ks[1], ks[2], rows, cols,
                         pads[0], pads[1], pads[2], lvl)
    return header + b"".join(streams) + b"".join(tables)


def fixed_length_comp(rows, cols, lvl, seed=0):
    """Luma coded with eight equiprobable symbols (all codes 3 bits: a decoder that starts at a
    wrong bit offset never resynchronises); chroma planes hold the single symbol 0."""
    rng = np.random.default_rng(seed)
    n = rows * cols
    syms = np.array([0, 1, -1, 2, -2, 3, -3, 4], np.int32)
    seq = syms[rng.integers(0, 8, n)]
    counts = np.array([int((seq == s).sum()) for s in syms], np.int32)
    # equal counts keep every code at 3 bits; the counts need not match the stream for decoding
    counts[:] = n // 8
    zero = (np.array([n // 4], np.int32), np.array([0], np.int32), [])
    return make_comp(rows, cols, lvl, [(counts, syms, seq), zero, zero])


def long_code_comp(rows, cols, lvl, depth=30, seed=0):
    """Luma table with Fibonacci counts: code lengths 1..depth (beyond both decode-table levels);
    every symbol occurs, the rare ones a few times.  Chroma planes hold the single symbol 0."""
    rng = np.random.default_rng(seed)
    n = rows * cols
    fib = [1, 1]
    while len(fib) < depth + 1:
        fib.append(fib[-1] + fib[-2])
    counts = np.array(fib[:depth + 1], np.int64)
    syms = np.arange(depth + 1, dtype=np.int32) - depth // 2
    p = counts / counts.sum()
    seq = syms[rng.choice(depth + 1, n, p=p)]
    seq[rng.choice(n, 4 * (depth + 1), replace=False)] = np.tile(syms, 4)  # every code length shows up
    zero = (np.array([n // 4], np.int32), np.array([0], np.int32), [])
    return make_comp(rows, cols, lvl, [(counts.astype(np.int32), syms, seq), zero, zero])


def stress_images(h, w):
    """Pictures that push the path to its corners: one flat colour (single-entry tables, empty streams), white
    noise (hundreds of symbols, long codes, large detail coefficients), a black/white checkerboard (the largest
    high-pass values 8-bit input can produce) and vertical bars of extreme colours (large chroma)."""
    import numpy as np
    rng = np.random.default_rng(h * 1000 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    flat = np.full((3, h, w), 77, np.uint8)
    noise = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    chk = np.broadcast_to((((yy + xx) & 1) * 255).astype(np.uint8), (3, h, w)).copy()
    bars = np.zeros((3, h, w), np.uint8)
    bars[0] = np.where((xx // 3) % 2 == 0, 255, 0)
    bars[2] = np.where((xx // 5) % 2 == 0, 0, 255)
    return {"flat": flat, "noise": noise, "checkerboard": chk, "bars": bars}
