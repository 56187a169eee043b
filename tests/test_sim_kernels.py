"""The HIP kernel bodies, compiled with -DKHG_SIM and run sequentially on the CPU, against the
oracle: checks indexing, arithmetic order and the launch plan without a GPU.  (The simulator is
test infrastructure; the product library has no CPU path.)"""
import os

import numpy as np
import pytest

from conftest import RUN
import oracle_py as O
import simlib as S
import craft
from khugle_amd import synth_frame

SHAPES = [(8, 8), (16, 24), (24, 16), (64, 64), (264, 520), (520, 264), (256, 512)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_forward_stages_and_bytes(shape, lvl):
    rgb = synth_frame(shape[0], shape[1], 3, lvl)
    rc, coef, q = S.forward(rgb, lvl)
    assert rc == 0
    for g, w in zip(coef, O.forward_planes(rgb)):
        assert np.array_equal(g, w)
    for g, w in zip(q, O.symbols(rgb, lvl)):
        assert np.array_equal(g.astype(np.int32), w)
    rc, comp = S.encode(rgb, lvl)
    assert rc == 0 and comp == O.encode(rgb, lvl)


@pytest.mark.parametrize("shape", [(8, 8), (16, 24), (24, 40), (64, 72), (264, 520), (256, 136)])
@pytest.mark.parametrize("lvl", [0, 2])
def test_bmp_pixel_array_source_gives_the_same_bytes(shape, lvl):
    """The forward kernel reading a 24-bit BMP's pixel array (B, G, R, bottom-up, padded rows:
    KhuGleSignal.cpp:264-272) against the oracle on the planar picture; widths whose rows need padding included."""
    rgb = synth_frame(shape[0], shape[1], 5, lvl)
    rc, comp = S.encode_bgr(rgb, lvl)
    assert rc == 0 and comp == O.encode(rgb, lvl)


@pytest.mark.parametrize("shape,extra", [((8, 8), 0), ((16, 24), 6), ((264, 520), 0), ((256, 136), 2)])
def test_decode_into_a_bmp_pixel_array(shape, extra):
    """The last inverse kernel writing SaveBmp's pixel array (KhuGleSignal.cpp:285-293: B, G, R, rows bottom-up)
    against the oracle's planar picture; bytes of a row beyond the pixels stay untouched."""
    rgb = synth_frame(shape[0], shape[1], 6, 1)
    comp = O.encode(rgb, 1)
    want = O.decode(comp)
    stride = 3 * shape[1] + extra
    rc, pixels = S.decode_bgr(comp, shape[0], shape[1], stride, fill=0x5C)
    assert rc == 0
    rows = pixels.reshape(shape[0], stride)
    assert np.array_equal(rows[:, :3 * shape[1]], S.bgr_pixels(want)[0].reshape(shape[0], -1)[:, :3 * shape[1]])
    assert (rows[:, 3 * shape[1]:] == 0x5C).all()


def test_single_pass_packer_over_several_groups_of_entries():
    """A frame large enough for the packer's chain to span three groups of 64 entries (147 entries): group bit
    counts, group ends and the look-back across groups, against the oracle's bytes."""
    rgb = synth_frame(1024, 1536, 9, 0)
    rc, comp = S.encode(rgb, 1)
    assert rc == 0 and comp == O.encode(rgb, 1)


@pytest.mark.parametrize("lvl", [0, 1])
def test_single_pass_packer_on_noise(lvl):
    """White noise: nearly every symbol lies outside the packer's pair table (-8 .. 7), and an entry of 16384 symbols
    (four to six bits each) is assembled in several batches (the simulator's build holds 20000 bits per batch)."""
    rng = np.random.default_rng(77 + lvl)
    rgb = rng.integers(0, 256, (3, 256, 384), dtype=np.uint8)
    rc, comp = S.encode(rgb, lvl)
    assert rc == 0 and comp == O.encode(rgb, lvl)


def _file_from_symbols(planes, rows, cols, lvl):
    """The .comp file of given symbol planes, put together here from the oracle's table and codes (Huffman.h:57-139,
    Writer.h:9-69): header, the three streams MSB first and zero-padded to a byte, the tables."""
    import struct
    streams, tables, lens_bits, pads, ks, longest = [], [], [], [], [], 0
    for q in planes:
        q = np.ascontiguousarray(q, np.int32).ravel()
        counts, syms = O.symbol_table(q)
        codes, lens = O.build_codes(counts, syms)
        order = np.argsort(syms, kind="stable")
        at = order[np.searchsorted(syms[order], q)]
        ln = lens[at].astype(np.int64)
        cd = codes[at]
        start = np.cumsum(ln) - ln
        total = int(ln.sum())
        idx = np.repeat(np.arange(q.size), ln)
        k = np.arange(total) - np.repeat(start, ln)
        bits = ((cd[idx] >> (ln[idx] - 1 - k).astype(np.uint64)) & np.uint64(1)).astype(np.uint8)
        pad = (8 - total % 8) % 8
        streams.append(np.packbits(np.concatenate([bits, np.zeros(pad, np.uint8)])).tobytes())
        tables.append(b"".join(struct.pack("<ii", int(c), int(v)) for c, v in zip(counts, syms)))
        lens_bits.append(total + pad)
        pads.append(pad)
        ks.append(len(counts))
        longest = max(longest, int(lens.max()))
    head = struct.pack("<8i4B", *lens_bits, *ks, rows, cols, *pads, lvl)
    return head + b"".join(streams) + b"".join(tables), longest


def test_entropy_coder_on_given_symbols_matches_the_oracle_file():
    """Checks the helper above (and the simulator's symbol-plane entry) against a whole oracle file."""
    rgb = synth_frame(64, 96, 3, 1)
    q = O.symbols(rgb, 1)
    want, _ = _file_from_symbols(q, 64, 96, 1)
    assert want == O.encode(rgb, 1)
    rc, comp = S.pack_symbols(*[a.astype(np.int16) for a in q], 1)
    assert rc == 0 and comp == want


def test_single_pass_packer_with_pairs_beyond_56_bits():
    """Counts that follow the Fibonacci numbers give the deepest tree a table of 30 symbols can have: the two rarest
    symbols, -8 and 7, get 29-bit codes, so their pair has no joint entry in the packer's pair table (58 bits), other
    pairs of long codes have, threads string more than 64 bits together, and symbols outside -8 .. 7 sit beside
    them."""
    rows, cols = 1024, 2304
    n = rows * cols
    vals = [-8, 7, -7, 6, -6, 5, -5, 4, -4, 3, -3, 2, -2, 1, -1, 0] + list(range(8, 22))  # rarest first
    fib = [1, 1]
    while len(fib) < len(vals):
        fib.append(fib[-1] + fib[-2])
    assert sum(fib) <= n
    rng = np.random.default_rng(11)
    body = np.repeat(np.array(vals[2:], np.int16), fib[2:])
    body = np.concatenate([body, np.full(n - 2 - body.size, vals[-1], np.int16)])
    rng.shuffle(body)
    qy = np.concatenate([body[:16], np.array([-8, 7], np.int16), body[16:]]).reshape(rows, cols)  # an aligned pair
    qc = rng.integers(-3, 4, (2, rows // 2, cols // 2)).astype(np.int16)
    want, longest = _file_from_symbols([qy, qc[0], qc[1]], rows, cols, 1)
    assert longest == 29
    rc, comp = S.pack_symbols(qy, qc[0], qc[1], 1)
    assert rc == 0 and comp == want
    rc, out, q, _ = S.decode(comp, rows, cols)  # 29-bit codes: beyond both decode tables
    assert rc == 0 and np.array_equal(q[0], qy) and np.array_equal(out, O.decode(comp))


@pytest.mark.parametrize("name,lvl", [("cat", 1), ("couple", 0), ("library", 2), ("lena", 1), ("baboon", 0)])
def test_run_images(run_images, name, lvl):
    rgb = run_images[name]
    rc, comp = S.encode(rgb, lvl)
    assert rc == 0 and comp == O.encode(rgb, lvl)
    rc, out, q, passes = S.decode(comp, rgb.shape[1], rgb.shape[2])
    assert rc == 0
    assert np.array_equal(out, O.decode(comp))
    assert S.ssim(rgb[0], out[0]) == O.ssim(rgb[0], out[0])


@pytest.mark.parametrize("shape", SHAPES)
def test_decode(shape):
    rgb = synth_frame(shape[0], shape[1], 4, 1)
    comp = O.encode(rgb, 1)
    rc, out, q, passes = S.decode(comp, shape[0], shape[1])
    assert rc == 0
    (wq, _) = O.decode_planes(comp)
    for g, w in zip(q, wq):
        assert np.array_equal(g.astype(np.int32), w)
    assert np.array_equal(out, O.decode(comp))


def test_decode_with_the_chain_of_long_streams(monkeypatch):
    """The decoder's chain as long streams take it: one workgroup per batch of workgroup maps, two launches.  A
    1024 x 1536 frame has 54 workgroups in its luma stream, two batches of 32 in the simulator's build; noise at the
    Normal preset 150, five batches; small pictures a single, partly filled batch; a flat one an empty stream."""
    import craft
    monkeypatch.setenv("KHGSIM_CHAIN_SPLIT", "1")
    rng = np.random.default_rng(3)
    cases = [(synth_frame(1024, 1536, 9, 0), 1), (rng.integers(0, 256, (3, 1024, 1536), dtype=np.uint8), 0),
             (synth_frame(64, 96, 2, 1), 2)]
    cases += [(rgb, 0) for rgb in craft.stress_images(72, 136).values()]
    for rgb, lvl in cases:
        comp = O.encode(rgb, lvl)
        rc, out, q, _ = S.decode(comp, rgb.shape[1], rgb.shape[2])
        assert rc == 0
        (wq, _) = O.decode_planes(comp)
        for g, w in zip(q, wq):
            assert np.array_equal(g.astype(np.int32), w)
        assert np.array_equal(out, O.decode(comp))


def test_decode_reference_files():
    for ref in ("cat_high", "couple_normal", "library_extreme"):
        comp = open(os.path.join(RUN, ref + ".comp"), "rb").read()
        hd = O.peek(comp)
        rc, out, _, _ = S.decode(comp, hd.h, hd.w)
        assert rc == 0 and np.array_equal(out, O.decode(comp))


def test_decode_fixed_length_codes():
    """3-bit codes only: a decoder started at a wrong offset never resynchronises, so the
    per-offset maps (not self-synchronisation) must carry the exact starts across 6 workgroups."""
    comp = craft.fixed_length_comp(512, 512, 1)  # 786432 luma bits = 1536 subsequences
    rc, out, _, _ = S.decode(comp, 512, 512)
    assert rc == 0
    assert np.array_equal(out, O.decode(comp))


def test_decode_long_codes():
    """Code lengths up to 30 bits: first-level table, second-level table and the bitwise tail."""
    comp = craft.long_code_comp(128, 128, 1)
    rc, out, q, _ = S.decode(comp, 128, 128)
    assert rc == 0
    (wq, _) = O.decode_planes(comp)
    assert np.array_equal(q[0].astype(np.int32), wq[0])
    assert np.array_equal(out, O.decode(comp))


def test_decode_codes_beyond_32_bits():
    """A plane whose longest code exceeds 32 bits is mapped by the kernel's wide instantiation (56 start offsets),
    the other planes of the same file by the compact one."""
    comp = craft.long_code_comp(128, 128, 1, depth=44)
    rc, out, q, _ = S.decode(comp, 128, 128)
    assert rc == 0
    (wq, _) = O.decode_planes(comp)
    assert np.array_equal(q[0].astype(np.int32), wq[0])
    assert np.array_equal(out, O.decode(comp))


def test_grid_planner_balances_the_workgroups_of_a_launch():
    """khg_plan_groups: every group's rows are covered by even segments of 16..120 rows that differ by at most one
    row pair; all workgroups of the launch fit the chip at once unless every group is at its tallest already; and
    the groups' workgroups cost about the same ((rows + 8 warm-up rows) x weight)."""
    cases = [([(7680, 4320, 4), (3840, 2160, 9)], 1024),     # 8K level 1: luma, Cb + Cr
             ([(3840, 2160, 1), (1920, 1080, 1), (1920, 1080, 1)], 1024),  # 8K level 2
             ([(5760, 5760, 4), (2880, 2880, 9)], 1024), ([(8, 8, 4), (4, 4, 9)], 1024),
             ([(520, 264, 1)], 1024), ([(32768, 4096 + 16, 4), (16384, 2048 + 8, 9)], 1024),  # a tile of 32768^2
             ([(7680, 4320, 4), (3840, 2160, 9)], 512)]
    for groups, slots in cases:
        wgs, plan = S.plan_groups(groups, slots)
        total, costs, all_max = 0, [], True
        for (pw, span, weight), (seg_rows, nseg, nstrip_wg, wg0) in zip(groups, plan):
            assert wg0 == total
            assert nstrip_wg == -(-(-(-(pw // 2) // 60)) // 4)
            total += nstrip_wg * nseg
            assert seg_rows % 2 == 0 and seg_rows <= 120
            rows = [S.segment_rows(0, span, nseg, i) for i in range(nseg)]
            assert rows[0][0] == 0 and rows[-1][1] == span
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            heights = [r1 - r0 for r0, r1 in rows]
            assert all(h % 2 == 0 and 0 < h <= seg_rows for h in heights) and max(heights) - min(heights) <= 2
            if span > 16:
                assert max(heights) >= 16 or nseg == 1
            costs.append((max(heights) + 8) * weight)
            all_max = all_max and nseg == -(-span // 120)  # as few segments as the 120-row limit allows
        assert total == wgs
        assert wgs <= slots or all_max
        if len(costs) > 1 and not all_max and min(c for c in costs) > 24 * 9:
            assert max(costs) <= 1.25 * min(costs), (groups, plan, costs)


def test_output_capacity_is_checked():
    rgb = synth_frame(64, 64, 1, 0)
    rc, _ = S.encode(rgb, 0, cap=64)
    assert rc == -2


def test_device_synth_twin():
    for (h, w, seed, idx) in [(16, 24, 2025, 0), (136, 264, 2025, 77)]:
        assert np.array_equal(S.synth(h, w, seed, idx), synth_frame(h, w, seed, idx))


def _random_doubles(rng, n):
    """Signed doubles over 120 binades with random mantissas, plus values with few mantissa bits."""
    m = rng.integers(0, 1 << 52, n, dtype=np.uint64)
    m[: n // 4] &= ~np.uint64((1 << 40) - 1)  # short mantissas: products and residuals with long runs of zeros
    e = rng.integers(1023 - 60, 1023 + 60, n, dtype=np.uint64)
    s = rng.integers(0, 2, n, dtype=np.uint64)
    return ((s << np.uint64(63)) | (e << np.uint64(52)) | m).view(np.float64)


def test_division_by_the_scale_constant_is_exact():
    """khg_div_const must equal IEEE division for the two constants the lifting scale step divides by
    (DWT.h:45-50 forward, DWT.h:76-82 inverse)."""
    rng = np.random.default_rng(7)
    a = 1 / 1.149604398
    for c in (a, 1.149604398):
        for _ in range(8):
            x = _random_doubles(rng, 1 << 21)
            assert np.array_equal(S.div_const(x, c), x / c)
        x = np.concatenate([np.arange(-4096, 4096, dtype=np.float64), np.arange(-4096, 4096) * c,
                            np.nextafter(np.arange(1, 4096) * c, np.inf), np.nextafter(np.arange(1, 4096) * c, -np.inf)])
        assert np.array_equal(S.div_const(x, c), x / c)


def test_quantiser_rounding_shortcut():
    """round(c/div) through the reciprocal: equal to the division everywhere, including quotients on and
    next to half-integers, where the shortcut must hand over to the division."""
    rng = np.random.default_rng(8)
    d = rng.uniform(0.05, 900.0, 1 << 20)
    k = rng.integers(-8000, 8000, d.size).astype(np.float64)
    half = (k + 0.5) * d
    cases = [rng.uniform(-8000, 8000, d.size) * d, half, np.nextafter(half, np.inf), np.nextafter(half, -np.inf),
             half * (1 + 2.0 ** -44), half * (1 - 2.0 ** -44), k * d, np.zeros_like(d)]
    for c in cases:
        q = c / d
        tr = np.trunc(q)
        want = tr + np.sign(q) * (np.abs(q - tr) >= 0.5)  # std::round: halves away from zero
        got, fast = S.round_quot(c, d)
        assert np.array_equal(got, want)
    got, fast = S.round_quot(cases[0], d)
    assert fast.mean() > 0.999  # the division is the rare path
    got, fast = S.round_quot(half, d)
    assert fast.sum() == 0


@pytest.mark.parametrize("shape,lvl", [((64, 64), 1), ((72, 136), 0), ((264, 520), 2)])
def test_ycbcr_plane_entry_points(shape, lvl):
    """CompressImage / DecompressImage as the reference calls them: f64 Y, Cb, Cr planes in, .comp bytes out (and
    the DWT coefficients left behind), and back to f64 planes."""
    rgb = synth_frame(shape[0], shape[1], 5, 1)
    y, cb, cr = O.rgb_to_ycbcr(rgb)
    rc, comp, coef = S.encode_ycbcr(y, cb, cr, lvl, want_coef=True)
    assert rc == 0
    assert comp == O.encode(rgb, lvl)
    for got, want in zip(coef, O.forward_planes(rgb)):
        assert np.array_equal(got, want)
    rc, dy, dcb, dcr = S.decode_ycbcr(comp, shape[0], shape[1])
    assert rc == 0
    _, (wy, wcb, wcr) = O.decode_planes(comp)
    assert np.array_equal(dy, wy) and np.array_equal(dcb, wcb) and np.array_equal(dcr, wcr)


@pytest.mark.parametrize("lvl", [0, 2])
def test_stress_pictures(lvl):
    import craft
    for name, rgb in craft.stress_images(72, 136).items():
        rc, comp = S.encode(rgb, lvl)
        assert rc == 0, name
        assert comp == O.encode(rgb, lvl), name
        rc, out, _, _ = S.decode(comp, 72, 136)
        assert rc == 0, name
        assert np.array_equal(out, O.decode(comp)), name
