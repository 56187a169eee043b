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


def test_strip_grid_plans_whole_batches():
    """The planner covers every row with even segments inside the limits, and counts cost in whole 16-row batches:
    no other segment height has a shorter critical path (rounds x batches), nor the same with fewer batches in all."""
    for ph in (8, 40, 264, 1080, 2160, 4320, 5760):
        for nstrip, slots, max_rows in ((1, 1024, 128), (8, 768, 128), (30, 1024, 128), (32, 1280, 512)):
            seg, nseg = S.strip_grid(ph, nstrip, slots, 8, max_rows)
            assert seg % 2 == 0 and seg * nseg >= ph and seg * (nseg - 1) < ph
            if ph >= 32:
                assert 32 <= seg <= max_rows or nseg == 1

                def cost(s_):
                    n_ = -(-ph // s_)
                    rounds, batches = -(-nstrip * n_ // slots), -(-(s_ + 8) // 16)
                    return rounds * batches, nstrip * n_ * batches
                best = cost(seg)
                for n_ in range(max(1, -(-ph // max_rows)), ph // 32 + 1):
                    s_ = (-(-ph // n_) + 1) & ~1
                    if s_ <= max_rows:
                        assert cost(s_) >= best, (ph, nstrip, slots, seg, s_)


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
