"""The pictures and status lines the application shows beside a result (Compress.h:116-141, 245-300, 330-391;
Decompress.h:128-271).  The two status lines are pinned by the published screenshots (tests/golden/screenshots.json:
Pictures/{Normal,High,Extreme}.png); the pictures are "parity unpinned" against the reference itself (its Compress.h /
Decompress.h need the Win32 GUI classes and cannot be built here) and are checked against the oracle's restatement."""
import json
import os

import numpy as np
import pytest

import oracle_py as O
import simlib as S
from conftest import GOLDEN
from khugle_amd import synth_frame

SHAPES = [(8, 8), (16, 24), (64, 40), (256, 256), (264, 520)]


def test_oracle_status_lines_match_the_published_screenshots():
    shots = json.load(open(os.path.join(GOLDEN, "screenshots.json")))
    for lvl in (0, 1, 2):
        want = shots["presets"][str(lvl)]
        _, s1, s2 = O.vis_steps(256, 256, lvl, want["tables"])  # cat.bmp is 256 x 256
        assert s1 == "Step Size (min, max) : %s, %s" % tuple(want["step"])
        assert s2 == "RLE Size (Y, Cb, Cr) : %d, %d, %d" % tuple(want["tables"])


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_step_picture_and_its_extremes(shape, lvl):
    """The kernel's picture against the oracle's; the extremes taken at the bands' corners against the oracle's search
    of the whole plane (they are what the status line prints)."""
    rc, got, (lo, hi) = S.vis_steps(shape[0], shape[1], lvl)
    want, s1, _ = O.vis_steps(shape[0], shape[1], lvl)
    assert rc == 0 and np.array_equal(got, want)
    assert s1 == "Step Size (min, max) : %.3f, %.3f" % (lo, hi)
    dy = np.zeros(shape)
    O.lib().khgo_divisor_plane(shape[0], shape[1], lvl, 0, dy)
    assert lo == dy.min() and hi == dy.max()


@pytest.mark.parametrize("shape", SHAPES)
def test_bar_picture(shape):
    h, w = shape
    rng = np.random.default_rng(h * w)
    cases = [(0, 0, 0), (h, 2 * h, 3 * h), (h * h - 1, h * h // 2, 5), (h * w // 3, h * w // 7 + 3, h * w // 11 + 1)]
    cases += [tuple(int(v) for v in rng.integers(0, h * min(h, w), 3)) for _ in range(4)]
    for lens in cases:
        rc, got = S.vis_bars(h, w, lens)
        assert rc == 0 and np.array_equal(got, O.vis_bars(h, w, lens)), lens


@pytest.mark.parametrize("shape", SHAPES)
def test_coefficient_pictures(shape):
    rgb = synth_frame(shape[0], shape[1], 8, 2)
    rc, got = S.vis_coeffs(rgb)
    assert rc == 0 and np.array_equal(got, O.vis_coeffs(rgb))
    for lvl in (0, 1, 2):
        comp = O.encode(rgb, lvl)
        rc, got = S.vis_dequant(comp, shape[0], shape[1])
        assert rc == 0 and np.array_equal(got, O.vis_dequant(comp)), lvl


DOWN_CASES = [((256, 256), (256, 256)), ((512, 512), (256, 256)), ((264, 520), (256, 256)), ((72, 136), (256, 256)),
              ((1000, 1504), (256, 256)), ((64, 64), (7, 13)), ((8, 8), (1, 1)), ((300, 200), (299, 201))]


@pytest.mark.parametrize("src,dst", DOWN_CASES)
def test_thumbnails(src, dst):
    """DownSampleImage (DownSample.h:5-37): the kernel body against the oracle's single-precision restatement,
    including the sizes at which the reference divides by zero on the last row and column (equal sizes, enlargement)."""
    rng = np.random.default_rng(src[0] * 7 + dst[1])
    plane = rng.integers(0, 256, src, dtype=np.uint8)
    assert np.array_equal(S.downsample(plane, *dst), O.downsample(plane, *dst))
    smooth = synth_frame((src[0] + 7) // 8 * 8, (src[1] + 7) // 8 * 8, 3, 0)[0][:src[0], :src[1]]
    assert np.array_equal(S.downsample(smooth, *dst), O.downsample(smooth, *dst))


# ------------------------------------------------------------------ the same on the GPU, through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", DOWN_CASES + [((4320, 7680), (256, 256))])
def test_gpu_thumbnails(codec, src, dst):
    import torch
    rng = np.random.default_rng(src[0] * 7 + dst[1])
    plane = rng.integers(0, 256, src, dtype=np.uint8)
    got = codec.downsample_dev(torch.from_numpy(plane).cuda(), *dst).cpu().numpy()
    assert np.array_equal(got, O.downsample(plane, *dst))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES + [(2160, 3840)])
def test_gpu_pictures(codec, shape):
    import torch
    h, w = shape
    small = h * w <= 1 << 20
    rgb = synth_frame(h, w, 8, 2)
    d = torch.from_numpy(rgb).cuda()
    got = codec.vis_coeffs_dev(d).cpu().numpy()
    if small:
        assert np.array_equal(got, O.vis_coeffs(rgb))
    else:  # beyond the oracle's reach in seconds: against the library's own f64 planes through the oracle's conversion
        assert np.array_equal(got, O.ycbcr_to_rgb(*[t.cpu().numpy() for t in codec.forward_planes_dev(d)]))
    for lvl in (0, 1, 2):
        got = codec.vis_steps_dev(h, w, lvl).cpu().numpy()
        rc, want, (lo, hi) = S.vis_steps(h, w, lvl)
        assert np.array_equal(got, want)
        if small:
            assert np.array_equal(got, O.vis_steps(h, w, lvl)[0])
        s1, s2 = codec.status_strings(h, w, lvl, (5, 6, 7))
        assert s1 == "Step Size (min, max) : %.3f, %.3f" % (lo, hi) and s2 == "RLE Size (Y, Cb, Cr) : 5, 6, 7"
        if small:
            assert (s1, s2) == O.vis_steps(h, w, lvl, (5, 6, 7))[1:]
    for lens in [(0, 0, 0), (h * h - 1, h * h // 2, 5), (h * w // 3, h * w // 7 + 3, h * w // 11 + 1)]:
        got = codec.vis_bars_dev(h, w, lens).cpu().numpy()
        assert np.array_equal(got, O.vis_bars(h, w, lens)), lens
    lvl = 1
    comp = codec.encode(rgb, lvl)
    padded = np.zeros((len(comp) + 3) // 4 * 4, np.uint8)
    padded[:len(comp)] = np.frombuffer(comp, np.uint8)
    got = codec.vis_dequant_dev(torch.from_numpy(padded).cuda(), len(comp), h, w).cpu().numpy()
    if small:
        assert np.array_equal(got, O.vis_dequant(comp))
    else:
        assert np.array_equal(got, S.vis_dequant(comp, h, w)[1])


@pytest.mark.gpu
def test_gpu_harness_lines_match_the_published_screenshots(codec, tmp_path):
    """LoadBMPAndCompress / LoadComp with their by-products on cat.bmp: both status lines as in
    Pictures/{Normal,High,Extreme}.png, the pictures as the oracle draws them."""
    from khugle_amd import harness
    shots = json.load(open(os.path.join(GOLDEN, "screenshots.json")))
    bmp = os.path.join(GOLDEN, "run", "cat.bmp")
    for name, lvl in harness.PRESETS.items():
        want = shots["presets"][str(lvl)]
        rep = harness.load_bmp_and_compress(bmp, name, codec, visualize=True)
        assert rep.step_str == "Step Size (min, max) : %s, %s" % tuple(want["step"])
        assert rep.rle_str == "RLE Size (Y, Cb, Cr) : %d, %d, %d" % tuple(want["tables"])
        hd = O.peek(rep.comp)
        rgb = harness.read_bmp(bmp)
        assert np.array_equal(rep.pictures["coefficients"], O.vis_coeffs(rgb))
        assert np.array_equal(rep.pictures["steps"], O.vis_steps(256, 256, lvl)[0])
        assert np.array_equal(rep.pictures["bars"], O.vis_bars(256, 256, [hd.y_len // 8, hd.cb_len // 8, hd.cr_len // 8]))
        path = str(tmp_path / ("cat_%s.comp" % name))
        harness.save_comp(rep, path)
        _, _, extra = harness.load_comp(path, codec, visualize=True)
        assert (extra["step_str"], extra["rle_str"]) == (rep.step_str, rep.rle_str)
        assert np.array_equal(extra["pictures"]["coefficients"], O.vis_dequant(rep.comp))
        assert np.array_equal(extra["pictures"]["steps"], rep.pictures["steps"])
        lens = [(hd.y_len - hd.y_pad) // 8, (hd.cb_len - hd.cb_pad) // 8, (hd.cr_len - hd.cr_pad) // 8]
        assert np.array_equal(extra["pictures"]["bars"], O.vis_bars(256, 256, lens))
    out = str(tmp_path / "cli.comp")
    assert harness.main(["compress", bmp, out, "--pictures", str(tmp_path / "enc")]) == 0
    assert harness.main(["decompress", out, str(tmp_path / "cli.bmp"), "--pictures", str(tmp_path / "dec")]) == 0
    assert sorted(os.listdir(str(tmp_path / "dec"))) == ["bars.bmp", "coefficients.bmp", "steps.bmp"]
