"""GPU parity of the encode path against the CPU oracle (bit-exact: bytes, symbols, f64 planes)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, IMAGES, RUN

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _dev(rgb):
    return _torch().from_numpy(np.ascontiguousarray(rgb)).cuda()


@pytest.mark.parametrize("name", IMAGES)
@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_run_images_byte_identical(codec, run_images, name, lvl):
    """All Run/*.bmp x {Normal, High, Extreme}: .comp bytes equal the oracle's and the recorded SHA-256."""
    import oracle_py as O
    rgb = run_images[name]
    comp = codec.encode(rgb, lvl)
    want = O.encode(rgb, lvl)
    assert comp == want
    sha = json.load(open(os.path.join(GOLDEN, "comp_sha256.json")))
    assert hashlib.sha256(comp).hexdigest() == sha["%s_%d" % (name, lvl)]


@pytest.mark.parametrize("ref,name,lvl", [("cat_high", "cat", 1), ("couple_normal", "couple", 0),
                                          ("library_extreme", "library", 2)])
def test_reference_comp_files(codec, run_images, ref, name, lvl):
    """The three .comp files shipped with the reference are reproduced byte for byte."""
    want = open(os.path.join(RUN, ref + ".comp"), "rb").read()
    assert codec.encode(run_images[name], lvl) == want


@pytest.mark.parametrize("shape", [(8, 8), (16, 24), (64, 64), (264, 520), (512, 256), (1024, 1024)])
@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_synthetic_stages(codec, shape, lvl):
    """Coefficient planes, symbols and bytes on seeded synthetic frames, square and rectangular."""
    import oracle_py as O
    from khugle_amd import synth_frame
    rgb = synth_frame(shape[0], shape[1], 7, lvl)
    d = _dev(rgb)
    got = [t.cpu().numpy() for t in codec.forward_planes_dev(d)]
    want = O.forward_planes(rgb)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)  # bit-exact f64 (the stated tolerance would be 1e-6 relative)
    gq = [t.cpu().numpy() for t in codec.symbols_dev(d, lvl)]
    wq = O.symbols(rgb, lvl)
    for g, w in zip(gq, wq):
        assert np.array_equal(g.astype(np.int32), w)
    assert codec.encode(rgb, lvl) == O.encode(rgb, lvl)


@pytest.mark.parametrize("shape", [(8, 8), (24, 40), (264, 520), (1024, 1016), (2160, 3840)])
def test_bmp_pixel_array_source(codec, shape):
    """khg_encode_bgr_dev on a 24-bit BMP's pixel array (KhuGleSignal.cpp:264-272 on the GPU) gives the bytes of the
    planar path and of the oracle."""
    import torch
    import oracle_py as O
    import simlib
    from khugle_amd import synth_frame
    rgb = synth_frame(shape[0], shape[1], 5, 1)
    pixels, stride = simlib.bgr_pixels(rgb)
    d_pix = torch.from_numpy(pixels).cuda()
    out = torch.zeros(codec.comp_bound(*shape), dtype=torch.uint8, device="cuda")
    codec.encode_bgr_dev(d_pix, stride, shape[0], shape[1], 1, out)
    n = codec.encode_result()
    got = out[:n].cpu().numpy().tobytes()
    assert got == codec.encode(rgb, 1)
    if shape[0] * shape[1] <= 1 << 21:
        assert got == O.encode(rgb, 1)
    # geometry the entry point refuses
    from khugle_amd.lib import ERR_DIMS, KhgError
    with pytest.raises(KhgError) as e:
        codec.encode_bgr_dev(d_pix, 3 * shape[1] - 1, shape[0], shape[1], 1, out)
    assert e.value.code == ERR_DIMS


def test_run_bmp_files_through_the_harness(codec, run_images):
    """load_bmp_and_compress sends a 24-bit file's pixel array to the GPU as it is; palettised files are expanded
    on the host.  Either way the bytes are the oracle's."""
    import oracle_py as O
    from khugle_amd import harness
    for name in ("cat", "lena"):  # 24-bit, 8-bit palettised
        rep = harness.load_bmp_and_compress(os.path.join(RUN, name + ".bmp"), 1)
        assert rep.comp == O.encode(run_images[name], 1)


def test_device_synth_matches_host(codec):
    from khugle_amd import synth_frame
    for (h, w, seed, idx) in [(64, 96, 2025, 0), (136, 264, 2025, 5), (256, 256, 1, 1023)]:
        got = codec.synth_dev(h, w, seed, idx).cpu().numpy()
        assert np.array_equal(got, synth_frame(h, w, seed, idx))


def test_encode_dev_async_and_errors(codec):
    import oracle_py as O
    from khugle_amd import synth_frame, KhgError
    torch = _torch()
    rgb = synth_frame(256, 384, 3, 1)
    d = _dev(rgb)
    out = torch.zeros(codec.comp_bound(256, 384), dtype=torch.uint8, device="cuda")
    codec.encode_dev(d, 1, out)
    n = codec.encode_result()
    assert out[:n].cpu().numpy().tobytes() == O.encode(rgb, 1)
    # too-small output buffer is reported, not overrun
    small = torch.zeros(64, dtype=torch.uint8, device="cuda")
    codec.encode_dev(d, 1, small)
    with pytest.raises(KhgError) as e:
        codec.encode_result()
    assert e.value.code == -2
    assert int(small[40:].sum()) == 0
    # bad dimensions
    with pytest.raises(KhgError):
        codec.encode(np.zeros((3, 12, 16), np.uint8), 0)
    with pytest.raises(KhgError):
        codec.encode(np.zeros((3, 16, 16), np.uint8), 3)


@pytest.mark.parametrize("kind", ["binary", "uniform"])
def test_packer_batches_on_noise(codec, kind):
    """Noise at the Normal preset: nearly every symbol lies outside the single-pass packer's pair table; black and
    white noise has more than seven bits per luma symbol, so its entries of 16384 symbols do not fit the
    workgroup's word buffer and are assembled in several batches."""
    import oracle_py as O
    rng = np.random.default_rng(5)
    if kind == "binary":
        rgb = (rng.integers(0, 2, (3, 512, 512), dtype=np.uint8) * 255).astype(np.uint8)
    else:
        rgb = rng.integers(0, 256, (3, 512, 512), dtype=np.uint8)
    want = O.encode(rgb, 0)
    if kind == "binary":
        assert O.peek(want).y_len > 7 * 512 * 512
    comp = codec.encode(rgb, 0)
    assert comp == want
    assert np.array_equal(codec.decode(comp), O.decode(want))
