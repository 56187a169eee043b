"""Frames of the benchmark's own sizes through the C ABI against the oracle: bytes, pixels, SSIM and PSNR.
Different code runs at these sizes than at the small test sizes (a grid that fills the chip in one round, tens of
thousands of chunks, bit offsets beyond 2^26, planes wider than one workgroup of strips)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(4320, 7680), (5760, 5760)])
def test_full_size_frame_encode_decode_metrics(shape):
    import torch
    import oracle_py as O
    from khugle_amd import Codec
    rows, cols = shape
    c = Codec(rows, cols)
    frame = c.synth_dev(rows, cols, 2025, 3)
    rgb = frame.cpu().numpy()
    want = O.encode(rgb, 1)                      # the oracle's O(N) table mode: same bytes as the reference's search
    out = torch.zeros(c.comp_bound(rows, cols), dtype=torch.uint8, device="cuda")
    c.encode_dev(frame, 1, out)
    n = c.encode_result()
    assert n == len(want) and out[:n].cpu().numpy().tobytes() == want
    # decode-only of the oracle-encoded file
    comp = torch.zeros((len(want) + 3) & ~3, dtype=torch.uint8, device="cuda")
    comp[:len(want)] = torch.from_numpy(np.frombuffer(want, np.uint8).copy()).cuda()
    dec = torch.zeros((3, rows, cols), dtype=torch.uint8, device="cuda")
    c.decode_dev(comp, len(want), rows, cols, dec)
    assert c.last_status() == 0
    pic = O.decode(want)
    assert np.array_equal(dec.cpu().numpy(), pic)
    assert c.ssim_dev(frame[0], dec[0]) == O.ssim(rgb[0], pic[0])
    assert c.psnr_dev(frame, dec) == O.psnr(rgb, pic)
    c.close()


@pytest.mark.parametrize("lvl", [0, 2])
def test_full_size_noise_frame(lvl):
    """White noise at the benchmark's size: every symbol table at its largest for the preset, long codes in every
    chunk (the packer's one-code-at-a-time path), the decoder's widest start-offset tables -- bytes and pixels against
    the oracle, and the indexed decoder (format 2) against the ordinary one."""
    import torch
    import oracle_py as O
    from khugle_amd import Codec
    rows, cols = 4320, 7680
    c = Codec(rows, cols)
    g = torch.Generator(device="cuda").manual_seed(1234 + lvl)
    frame = torch.randint(0, 256, (3, rows, cols), dtype=torch.uint8, device="cuda", generator=g)
    rgb = frame.cpu().numpy()
    want = O.encode(rgb, lvl)
    out = torch.zeros(c.comp_bound(rows, cols) + c.index_bound(rows, cols), dtype=torch.uint8, device="cuda")
    c.encode_dev(frame, lvl, out)
    n = c.encode_result()
    assert n == len(want) and out[:n].cpu().numpy().tobytes() == want
    dec = torch.zeros((3, rows, cols), dtype=torch.uint8, device="cuda")
    c.decode_dev(out, n, rows, cols, dec)
    assert c.last_status() == 0
    assert np.array_equal(dec.cpu().numpy(), O.decode(want))
    c.set_format(2)
    c.encode_dev(frame, lvl, out)
    n2 = c.encode_result()
    c.set_format(1)
    assert n2 > n and out[:n].cpu().numpy().tobytes() == want
    dec2 = torch.zeros_like(dec)
    c.decode_indexed_dev(out, n2, rows, cols, dec2)
    assert c.last_status() == 0 and torch.equal(dec, dec2)
    c.close()
