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
