"""Format 2: the reference's file followed by a seek index (csrc/khg_index.h).  It has no counterpart in the reference;
what is checked is that the format-1 bytes in front are untouched (the reference's reader stops there), that the index
is what its definition says (oracle_py.index_trailer derives it from a cumulative sum of code lengths), and that the
indexed decoder produces the pixels of the ordinary one."""
import struct

import numpy as np
import pytest

import oracle_py as O
import simlib as S
from khugle_amd import synth_frame

SHAPES = [(8, 8), (16, 24), (64, 72), (264, 520), (256, 136), (1024, 512)]


def stress_frames():
    from craft import stress_images
    return stress_images(72, 136)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_index_bytes_and_indexed_decode(shape, lvl):
    rgb = synth_frame(shape[0], shape[1], 9, lvl)
    v1 = O.encode(rgb, lvl)
    rc, v2 = S.encode_indexed(rgb, lvl)
    assert rc == 0
    assert v2[:len(v1)] == v1                      # what ReadAll reads is unchanged
    assert v2[len(v1):] == O.index_trailer(v1)     # the index is what its definition says
    assert O.decode(v2[:len(v1)]).shape == rgb.shape
    rc, out = S.decode_indexed(v2, shape[0], shape[1])
    assert rc == 0 and np.array_equal(out, O.decode(v1))
    rc, out, _, _ = S.decode(v2, shape[0], shape[1])  # the ordinary decoder ignores the index
    assert rc == 0 and np.array_equal(out, O.decode(v1))


def test_pictures_at_the_corners_of_the_format():
    """Flat colour (empty streams, no entries), white noise (long codes), and the other stress pictures."""
    for name, rgb in stress_frames().items():
        for lvl in (0, 2):
            v1 = O.encode(rgb, lvl)
            rc, v2 = S.encode_indexed(rgb, lvl)
            assert rc == 0 and v2[:len(v1)] == v1 and v2[len(v1):] == O.index_trailer(v1), name
            rc, out = S.decode_indexed(v2, rgb.shape[1], rgb.shape[2])
            assert rc == 0 and np.array_equal(out, O.decode(v1)), name


def test_files_without_a_sound_index_are_refused():
    rgb = synth_frame(64, 72, 3, 1)
    v1 = O.encode(rgb, 1)
    rc, v2 = S.encode_indexed(rgb, 1)
    assert rc == 0
    assert S.decode_indexed(v1, 64, 72)[0] == -3                 # no index at all
    begin = (len(v1) + 7) // 8 * 8
    for pos, val in [(begin, 0x00), (begin + 4, 0x09), (begin + 8, 0x80), (begin + 12, 0x48), (begin + 24, 0x01),
                     (begin + 60, 0x7F), (len(v2) - 1, 0x00), (len(v2) - 8, 0x01)]:
        bad = bytearray(v2)
        bad[pos] ^= val or 0xFF
        assert S.decode_indexed(bytes(bad), 64, 72)[0] == -3, pos
    assert S.decode_indexed(v2[:-4], 64, 72)[0] == -3             # cut short


def test_an_index_that_lies_cannot_reach_outside_the_buffers():
    """Random damage to the entries of a sound index: wrong pixels at worst (run under AddressSanitizer by
    tools/asan_sim.sh, with the workspace sized exactly)."""
    rgb = synth_frame(136, 264, 4, 1)
    v1 = O.encode(rgb, 1)
    rc, v2 = S.encode_indexed(rgb, 1)
    assert rc == 0
    begin = (len(v1) + 7) // 8 * 8 + 72
    rng = np.random.default_rng(5)
    for rep in range(40):
        bad = bytearray(v2)
        for _ in range(int(rng.integers(1, 30))):
            bad[int(rng.integers(begin, len(v2) - 8))] = int(rng.integers(0, 256))
        rc, out = S.decode_indexed(bytes(bad), 136, 264)
        assert rc in (0, -3, -4)  # sound, malformed, or tables the decoder's limits refuse
    for rep in range(40):  # ... and damage to the streams and tables under a sound index
        bad = bytearray(v2)
        for _ in range(int(rng.integers(1, 30))):
            bad[int(rng.integers(36, len(v1)))] = int(rng.integers(0, 256))
        rc, out = S.decode_indexed(bytes(bad), 136, 264)
        assert rc in (0, -3, -4)  # sound, malformed, or tables the decoder's limits refuse


def test_forged_counts_in_a_stream_of_several_workgroups_stay_inside_the_staged_span():
    """The index's first[] entries become the per-subsequence codeword counts of the write kernel.  Entries that
    are not monotone, or inflated, must neither send a thread's cursor beyond the 256 subsequences its workgroup
    staged nor make it decode more than its own 256 bits (run under AddressSanitizer by tools/asan_sim.sh: the
    simulator's staging array has no slack).  Noise at the Normal preset: streams of several workgroup spans."""
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (3, 256, 256), dtype=np.uint8)
    v1 = O.encode(rgb, 0)
    rc, v2 = S.encode_indexed(rgb, 0)
    assert rc == 0 and v2[:len(v1)] == v1
    hdr = np.frombuffer(v1[:12], dtype="<i4")
    assert hdr[0] > 3 * 65536  # the luma stream spans more than three workgroups
    start = (len(v1) + 7) // 8 * 8
    head = np.frombuffer(v2[start:start + 72], dtype="<u4")
    ne = int(head[15] + head[16] + head[17])  # entries of the three planes: a raster index (u32) and an offset (u8) each
    first, offs = start + 72, start + 72 + 4 * ne
    for rep in range(30):
        bad = bytearray(v2)
        kind = rep % 3
        for _ in range(int(rng.integers(1, 40))):
            e = int(rng.integers(0, ne))
            if kind == 0:    # a raster index far too large, or one that steps backwards
                bad[first + 4 * e:first + 4 * e + 4] = int(rng.integers(0, 1 << 32)).to_bytes(4, "little")
            elif kind == 1:  # a stretch of entries that claim symbol 0: counts of zero, then a jump
                for k in range(e, min(ne, e + 8)):
                    bad[first + 4 * k:first + 4 * k + 4] = (0).to_bytes(4, "little")
            else:            # an offset that points anywhere into the subsequence
                bad[offs + e] = int(rng.integers(0, 256))
        rc, out = S.decode_indexed(bytes(bad), 256, 256)
        assert rc in (0, -3, -4)


# ------------------------------------------------------------------ on the GPU, through the C ABI
def _gpu_encode_indexed(codec, rgb, lvl):
    import torch
    _, h, w = rgb.shape
    d = torch.from_numpy(rgb).cuda()
    out = torch.zeros(codec.comp_bound(h, w) + codec.index_bound(h, w), dtype=torch.uint8, device="cuda")
    codec.set_format(2)
    try:
        codec.encode_dev(d, lvl, out)
        n = codec.encode_result()
    finally:
        codec.set_format(1)
    return out, n


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES + [(2160, 3840)])
def test_gpu_index(codec, shape):
    import torch
    h, w = shape
    rgb = synth_frame(h, w, 9, 1)
    v1 = codec.encode(rgb, 1)
    out, n = _gpu_encode_indexed(codec, rgb, 1)
    v2 = out[:n].cpu().numpy().tobytes()
    assert v2[:len(v1)] == v1
    if h * w <= 1 << 20:
        assert v1 == O.encode(rgb, 1) and v2[len(v1):] == O.index_trailer(v1)
    else:
        assert v2 == S.encode_indexed(rgb, 1)[1]
    codec.debug_poison(0xA5)
    dec = torch.zeros((3, h, w), dtype=torch.uint8, device="cuda")
    codec.decode_indexed_dev(out, n, h, w, dec)
    assert codec.last_status() == 0
    assert np.array_equal(dec.cpu().numpy(), codec.decode(v1))
    assert np.array_equal(codec.decode(v2), codec.decode(v1))  # the ordinary decoder ignores the index
    # a file without an index, a damaged one, and a buffer too small for the index
    from khugle_amd.lib import ERR_CAP, ERR_FORMAT
    pad = np.zeros((len(v1) + 3) // 4 * 4, np.uint8)
    pad[:len(v1)] = np.frombuffer(v1, np.uint8)
    codec.decode_indexed_dev(torch.from_numpy(pad).cuda(), len(v1), h, w, dec)
    assert codec.last_status() == ERR_FORMAT
    bad = out.clone()
    bad[(len(v1) + 7) // 8 * 8 + 4] ^= 0x7F
    codec.decode_indexed_dev(bad, n, h, w, dec)
    assert codec.last_status() == ERR_FORMAT
    small = torch.zeros((len(v1) + 3) // 4 * 4 + 64, dtype=torch.uint8, device="cuda")
    codec.set_format(2)
    try:
        codec.encode_dev(torch.from_numpy(rgb).cuda(), 1, small)
        with pytest.raises(Exception) as e:
            codec.encode_result()
        assert e.value.code == ERR_CAP
    finally:
        codec.set_format(1)


@pytest.mark.gpu
def test_gpu_index_through_the_host_entry_points(codec):
    """khg_encode / khg_encode_ycbcr / khg_encode_bgr_dev honour the context's format too; the row-tile steps do not."""
    import torch
    rgb = synth_frame(264, 520, 2, 1)
    v1 = codec.encode(rgb, 1)
    want = v1 + O.index_trailer(v1)
    codec.set_format(2)
    try:
        assert codec.encode(rgb, 1) == want
        y, cb, cr = O.rgb_to_ycbcr(rgb)
        assert codec.encode_ycbcr(y, cb, cr, 1) == want
        pixels, stride = S.bgr_pixels(rgb)
        out = torch.zeros(codec.comp_bound(264, 520) + codec.index_bound(264, 520), dtype=torch.uint8, device="cuda")
        codec.encode_bgr_dev(torch.from_numpy(pixels).cuda(), stride, 264, 520, 1, out)
        assert out[:codec.encode_result()].cpu().numpy().tobytes() == want
    finally:
        codec.set_format(1)
    assert codec.encode(rgb, 1) == v1


@pytest.mark.gpu
def test_gpu_index_full_frame_and_stress():
    """An 8K frame: same pixels with and without the index; the stress pictures through both decoders."""
    import torch
    from khugle_amd import Codec
    codec = Codec(4320, 7680)
    rgb = codec.synth_dev(4320, 7680, 2025, 3).cpu().numpy()
    v1 = codec.encode(rgb, 1)
    out, n = _gpu_encode_indexed(codec, rgb, 1)
    assert out[:len(v1)].cpu().numpy().tobytes() == v1
    dec = torch.zeros((3, 4320, 7680), dtype=torch.uint8, device="cuda")
    codec.decode_indexed_dev(out, n, 4320, 7680, dec)
    assert codec.last_status() == 0
    want = torch.zeros_like(dec)
    codec.decode_dev(out, len(v1), 4320, 7680, want)
    assert codec.last_status() == 0
    assert torch.equal(dec, want)
    for name, pic in stress_frames().items():
        v1 = codec.encode(pic, 2)
        out, n = _gpu_encode_indexed(codec, pic, 2)
        v2 = out[:n].cpu().numpy().tobytes()
        assert v2[:len(v1)] == v1 and v2[len(v1):] == O.index_trailer(v1), name
        dec = torch.zeros((3,) + pic.shape[1:], dtype=torch.uint8, device="cuda")
        codec.decode_indexed_dev(out, n, pic.shape[1], pic.shape[2], dec)
        assert codec.last_status() == 0 and np.array_equal(dec.cpu().numpy(), O.decode(v1)), name
    codec.close()
