"""The CPU oracle against the reference's own golden data (no GPU needed)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, IMAGES, RUN
import oracle_py as O


@pytest.mark.parametrize("ref,name,lvl", [("cat_high", "cat", 1), ("couple_normal", "couple", 0),
                                          ("library_extreme", "library", 2)])
def test_reference_comp_files_byte_identical(run_images, ref, name, lvl):
    want = open(os.path.join(RUN, ref + ".comp"), "rb").read()
    assert O.encode(run_images[name], lvl) == want
    assert O.encode(run_images[name], lvl, faithful=True) == want


def test_comp_sha256_table(run_images):
    sha = json.load(open(os.path.join(GOLDEN, "comp_sha256.json")))
    assert len(sha) == 27
    for name in IMAGES:
        for lvl in range(3):
            assert hashlib.sha256(O.encode(run_images[name], lvl)).hexdigest() == sha["%s_%d" % (name, lvl)]


def test_published_quality_figures(run_images):
    """Pictures/{Normal,High,Extreme}.png: PSNR, SSIM, size and table sizes for cat.bmp."""
    shots = json.load(open(os.path.join(GOLDEN, "screenshots.json")))
    rgb = run_images["cat"]
    for lvl, want in shots["presets"].items():
        comp = O.encode(rgb, int(lvl))
        dec = O.decode(comp)
        hd = O.peek(comp)
        assert "%6.3f" % O.psnr(rgb, dec) == want["psnr"]
        assert "%4.3f" % O.ssim(rgb[0], dec[0]) == want["ssim"]
        assert "%6.3f" % (len(comp) / 1024) == want["size_kb"]
        assert "%.3f" % (len(comp) / shots["bmp_bytes"] * 100) == want["percent"]
        assert [hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len] == want["tables"]


def test_header_fields_of_reference_files():
    for ref, bits, ks, pads, lvl, n in [("cat_high", (151896, 22504, 22080), (332, 50, 60), (5, 5, 4), 1, 256),
                                        ("couple_normal", (220128, 33704, 39872), (559, 88, 126), (7, 7, 1), 0, 256),
                                        ("library_extreme", (360856, 72168, 70832), (51, 13, 11), (4, 6, 6), 2, 512)]:
        hd = O.peek(open(os.path.join(RUN, ref + ".comp"), "rb").read())
        assert (hd.y_len, hd.cb_len, hd.cr_len) == bits
        assert (hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len) == ks
        assert (hd.y_pad, hd.cb_pad, hd.cr_pad) == pads
        assert hd.lvl == lvl and hd.h == n and hd.w == n


def test_faithful_table_equals_dense_table():
    rng = np.random.default_rng(0)
    q = rng.integers(-40, 40, 5000).astype(np.int32)
    a = O.symbol_table(q, faithful=True)
    b = O.symbol_table(q, faithful=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[0].sum() == q.size and a[1][0] == q[0]


def test_grey_images_have_single_entry_chroma_tables(run_images):
    for name in ["bird", "camera", "girl", "house", "lena"]:
        hd = O.peek(O.encode(run_images[name], 1))
        assert (hd.rle_cb_len, hd.rle_cr_len, hd.cb_len, hd.cr_len, hd.cb_pad, hd.cr_pad) == (1, 1, 0, 0, 0, 0)


def test_dims_and_format_errors():
    with pytest.raises(O.OracleError):
        O.encode(np.zeros((3, 12, 16), np.uint8), 0)
    with pytest.raises(O.OracleError):
        O.encode(np.zeros((3, 16, 16), np.uint8), 5)
    with pytest.raises(O.OracleError):
        O.decode(b"\0" * 20)


# ---- the reference's own headers, compiled (oracle/_ref): only where it has been built

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (reference checkout absent)")


@needs_ref
def test_ref_dwt_matches_restatement():
    R = O.ref()
    rng = np.random.default_rng(1)
    for n in (8, 16, 64, 128):
        x = rng.uniform(-300, 300, (n, n))
        a = x.copy(); b = x.copy()
        R.kref_fwt2d(a, n, n)
        assert O.lib().khgo_fwt2d(b, n, n, n) == 0
        assert np.array_equal(a, b)
        R.kref_iwt2d(a, n, n)
        assert O.lib().khgo_iwt2d(b, n, n, n) == 0
        assert np.array_equal(a, b)


@needs_ref
def test_ref_table_and_codes_match_restatement():
    R = O.ref()
    rng = np.random.default_rng(2)
    for trial in range(6):
        n = 64
        q = np.rint(rng.laplace(0, 1 + 3 * trial, (n, n))).astype(np.int32)
        counts = np.zeros(q.size, np.int32); syms = np.zeros(q.size, np.int32)
        k = R.kref_rle(q, n, n, counts, syms)
        oc, osym = O.symbol_table(q)
        assert k == oc.size and np.array_equal(counts[:k], oc) and np.array_equal(syms[:k], osym)
        buf = bytes(64 * k + 64)
        import ctypes as C
        out = C.create_string_buffer(64 * k + 64)
        R.kref_codes(oc, osym, k, out, len(out))
        ref_codes = out.value.decode().split("\n")[:k]
        codes, lens = O.build_codes(oc, osym)
        mine = [format(int(c), "b").zfill(int(l)) if l else "" for c, l in zip(codes, lens)]
        assert mine == ref_codes


@needs_ref
def test_ref_hybrid_files_match_oracle(tmp_path, run_images):
    """FWT2D / RunLengthEncoding / BuildHuffmanTree / EncodeWithHuffman / WriteAll / ReadAll /
    DecodeWithHuffman / IWT2D taken from the reference's headers give the oracle's bytes and pixels."""
    import ctypes as C
    from khugle_amd import synth_frame
    R = O.ref()
    cases = [(run_images["cat"], 1), (run_images["lena"], 0), (synth_frame(64, 64, 9, 0), 2),
             (synth_frame(128, 128, 9, 1), 1)]
    for rgb, lvl in cases:
        path = str(tmp_path / "t.comp").encode()
        n = rgb.shape[1]
        assert R.kref_encode_file(rgb[0], rgb[1], rgb[2], n, lvl, path) == 0
        got = open(path, "rb").read()
        assert got == O.encode(rgb, lvl)
        out = np.zeros_like(rgb)
        nn = C.c_int(0)
        assert R.kref_decode_file(path, out[0], out[1], out[2], C.byref(nn)) == 0
        assert nn.value == n and np.array_equal(out, O.decode(got))
        assert R.kref_ssim(rgb[0], out[0], n, n) == O.ssim(rgb[0], out[0])


def test_decoded_pictures_match_the_reference_decoder(run_images):
    """Decode side: the oracle's pictures for all 27 files and the three shipped .comp files hash to what the
    reference's own reader / Huffman decoder / inverse DWT produced (tests/golden/make_decode_sha256.py)."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN, IMAGES, RUN
    golden = json.load(open(os.path.join(GOLDEN, "decode_sha256.json")))
    for name in IMAGES:
        for lvl in (0, 1, 2):
            pic = O.decode(O.encode(run_images[name], lvl))
            assert hashlib.sha256(pic.tobytes()).hexdigest() == golden["%s_%d" % (name, lvl)], (name, lvl)
    for f in ("cat_high", "couple_normal", "library_extreme"):
        pic = O.decode(open(os.path.join(RUN, f + ".comp"), "rb").read())
        assert hashlib.sha256(pic.tobytes()).hexdigest() == golden[f + ".comp"]


def test_simulated_decoder_matches_the_reference_decoder():
    import hashlib
    import json
    import os
    import simlib as S
    from conftest import GOLDEN, RUN
    golden = json.load(open(os.path.join(GOLDEN, "decode_sha256.json")))
    for f in ("cat_high", "couple_normal", "library_extreme"):
        comp = open(os.path.join(RUN, f + ".comp"), "rb").read()
        hd = O.peek(comp)
        rc, pic, _, _ = S.decode(comp, hd.h, hd.w)
        assert rc == 0 and hashlib.sha256(pic.tobytes()).hexdigest() == golden[f + ".comp"]
