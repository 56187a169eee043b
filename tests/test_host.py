"""Host-side pieces: BMP I/O, the C-ABI surface of the built library, multi-rank sharding plan."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, RUN
from khugle_amd import read_bmp, write_bmp
from khugle_amd import lib as khg_lib


def test_bmp_reader_handles_the_run_set():
    assert read_bmp(os.path.join(RUN, "lena.bmp")).shape == (3, 512, 512)      # 8-bpp palettised
    cat = read_bmp(os.path.join(RUN, "cat.bmp"))                                # 124-byte info header
    assert cat.shape == (3, 256, 256)
    lena = read_bmp(os.path.join(RUN, "lena.bmp"))
    assert np.array_equal(lena[0], lena[1]) and np.array_equal(lena[1], lena[2])  # grey palette


def test_bmp_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (3, 10, 7), dtype=np.uint8)  # 7*3 = 21 bytes per row: exercises row padding
    p = str(tmp_path / "x.bmp")
    write_bmp(p, rgb)
    assert np.array_equal(read_bmp(p), rgb)
    assert os.path.getsize(p) == 54 + 24 * 10


def test_library_exports_every_declared_symbol():
    """Every function include/khugle_hip.h declares is exported by the built library (no compute calls)."""
    if not os.path.exists(khg_lib.LIB_PATH):
        pytest.skip("libkhugle_hip.so not built yet (run __graft_entry__.build())")
    header = open(os.path.join(ROOT, "include", "khugle_hip.h")).read()
    declared = set(re.findall(r"\b(khg_[a-z0-9_]+)\s*\(", header))
    bound = {name for name, _, _ in khg_lib.SYMBOLS}
    assert declared == bound
    lib = C.CDLL(khg_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.khg_status_string.restype = C.c_char_p
    assert lib.khg_status_string(-2) == b"output buffer too small"
    lib.khg_comp_bound.restype = C.c_size_t
    assert lib.khg_comp_bound(12, 16) == 0 and lib.khg_comp_bound(16, 16) % 4 == 0
    # 2^31 pixels and more are refused before anything is allocated: raster indices and plane offsets are 32-bit
    assert lib.khg_comp_bound(32768, 32768) > 0
    assert lib.khg_comp_bound(65536, 32768) == 0 and lib.khg_comp_bound(46344, 46344) == 0
    ctx = C.c_void_p(0)
    assert lib.khg_ctx_create(0, 65536, 65536, C.byref(ctx)) == khg_lib.ERR_ARG and not ctx.value


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from khugle_amd import Codec
    with pytest.raises(RuntimeError):
        Codec(64, 64)
    if os.path.exists(khg_lib.LIB_PATH):
        lib = khg_lib.load()
        ctx = C.c_void_p(0)
        assert lib.khg_ctx_create(0, 64, 64, C.byref(ctx)) == khg_lib.ERR_HIP  # fails loudly, no fallback


def test_product_sources_do_not_touch_the_oracle():
    """The shipped library and package must not include, link or import anything under oracle/."""
    pkg = os.path.join(ROOT, "image-compression_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".h", ".hip", ".inl", ".py", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "khugle_oracle" not in text and "oracle_py" not in text and "khgo_" not in text, f
                assert "KHG_SIM 1" not in text, f


def test_cli_parses_and_fails_loudly_without_a_gpu(tmp_path):
    """The bmp <-> comp command line has no CPU path: on a machine without a HIP device it must raise."""
    import torch
    from khugle_amd import harness
    with pytest.raises(SystemExit):
        harness.main(["compress"])  # missing arguments
    if not torch.cuda.is_available():
        bmp = os.path.join(GOLDEN, "run", "cat.bmp")
        assert os.path.exists(bmp)
        with pytest.raises((khg_lib.KhgError, RuntimeError, OSError)):
            harness.main(["compress", bmp, str(tmp_path / "x.comp")])
        assert not os.path.exists(str(tmp_path / "x.comp"))


def test_dropin_driver_fails_loudly_without_a_gpu(tmp_path):
    """The C++ drop-in header has no CPU path either: its driver must report the library's error and exit 1."""
    import subprocess
    import torch
    driver = os.path.join(ROOT, "tests", "host", "dropin_main")
    if not os.path.exists(driver) or torch.cuda.is_available():
        pytest.skip("driver not built, or a GPU is present")
    raw = str(tmp_path / "z.raw")
    np.zeros(3 * 64 * 64, np.uint8).tofile(raw)
    r = subprocess.run([driver, "64", "64", "1", raw, str(tmp_path / "z.comp"), str(tmp_path / "c"), str(tmp_path / "p")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "khg_ctx_create" in r.stderr
    assert not os.path.exists(str(tmp_path / "z.comp"))
