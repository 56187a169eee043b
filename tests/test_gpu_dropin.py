"""The C++ drop-in header (CompressImage / DecompressImage / WriteAll / ReadAll on double** planes, as the
application calls them) driven by tests/host/dropin_main on the GPU and checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN

pytestmark = pytest.mark.gpu
DRIVER = os.path.join(ROOT, "tests", "host", "dropin_main")


def _run(rgb, lvl, tmp_path):
    _, h, w = rgb.shape
    raw = str(tmp_path / "in.raw"); comp = str(tmp_path / "out.comp")
    coef = str(tmp_path / "coef.f64"); planes = str(tmp_path / "planes.f64")
    np.ascontiguousarray(rgb, np.uint8).tofile(raw)
    r = subprocess.run([DRIVER, str(h), str(w), str(lvl), raw, comp, coef, planes], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    n = h * w

    def split(path):
        v = np.fromfile(path, np.float64)
        assert v.size == n + n // 2
        return v[:n].reshape(h, w), v[n:n + n // 4].reshape(h // 2, w // 2), v[n + n // 4:].reshape(h // 2, w // 2)

    _run.lines = r.stdout.splitlines()  # the two status lines CompressImage / DecompressImage hand back
    return open(comp, "rb").read(), split(coef), split(planes)


@pytest.mark.parametrize("name,lvl", [("cat", 1), ("couple", 0), ("library", 2)])
def test_dropin_reproduces_the_reference_files(run_images, name, lvl, tmp_path):
    import oracle_py as O
    assert os.path.exists(DRIVER), "build it with __graft_entry__.build()"
    rgb = run_images[name]
    comp, coef, planes = _run(rgb, lvl, tmp_path)
    ref_name = {0: "normal", 1: "high", 2: "extreme"}[lvl]
    assert comp == open(os.path.join(GOLDEN, "run", "%s_%s.comp" % (name, ref_name)), "rb").read()
    for got, want in zip(coef, O.forward_planes(rgb)):    # what CompressImage leaves in its input planes
        assert np.array_equal(got, want)
    for got, want in zip(planes, O.decode_planes(comp)[1]):  # DecompressImage's outputs
        assert np.array_equal(got, want)
    hd = O.peek(comp)
    assert _run.lines == list(O.vis_steps(hd.h, hd.w, lvl, (hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len))[1:])
    if name == "cat":  # the lines of Pictures/High.png
        import json
        want = json.load(open(os.path.join(GOLDEN, "screenshots.json")))["presets"][str(lvl)]
        assert _run.lines == ["Step Size (min, max) : %s, %s" % tuple(want["step"]),
                              "RLE Size (Y, Cb, Cr) : %d, %d, %d" % tuple(want["tables"])]


def test_dropin_rectangular(tmp_path):
    import oracle_py as O
    from khugle_amd.synth import synth_frame
    rgb = synth_frame(136, 264, 3, 2)
    comp, _, planes = _run(rgb, 1, tmp_path)
    assert comp == O.encode(rgb, 1)
    for got, want in zip(planes, O.decode_planes(comp)[1]):
        assert np.array_equal(got, want)
