"""The drop-in header in the mode a maintainer of the reference uses: KHUGLE_DROPIN_USE_REFERENCE_TYPES with the
reference's own Type.h in front (build container only: needs the reference checkout; the sources are included where
they lie, nothing is copied)."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/KhuGle"

TU = r'''
#include <memory>
#include <string>
#include <vector>
#include "Type.h"
#define KHUGLE_DROPIN_USE_REFERENCE_TYPES
#include "khugle_dropin.hpp"
// every entry point, on the application's own types
void use(double** y, double** cb, double** cr, const char* path) {
    std::unique_ptr<CompResult> res;
    khugle_hip::CompressImage(res, y, cb, cr, 256, 256, 1);
    khugle_hip::WriteAll(path, res.get());
    CompResult* back = khugle_hip::ReadAll(path);
    khugle_hip::DecompressImage(back, y, cb, cr);
    HeaderInfo h = back->info_;
    (void)h.y_len;
    delete back;
}
'''


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Type.h")), reason="reference checkout not present")
def test_dropin_header_compiles_against_the_reference_types(tmp_path):
    src = tmp_path / "dropin_ref.cpp"
    src.write_text(TU)
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + REF, "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "image-compression_amd", "host"), str(src)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
