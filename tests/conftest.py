import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "image-compression_amd", "python"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
RUN = os.path.join(GOLDEN, "run")
IMAGES = ["baboon", "bird", "camera", "cat", "couple", "girl", "house", "lena", "library"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")
    config.addinivalue_line("markers", "slow: a GPU test of tens of seconds (the largest configuration at its size)")


@pytest.fixture(scope="session")
def run_images():
    from khugle_amd.bmp import read_bmp
    return {n: read_bmp(os.path.join(RUN, n + ".bmp")) for n in IMAGES}


@pytest.fixture(scope="session")
def codec():
    """One context big enough for every GPU test (session-wide: contexts own device memory)."""
    from khugle_amd import Codec
    c = Codec(2304, 4096)
    yield c
    c.close()
