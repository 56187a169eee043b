"""khugle_amd -- Python harness for the MI355X implementation of the KhuGle codec path.

All codec work happens in image-compression_amd/csrc/libkhugle_hip.so (HIP, gfx950) behind the
C ABI of include/khugle_hip.h; this package only moves buffers and mirrors the reference's
calling sequence.  Importing the package does not load the library; using it does, and fails
loudly when the library or a HIP device is missing.
"""
from .bmp import bmp24_pixels, read_bmp, write_bmp  # noqa: F401
from .synth import synth_frame  # noqa: F401
from .lib import KhgError, LIB_PATH, load  # noqa: F401
from .codec import Codec, peek  # noqa: F401
from .harness import load_bmp_and_compress, load_comp, save_comp  # noqa: F401
