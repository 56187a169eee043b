"""The entry point that allocates nothing (khg_ssim_static_dev) against khg_ssim_dev: bit-identical results on two
calls in a row, the second on a smaller input (the fixed scratch array keeps the first call's values), and a clear
refusal beyond the array's size."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_scratch_entry_point_matches_and_allocates_nothing(codec):
    import torch
    import oracle_py as O
    from khugle_amd import lib
    from khugle_amd.synth import synth_frame
    L = lib.load()
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    tiny = torch.zeros((8, 8), dtype=torch.uint8, device="cuda")
    codec.ssim_dev(tiny, tiny)  # the runtime loads the library's code object (with the fixed array) at its first launch
    for i, (h, w) in enumerate(((1080, 1920), (264, 520))):
        a = synth_frame(h, w, 2025, i)
        b = synth_frame(h, w, 2025, i + 7)
        da, db = torch.from_numpy(a[0]).cuda(), torch.from_numpy(b[0]).cuda()
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.khg_ssim_static_dev(C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), w, h,
                                   C.c_void_p(out[i:].data_ptr()), st)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before  # nothing was allocated on the device
        got = float(out[i].item())
        assert got == codec.ssim_dev(da, db)   # the existing entry point, bit for bit
        assert got == O.ssim(a[0], b[0])       # and the oracle
    # beyond the fixed array: refused, not truncated
    big = torch.zeros((4328, 7680), dtype=torch.uint8, device="cuda")
    rc = L.khg_ssim_static_dev(C.c_void_p(big.data_ptr()), C.c_void_p(big.data_ptr()), 7680, 4328,
                               C.c_void_p(out.data_ptr()), C.c_void_p(0))
    assert rc == lib.ERR_DIMS


_COLD = r"""
import ctypes as C, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np, torch
import oracle_py as O
from khugle_amd import lib
from khugle_amd.synth import synth_frame
a, b = synth_frame(264, 520, 2025, 1), synth_frame(264, 520, 2025, 8)
da, db = torch.from_numpy(a[0]).cuda(), torch.from_numpy(b[0]).cuda()
out = torch.zeros(1, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
torch.cuda.synchronize()
L = lib.load()  # nothing of the library has run in this process
if sys.argv[3] == "prepare":
    assert L.khg_ssim_static_prepare() == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):  # the call itself under stream capture
        rc = L.khg_ssim_static_dev(C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), 520, 264,
                                   C.c_void_p(out.data_ptr()), C.c_void_p(s.cuda_stream))
    assert rc == 0
    g.replay()
else:
    rc = L.khg_ssim_static_dev(C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), 520, 264,
                               C.c_void_p(out.data_ptr()), C.c_void_p(s.cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
assert float(out[0].item()) == O.ssim(a[0], b[0]), (float(out[0].item()), O.ssim(a[0], b[0]))
print("cold ok")
"""


@pytest.mark.parametrize("mode", ["first_call", "prepare"])
def test_cold_first_call_in_a_fresh_process(mode):
    """khg_ssim_static_dev as the very first library call of a process (the code object and the array's address are
    not known yet), and -- after khg_ssim_static_prepare -- recorded into a HIP graph and replayed."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, "-c", _COLD, os.path.join(ROOT, "image-compression_amd", "python"),
                        os.path.join(ROOT, "oracle"), mode], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "cold ok" in p.stdout, p.stdout + p.stderr
