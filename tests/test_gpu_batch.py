"""Host-memory batches through khg_batch_* (pipelined contexts, pinned and pageable buffers) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batch_encode_decode_pageable(run_images):
    import oracle_py as O
    from khugle_amd.batch import Batch
    from khugle_amd.synth import synth_frame
    b = Batch(512, 520, depth=3)
    frames = [synth_frame(264, 520, 2025, i) for i in range(7)]  # more frames than slots
    comps = b.encode(frames, 1)
    assert comps == [O.encode(f, 1) for f in frames]
    # files of different sizes in one decode batch
    mixed = comps[:3] + [O.encode(run_images["cat"], 0), O.encode(run_images["lena"], 2)] + comps[3:]
    outs = b.decode(mixed)
    for got, comp in zip(outs, mixed):
        assert np.array_equal(got, O.decode(comp))
    b.close()


def test_batch_pinned_buffers_and_errors():
    import oracle_py as O
    from khugle_amd.batch import Batch
    from khugle_amd.synth import synth_frame
    b = Batch(256, 256, depth=2)
    n, h, w = 5, 128, 256
    frames = torch.stack([torch.from_numpy(synth_frame(h, w, 7, i)) for i in range(n)]).pin_memory()
    cap = int(b.L.khg_comp_bound(h, w))
    outs = torch.zeros((n, cap), dtype=torch.uint8).pin_memory()
    rc, lens, st = b.encode_ptrs([frames[i].data_ptr() for i in range(n)], h, w, 2, [outs[i].data_ptr() for i in range(n)],
                                 [cap] * n)
    assert rc == 0 and st == [0] * n
    for i in range(n):
        assert outs[i, :lens[i]].numpy().tobytes() == O.encode(frames[i].numpy(), 2)
    # one output buffer too small, one corrupt file: the other frames still go through
    caps = [cap, 64, cap, cap, cap]
    rc, lens2, st = b.encode_ptrs([frames[i].data_ptr() for i in range(n)], h, w, 2, [outs[i].data_ptr() for i in range(n)], caps)
    assert rc == -2 and st == [0, -2, 0, 0, 0] and lens2[0] == lens[0] and lens2[1] == 0
    comps = [outs[i, :lens[i]].numpy().tobytes() for i in range(n)]
    bad = bytearray(comps[2]); bad[12:16] = (0).to_bytes(4, "little")  # table size 0
    bufs = [np.frombuffer(bytes(c), np.uint8) for c in (comps[0], bytes(bad), comps[4])]
    dec = torch.zeros((3, 3, h, w), dtype=torch.uint8).pin_memory()
    rc, st = b.decode_ptrs([x.ctypes.data for x in bufs], [x.size for x in bufs], [dec[i].data_ptr() for i in range(3)])
    assert st[0] == 0 and st[2] == 0 and st[1] != 0 and rc == st[1]
    assert np.array_equal(dec[0].numpy(), O.decode(comps[0])) and np.array_equal(dec[2].numpy(), O.decode(comps[4]))
    b.close()
