"""A context's jobs do not see each other: jobs of different image sizes and directions run back to back in ONE
workspace, which is filled with a poison byte between them (what another job's leftovers look like), and every
result still equals the oracle's.  The simulator runs the kernel sources on the CPU; the GPU test runs the same
sequence through the C ABI (khg_debug_poison)."""
import numpy as np
import pytest

import oracle_py as O
from khugle_amd.synth import synth_frame


def _jobs(run_images):
    """(rgb, preset) of three geometries: 512 x 512, 264 x 520 (two strips and a ragged edge), 256 x 256."""
    return [(run_images["lena"], 2), (synth_frame(264, 520, 2025, 0), 1), (run_images["cat"], 0),
            (synth_frame(264, 520, 2025, 6), 1)]


def _sequence(ctx_encode, ctx_decode, poison, jobs):
    comps = [O.encode(rgb, lvl) for rgb, lvl in jobs]
    wants = [O.decode(c) for c in comps]
    # decodes of different geometry back to back
    for (rgb, lvl), comp, want in zip(jobs, comps, wants):
        poison()
        assert np.array_equal(ctx_decode(comp, rgb.shape[1], rgb.shape[2]), want)
    # encode and decode alternating, leftovers of the previous job in place (no poison) and poisoned
    for fill in (False, True):
        for i, (rgb, lvl) in enumerate(jobs):
            if fill:
                poison()
            assert ctx_encode(rgb, lvl) == comps[i]
            j = (i + 1) % len(jobs)
            if fill:
                poison()
            assert np.array_equal(ctx_decode(comps[j], jobs[j][0].shape[1], jobs[j][0].shape[2]), wants[j])


def test_simulated_context_jobs_do_not_see_each_other(run_images):
    import simlib as S
    ctx = S.SimCtx(512, 520)

    def enc(rgb, lvl):
        rc, comp = ctx.encode(rgb, lvl)
        assert rc == 0
        return comp

    def dec(comp, rows, cols):
        rc, out = ctx.decode(comp, rows, cols)
        assert rc == 0
        return out

    _sequence(enc, dec, ctx.poison, _jobs(run_images))


@pytest.mark.gpu
def test_context_jobs_do_not_see_each_other(run_images):
    from khugle_amd import Codec
    c = Codec(512, 520)
    _sequence(lambda rgb, lvl: c.encode(rgb, lvl), lambda comp, rows, cols: c.decode(comp), c.debug_poison,
              _jobs(run_images))
    c.close()


@pytest.mark.gpu
def test_context_refuses_work_while_a_result_is_unfetched_and_orders_streams(run_images):
    import torch
    from khugle_amd import Codec, lib
    rgb = synth_frame(264, 520, 2025, 3)
    other = synth_frame(264, 520, 2025, 5)
    comp_other = O.encode(other, 1)
    c = Codec(512, 520)
    d = torch.from_numpy(rgb).cuda()
    out = torch.zeros(c.comp_bound(264, 520), dtype=torch.uint8, device="cuda")
    dec = torch.zeros((3, 264, 520), dtype=torch.uint8, device="cuda")
    dcomp = torch.zeros((len(comp_other) + 3) & ~3, dtype=torch.uint8, device="cuda")
    dcomp[:len(comp_other)] = torch.frombuffer(bytearray(comp_other), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        c.encode_dev(d, 1, out)
        with pytest.raises(lib.KhgError) as e:
            c.decode_dev(dcomp, len(comp_other), 264, 520, dec)  # the encode's result has not been fetched
        assert e.value.code == lib.ERR_BUSY
        with pytest.raises(lib.KhgError) as e:
            c.debug_poison()
        assert e.value.code == lib.ERR_BUSY
    n = c.encode_result()
    assert out[:n].cpu().numpy().tobytes() == O.encode(rgb, 1)
    for _ in range(4):  # jobs alternate between two streams: each is ordered behind the one before it
        with torch.cuda.stream(s2):
            c.decode_dev(dcomp, len(comp_other), 264, 520, dec)
        assert c.last_status() == 0
        assert np.array_equal(dec.cpu().numpy(), O.decode(comp_other))
        with torch.cuda.stream(s1):
            c.encode_dev(d, 1, out)
        assert c.encode_result() == n
    c.close()
