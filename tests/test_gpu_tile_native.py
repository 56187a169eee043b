"""The library's own host for the tiled mode (khg_tile_encode / khg_tile_decode on a khg_comm), through the C ABI:
  * ranks as threads of this process on the one GPU (the local communicator): halo exchange, both merges to
    different roots, the split decode -- the bytes and the pixels of one encoder;
  * the same calls on an RCCL communicator (one rank: RCCL takes one rank per device and the box has one GPU),
    which sends every collective through RCCL from C++;
  * a failure that only one rank has is returned by all of them."""
import os
import socket
import sys
import threading

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))

pytestmark = pytest.mark.gpu


def _threads(world, fn):
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa: BLE001 - re-raised below
            err[r] = e

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("shape,world,lvl", [((64, 64), 2, 1), ((256, 264), 8, 0), ((512, 512), 4, 2),
                                             ((1080, 1920), 3, 1), ((2304, 4096), 8, 1), ((192, 64), 1, 1)])
def test_thread_ranks_give_the_bytes_and_pixels_of_one_encoder(codec, shape, world, lvl):
    from khugle_amd import Codec, tiled
    rows, cols = shape
    frame = codec.synth_dev(rows, cols, 2025, 5)
    want = codec.encode(frame.cpu().numpy(), lvl)
    full = torch.from_numpy(codec.decode(want)).cuda()
    ctxs = [Codec(rows, cols) for _ in range(world)]
    group = tiled.Comm.local_group(world)
    comms = [tiled.Comm.local(group, r, world) for r in range(world)]
    halo_ok = rows // world >= tiled.HALO

    def enc(rank, dst, merge):
        r0, r1 = tiled.tile_rows(rows, rank, world)
        if halo_ok:
            strip = tiled.exchange_halos_native(ctxs[rank], comms[rank], frame[:, r0:r1].contiguous(), rows)
        else:  # tiles lower than the halo: the strip is cut from the frame
            i0, i1 = tiled.input_rows(rows, rank, world)
            strip = frame[:, i0:i1].contiguous()
        comp = tiled.encode_tiled_native(ctxs[rank], comms[rank], strip, rows, cols, lvl, dst=dst, merge=merge)
        return None if comp is None else comp.cpu().numpy().tobytes()

    for dst, merge in ((0, "reduce"), (world - 1, "gather"), (0, "gather")):
        for c in ctxs:
            c.debug_poison(0x5A)
        res = _threads(world, lambda r: enc(r, dst, merge))
        assert res[dst] == want, (dst, merge)
        assert all(res[r] is None for r in range(world) if r != dst)

    comp_t = torch.from_numpy(np.frombuffer(want + b"\0" * 8, np.uint8).copy()).cuda()
    for c in ctxs:  # a symbol that is not exchanged shows in the picture
        c.debug_poison(0x7B)
    tiles = _threads(world, lambda r: tiled.decode_tiled_native(ctxs[r], comms[r], comp_t, len(want), rows, cols))
    for rank, tile in enumerate(tiles):
        r0, r1 = tiled.tile_rows(rows, rank, world)
        assert torch.equal(tile, full[:, r0:r1]), rank
    for cm in comms:
        cm.close()
    for c in ctxs:
        c.close()


def test_a_failure_on_one_rank_is_returned_by_all(codec):
    """Rank 1's context is too small for the image, rank 0's output buffer too small for the file, a damaged file
    shows on the rank that decodes the damaged part: every rank returns, none is left in a collective."""
    from khugle_amd import Codec, tiled
    from khugle_amd.lib import KhgError, ERR_CAP, ERR_DIMS
    import ctypes as C
    rows, cols, world, lvl = 256, 256, 2, 0
    frame = codec.synth_dev(rows, cols, 7, 1)
    want = codec.encode(frame.cpu().numpy(), lvl)
    group = tiled.Comm.local_group(world)
    comms = [tiled.Comm.local(group, r, world) for r in range(world)]
    ctxs = [Codec(rows, cols), Codec(128, 128)]

    def enc(rank, cx, cap_of):
        i0, i1 = tiled.input_rows(rows, rank, world)
        strip = frame[:, i0:i1].contiguous()
        out = torch.empty(int(cx[rank].L.khg_comp_bound(rows, cols)), dtype=torch.uint8, device="cuda")
        n = C.c_size_t(0)
        return cx[rank].L.khg_tile_encode(cx[rank]._ctx, comms[rank]._h, strip[0].data_ptr(), strip[1].data_ptr(),
                                          strip[2].data_ptr(), rows, cols, lvl, out.data_ptr(), cap_of(rank, out.numel()),
                                          C.byref(n), 0, 0, None)

    assert _threads(world, lambda r: enc(r, ctxs, lambda rank, cap: cap)) == [ERR_DIMS, ERR_DIMS]
    good = [ctxs[0], Codec(rows, cols)]
    assert _threads(world, lambda r: enc(r, good, lambda rank, cap: 64 if rank == 0 else cap)) == [ERR_CAP, ERR_CAP]
    assert _threads(world, lambda r: enc(r, good, lambda rank, cap: cap)) == [0, 0]  # and the contexts still work

    bad = bytearray(want)
    for i in range(len(bad) * 3 // 4, len(bad) * 3 // 4 + 64):  # in the streams' later part (the header's lengths stay)
        bad[i] ^= 0xFF
    comp_t = torch.from_numpy(np.frombuffer(bytes(bad) + b"\0" * 8, np.uint8).copy()).cuda()

    def dec(rank):
        try:
            tiled.decode_tiled_native(good[rank], comms[rank], comp_t, len(want), rows, cols)
            return 0
        except KhgError as e:
            return e.code
    res = _threads(world, dec)
    assert res[0] == res[1]  # damaged or, by luck, decodable: the ranks agree
    for cm in comms:
        cm.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_rank(rank, rows, cols, lvl, out_path):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    from khugle_amd import Codec, tiled
    c = Codec(rows, cols)
    comm = tiled.Comm.rccl(0, 1, 0)
    frame = c.synth_dev(rows, cols, 2025, 9)
    strip = tiled.exchange_halos_native(c, comm, frame, rows)
    assert torch.equal(strip, frame)
    for merge in ("reduce", "gather"):
        comp = tiled.encode_tiled_native(c, comm, strip, rows, cols, lvl, merge=merge)
        with open(out_path + "." + merge, "wb") as f:
            f.write(comp.cpu().numpy().tobytes())
    n = comp.numel()
    full = torch.zeros((n + 11) & ~3, dtype=torch.uint8, device="cuda")
    full[:n] = comp
    c.debug_poison(0x7B)
    tile = tiled.decode_tiled_native(c, comm, full, n, rows, cols)
    np.save(out_path + ".tile.npy", tile.cpu().numpy())
    comm.close()


def test_the_native_host_runs_on_rccl(tmp_path):
    from khugle_amd import Codec
    rows, cols, lvl = 512, 768, 1
    out = str(tmp_path / "native.comp")
    mp.spawn(_rccl_rank, args=(rows, cols, lvl, out), nprocs=1, join=True)
    codec = Codec(rows, cols)
    want = codec.encode(codec.synth_dev(rows, cols, 2025, 9).cpu().numpy(), lvl)
    assert open(out + ".reduce", "rb").read() == want
    assert open(out + ".gather", "rb").read() == want
    assert np.array_equal(np.load(out + ".tile.npy"), codec.decode(want))
    codec.close()


def _eight_tiles(side, lvl=1, seed=2025):
    """One side x side frame of the benchmark's generator in eight tiles (eight contexts on this GPU, ranks as
    threads of the library's host): every rank makes its own rows where it uses them, fetches its halos from its
    neighbours, the file is merged with every byte moving once, and the entropy decode is split over the ranks.
    Returns the file (device tensor), the tiles, and the single-tile encoder's file and picture."""
    from khugle_amd import Codec, tiled
    world = 8
    ctxs = [Codec(side, side) for _ in range(world)]
    group = tiled.Comm.local_group(world)
    comms = [tiled.Comm.local(group, r, world) for r in range(world)]
    box = {}

    def rank_fn(r):
        r0, r1 = tiled.tile_rows(side, r, world)
        own = ctxs[r].synth_rows_dev(r0, r1 - r0, side, seed, 0)
        strip = tiled.exchange_halos_native(ctxs[r], comms[r], own, side)
        del own
        comp = tiled.encode_tiled_native(ctxs[r], comms[r], strip, side, side, lvl, dst=world - 1, merge="gather")
        del strip
        if comp is not None:
            n = comp.numel()
            full = torch.zeros((n + 11) & ~3, dtype=torch.uint8, device="cuda")
            full[:n] = comp
            box["full"], box["n"] = full, n
        del comp
        # the file goes back to every rank (here: one GPU, one tensor); the encode's last barrier is behind us all
        # only when the decode's first collective completes, so wait for the keeper's tensor
        while "n" not in box:
            threading.Event().wait(0.001)
        return tiled.decode_tiled_native(ctxs[r], comms[r], box["full"], box["n"], side, side)

    tiles = _threads(world, rank_fn)
    for cm in comms:
        cm.close()
    for c in ctxs[1:]:
        c.close()
    one = ctxs[0]  # its workspace serves the single-tile job
    frame = one.synth_dev(side, side, seed, 0)
    out = torch.zeros(one.comp_bound(side, side), dtype=torch.uint8, device="cuda")
    one.encode_dev(frame, lvl, out)
    n1 = one.encode_result()
    dec = torch.empty((3, side, side), dtype=torch.uint8, device="cuda")
    one.decode_dev(out, n1, side, side, dec)
    assert one.last_status() == 0
    one.close()
    return box["full"][: box["n"]], tiles, out[:n1], dec, frame


@pytest.mark.slow
def test_a_32768_square_image_in_eight_tiles_is_identical_to_single_tile():
    """The largest configuration (one 32768 x 32768 image over eight ranks) at its size."""
    from khugle_amd import tiled
    side = 32768
    comp, tiles, want, dec, _ = _eight_tiles(side)
    assert comp.numel() == want.numel() and torch.equal(comp, want)
    for r, tile in enumerate(tiles):
        r0, r1 = tiled.tile_rows(side, r, 8)
        assert torch.equal(tile, dec[:, r0:r1]), r


def test_the_4096_square_sibling_matches_the_oracle():
    """The same generator and the same eight-tile schedule at a size the oracle finishes in seconds."""
    import oracle_py as O
    from khugle_amd import tiled
    side = 4096
    comp, tiles, want, dec, frame = _eight_tiles(side)
    ref = O.encode(frame.cpu().numpy(), 1)
    assert comp.cpu().numpy().tobytes() == ref and want.cpu().numpy().tobytes() == ref
    pic = O.decode(ref)
    got = torch.cat(tiles, dim=1).cpu().numpy()
    assert np.array_equal(got, pic) and np.array_equal(dec.cpu().numpy(), pic)
