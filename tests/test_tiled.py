"""One image tiled by rows over several ranks: tile invariance (n tiles give the bytes and pixels of one) on the
kernel simulator, with ranks as threads and as gloo processes; the GPU versions are in test_gpu_tiled.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _strip(rgb, rows, rank, world):
    from khugle_amd import tiled
    i0, i1 = tiled.input_rows(rows, rank, world)
    return torch.from_numpy(np.ascontiguousarray(rgb[:, i0:i1]))


@pytest.mark.parametrize("shape,world,lvl", [((64, 64), 2, 1), ((128, 72), 2, 0), ((192, 64), 4, 2), ((256, 264), 8, 1)])
def test_tiles_give_the_bytes_of_one_encoder(shape, world, lvl):
    from khugle_amd import tiled
    from khugle_amd.synth import synth_frame
    import oracle_py as O
    import simlib as S
    from tiledlib import run_ranks
    rows, cols = shape
    rgb = synth_frame(rows, cols, 7, 3)
    want = O.encode(rgb, lvl)

    def rank_fn(rank, coll):
        be = S.SimTileBackend(rows, cols)
        comp = tiled.encode_tiled(be, _strip(rgb, rows, rank, world), rows, cols, lvl, rank, world, coll=coll)
        return None if comp is None else comp.numpy().tobytes()

    res = run_ranks(world, rank_fn)
    assert res[0] == want
    assert all(r is None for r in res[1:])
    # decode: every rank reconstructs its rows from the whole file, decoding all of the stream itself ...
    full = O.decode(want)
    comp_t = torch.from_numpy(np.frombuffer(want + b"\0" * 8, np.uint8).copy())
    for rank in range(world):
        be = S.SimTileBackend(rows, cols)
        tile = tiled.decode_tiled(be, comp_t, len(want), rows, cols, rank, world, split=False).numpy()
        r0, r1 = tiled.tile_rows(rows, rank, world)
        assert np.array_equal(tile, full[:, r0:r1])

    # ... or 1/world of every plane's stream, with the workgroup records summed over the ranks and the decoded
    # symbols sent to the ranks whose rows need them (the simulator fills the symbol planes with junk first)
    def dec_fn(rank, coll):
        be = S.SimTileBackend(rows, cols)
        return tiled.decode_tiled(be, comp_t, len(want), rows, cols, rank, world, coll=coll).numpy()

    for rank, tile in enumerate(run_ranks(world, dec_fn)):
        r0, r1 = tiled.tile_rows(rows, rank, world)
        assert np.array_equal(tile, full[:, r0:r1]), rank


@pytest.mark.parametrize("merge,dst", [("reduce", 0), ("gather", 1), ("gather", 3)])
def test_file_merge_by_reduce_and_by_gather(merge, dst):
    """The ranks' shares become the file either by one reduce of the whole zero-padded shares or by sending each
    rank's words once to the rank that keeps the file (any rank; the header and tables come from rank 0).  Half rows
    of 36 and 68 symbols: most chunks are shorter than a word, so stretches of one rank meet stretches of others,
    and of itself, inside words."""
    from khugle_amd import tiled
    from khugle_amd.synth import synth_frame
    import oracle_py as O
    import simlib as S
    from tiledlib import run_ranks
    rows, cols, world, lvl = 128, 72, 4, 1
    rgb = synth_frame(rows, cols, 5, 2)
    want = O.encode(rgb, lvl)

    def rank_fn(rank, coll):
        be = S.SimTileBackend(rows, cols)
        comp = tiled.encode_tiled(be, _strip(rgb, rows, rank, world), rows, cols, lvl, rank, world, coll=coll, dst=dst,
                                  merge=merge)
        return None if comp is None else comp.numpy().tobytes()

    res = run_ranks(world, rank_fn)
    assert res[dst] == want
    assert all(r is None for i, r in enumerate(res) if i != dst)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_rank(rank, world, port, rows, cols, lvl, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from khugle_amd import tiled
    from khugle_amd.synth import synth_frame
    import simlib as S
    rgb = synth_frame(rows, cols, 11, 0)
    r0, r1 = tiled.tile_rows(rows, rank, world)
    own = torch.from_numpy(np.ascontiguousarray(rgb[:, r0:r1]))  # a rank holds its own rows only ...
    strip = tiled.exchange_halos(own, rank, world)              # ... and gets the halos from its neighbours
    i0, i1 = tiled.input_rows(rows, rank, world)
    assert np.array_equal(strip.numpy(), rgb[:, i0:i1])
    comp = tiled.encode_tiled(S.SimTileBackend(rows, cols), strip, rows, cols, lvl, rank, world)
    # the same file with every byte sent once, point to point, to rank 1 (the other merge of the shares)
    again = tiled.encode_tiled(S.SimTileBackend(rows, cols), strip, rows, cols, lvl, rank, world, dst=1, merge="gather")
    if rank == 0:
        assert again is None
        with open(out_path, "wb") as f:
            f.write(comp.numpy().tobytes())
    else:
        assert comp is None
        with open(out_path + ".gathered", "wb") as f:
            f.write(again.numpy().tobytes())
    # decode over the same ranks: the file goes to every rank, each entropy-decodes half of every plane's stream,
    # the workgroup records are all-reduced and the symbols travel point to point
    length = torch.tensor([comp.numel() if rank == 0 else 0], dtype=torch.int64)
    dist.broadcast(length, src=0)
    n = int(length.item())
    full = torch.zeros((n + 11) & ~3, dtype=torch.uint8)
    if rank == 0:
        full[:n] = comp
    dist.broadcast(full, src=0)
    tile = tiled.decode_tiled(S.SimTileBackend(rows, cols), full, n, rows, cols, rank, world)
    np.save(out_path + ".tile%d.npy" % rank, tile.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_exchange_halos_and_histograms(tmp_path):
    rows, cols, lvl, world = 128, 96, 1, 2
    out = str(tmp_path / "tiled.comp")
    mp.spawn(_gloo_rank, args=(world, _free_port(), rows, cols, lvl, out), nprocs=world, join=True)
    import oracle_py as O
    from khugle_amd import tiled
    from khugle_amd.synth import synth_frame
    want = O.encode(synth_frame(rows, cols, 11, 0), lvl)
    assert open(out, "rb").read() == want
    assert open(out + ".gathered", "rb").read() == want
    pixels = O.decode(want)
    for rank in range(world):
        r0, r1 = tiled.tile_rows(rows, rank, world)
        assert np.array_equal(np.load(out + ".tile%d.npy" % rank), pixels[:, r0:r1])


def _gloo_rank_damaged(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from khugle_amd import tiled
    from khugle_amd.lib import KhgError
    from khugle_amd.synth import synth_frame
    import oracle_py as O
    import simlib as S
    rows, cols = 128, 96
    comp = bytearray(O.encode(synth_frame(rows, cols, 11, 0), 1))
    # the luma table's first symbol out of the int16 range: the tree kernel flags it on every rank
    k = [int.from_bytes(comp[12 + 4 * i:16 + 4 * i], "little") for i in range(3)]
    tab = len(comp) - 8 * sum(k)
    comp[tab + 4:tab + 8] = (70000).to_bytes(4, "little")
    full = torch.from_numpy(np.frombuffer(bytes(comp) + b"\0" * 8, np.uint8).copy())
    try:
        tiled.decode_tiled(S.SimTileBackend(rows, cols), full, len(comp), rows, cols, rank, world)
        outcome = "returned"
    except KhgError as e:
        outcome = "raised %d" % e.code
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_a_damaged_file_fails_every_rank_of_a_split_decode(tmp_path):
    world = 2
    ctx = mp.spawn(_gloo_rank_damaged, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    import time
    deadline = time.time() + 120
    while not ctx.join(timeout=5):
        assert time.time() < deadline, "ranks hung"
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() == "raised -4"


def _gloo_rank_failing(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from khugle_amd import tiled
    from khugle_amd.lib import KhgError
    from khugle_amd.synth import synth_frame
    import simlib as S
    rows, cols = 128, 96

    class Failing(S.SimTileBackend):  # rank 1's tile reports "value out of range", rank 0's is fine
        def pack(self, cap):
            share, total, rc = super().pack(cap)
            return share, total, (-4 if rank == 1 else rc)

    rgb = synth_frame(rows, cols, 11, 0)
    i0, i1 = tiled.input_rows(rows, rank, world)
    strip = torch.from_numpy(np.ascontiguousarray(rgb[:, i0:i1]))
    try:
        tiled.encode_tiled(Failing(rows, cols), strip, rows, cols, 1, rank, world)
        outcome = "returned"
    except KhgError as e:
        outcome = "raised %d" % e.code
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_tile_fails_every_rank_instead_of_hanging(tmp_path):
    world = 2
    ctx = mp.spawn(_gloo_rank_failing, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    import time
    deadline = time.time() + 120
    while not ctx.join(timeout=5):  # join() returns after each process that ends
        assert time.time() < deadline, "ranks hung"
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() == "raised -4"


@pytest.mark.parametrize("step", ["forward", "lens", "chunk_off_get", "dec_map", "dec_write", "blk_set", "dec_inverse"])
def test_a_host_side_failure_on_one_rank_raises_on_all(step):
    """A step whose entry point fails on one rank only (here: a busy context, -8, on rank 1) raises on every rank at
    the next agreement instead of leaving the others in a collective."""
    from khugle_amd import tiled
    from khugle_amd.lib import KhgError
    from khugle_amd.synth import synth_frame
    import oracle_py as O
    import simlib as S
    from tiledlib import run_ranks
    rows, cols, world, lvl = 128, 96, 2, 1
    rgb = synth_frame(rows, cols, 11, 0)
    want = O.encode(rgb, lvl)
    comp_t = torch.from_numpy(np.frombuffer(want + b"\0" * 8, np.uint8).copy())

    def backend(rank):
        be = S.SimTileBackend(rows, cols)
        if rank == 1:
            inner = getattr(be, step)

            def failing(*a, **k):
                raise KhgError(-8, step)
            setattr(be, step, failing)
            assert inner is not None
        return be

    def rank_fn(rank, coll):
        be = backend(rank)
        try:
            if step in ("forward", "lens", "chunk_off_get"):
                tiled.encode_tiled(be, _strip(rgb, rows, rank, world), rows, cols, lvl, rank, world, coll=coll,
                                   merge="gather" if step == "chunk_off_get" else "reduce")
            else:
                tiled.decode_tiled(be, comp_t, len(want), rows, cols, rank, world, coll=coll)
            return 0
        except KhgError as e:
            return e.code

    assert run_ranks(world, rank_fn) == [-8, -8]


def test_tile_geometry():
    from khugle_amd import tiled
    assert tiled.tile_rows(256, 1, 4) == (64, 128)
    assert tiled.input_rows(256, 0, 4) == (0, 88)
    assert tiled.input_rows(256, 3, 4) == (168, 256)
    with pytest.raises(ValueError):
        tiled.tile_rows(100, 0, 2)
