"""The tiled mode's collectives on RCCL (torch.distributed backend "nccl").  The GPU boxes this suite runs on have
one GPU, and RCCL takes one rank per device, so the process group has a single rank -- which still sends every
collective of encode_tiled / decode_tiled through RCCL with the dtypes and reductions the real launch uses (int32
SUM and MIN all-reduce, reduce to a root, both file merges, the split entropy decode)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Counting:
    """DistCollectives with the world-size-1 short cut switched off, counting what went through the backend."""

    def __init__(self, tiled):
        self.c = tiled.DistCollectives()
        self.c.on = True
        self.calls = {"all_reduce_sum": 0, "all_reduce_min": 0, "reduce_sum": 0, "exchange": 0}

    def all_reduce_sum(self, t):
        self.calls["all_reduce_sum"] += 1
        assert t.is_cuda and t.dtype == torch.int32
        self.c.all_reduce_sum(t)

    def all_reduce_min(self, t):
        self.calls["all_reduce_min"] += 1
        assert t.is_cuda and t.dtype == torch.int32
        self.c.all_reduce_min(t)

    def reduce_sum(self, t, dst):
        self.calls["reduce_sum"] += 1
        assert t.is_cuda
        self.c.reduce_sum(t, dst)

    def exchange(self, sends, recvs):
        self.calls["exchange"] += 1
        self.c.exchange(sends, recvs)


def _rccl_rank(rank, port, rows, cols, lvl, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    from khugle_amd import Codec, tiled
    c = Codec(rows, cols)
    frame = c.synth_dev(rows, cols, 2025, 9)
    strip = tiled.exchange_halos(frame, 0, 1)  # a single rank has no neighbours: its own rows
    report = {}
    for merge in ("reduce", "gather"):
        coll = _Counting(tiled)
        comp = tiled.encode_tiled(tiled.HipTileBackend(c), strip, rows, cols, lvl, 0, 1, coll=coll, merge=merge)
        torch.cuda.synchronize()
        report[merge] = coll.calls
        with open(out_path + "." + merge, "wb") as f:
            f.write(comp.cpu().numpy().tobytes())
    n = comp.numel()
    full = torch.zeros((n + 11) & ~3, dtype=torch.uint8, device="cuda")
    full[:n] = comp
    coll = _Counting(tiled)
    tile = tiled.decode_tiled(tiled.HipTileBackend(c), full, n, rows, cols, 0, 1, coll=coll, split=True)
    torch.cuda.synchronize()
    report["decode"] = coll.calls
    np.save(out_path + ".tile.npy", tile.cpu().numpy())
    # a plain barrier and the max-over-ranks of a timing, as bench.py issues them
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    np.save(out_path + ".calls.npy", np.array([report], dtype=object), allow_pickle=True)
    dist.destroy_process_group()


def test_tiled_collectives_run_on_rccl(tmp_path):
    from khugle_amd import Codec
    rows, cols, lvl = 512, 768, 1
    out = str(tmp_path / "rccl.comp")
    mp.spawn(_rccl_rank, args=(_free_port(), rows, cols, lvl, out), nprocs=1, join=True)
    codec = Codec(rows, cols)
    want = codec.encode(codec.synth_dev(rows, cols, 2025, 9).cpu().numpy(), lvl)
    assert open(out + ".reduce", "rb").read() == want
    assert open(out + ".gather", "rb").read() == want
    assert np.array_equal(np.load(out + ".tile.npy"), codec.decode(want))
    calls = np.load(out + ".calls.npy", allow_pickle=True)[0]
    for merge in ("reduce", "gather"):
        # counts and bits summed; first indices and the three agreements (forward, code lengths, pack) by MIN
        assert calls[merge]["all_reduce_sum"] >= 2 and calls[merge]["all_reduce_min"] == 4
    assert calls["reduce"]["reduce_sum"] == 1    # the file
    assert calls["gather"]["reduce_sum"] == 1    # the chunk owners
    assert calls["gather"]["exchange"] >= 1
    # the workgroup records; one agreement behind each of the decode's three groups of steps
    assert calls["decode"]["all_reduce_sum"] == 1 and calls["decode"]["all_reduce_min"] == 3
    codec.close()
