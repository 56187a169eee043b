"""The N > 1 path of bench.py on CPU: two processes over gloo shard a batch of frames with no data-path
collective, each encodes its frames (kernel simulator standing in for the GPU), and the timings are reduced
with max-over-ranks exactly as bench.py does."""
import hashlib
import json
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FRAMES_PER_RANK = 3
SIDE = 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from khugle_amd.shard import frame_range, max_over_ranks, whole_job_rate
    from khugle_amd.synth import synth_frame
    import simlib as S
    mine = list(frame_range(rank, world, FRAMES_PER_RANK))
    digests = {}
    for g in mine:
        rc, comp = S.encode(synth_frame(SIDE, SIDE, 2025, g), 1)
        assert rc == 0
        digests[g] = hashlib.sha256(comp).hexdigest()
    dist.barrier()
    # rank r pretends to have taken (r + 1) seconds: every rank must see the slowest rank's time
    elapsed, = max_over_ranks([float(rank + 1)], torch.device("cpu"))
    rate = whole_job_rate(world, FRAMES_PER_RANK, 1, SIDE * SIDE, elapsed)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"digests": sorted(digests.items()), "elapsed": elapsed, "rate": rate}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_the_batch(tmp_path):
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import oracle_py as O
    from khugle_amd.synth import synth_frame
    seen = {}
    for r in range(world):
        with open(os.path.join(str(tmp_path), "rank%d.json" % r)) as f:
            rec = json.load(f)
        items, elapsed, rate = rec["digests"], rec["elapsed"], rec["rate"]
        assert elapsed == float(world)  # max over ranks
        assert rate == pytest.approx(world * FRAMES_PER_RANK * SIDE * SIDE / 1e6 / world)
        for g, d in items:
            assert g not in seen  # no frame is encoded twice
            seen[g] = d
    assert sorted(seen) == list(range(world * FRAMES_PER_RANK))  # and none is skipped
    for g, d in seen.items():
        want = O.encode(synth_frame(SIDE, SIDE, 2025, g), 1)
        assert hashlib.sha256(want).hexdigest() == d


def test_frame_range_partitions():
    from khugle_amd.shard import frame_range
    for world in (1, 2, 4, 8):
        got = [g for r in range(world) for g in frame_range(r, world, 128)]
        assert got == list(range(world * 128))
    with pytest.raises(ValueError):
        frame_range(2, 2, 1)
