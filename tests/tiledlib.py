"""Test helpers for the row-tiled path: collectives between ranks that run as threads of one process."""
import threading

import torch


class ThreadCollectives:
    """all-reduce / reduce between `world` threads of this process (each thread = one rank)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.sh = shared
        self.rank = rank

    def _gather(self, t):
        self.sh.slots[self.rank] = t
        self.sh.barrier.wait()
        parts = [x.to(t.device) for x in self.sh.slots]
        self.sh.barrier.wait()
        return parts

    def all_reduce_sum(self, t):
        parts = self._gather(t.clone())
        t.copy_(torch.stack(parts).sum(0).to(t.dtype))

    def all_reduce_min(self, t):
        parts = self._gather(t.clone())
        t.copy_(torch.stack(parts).min(0).values)

    def reduce_sum(self, t, dst):
        parts = self._gather(t.clone())
        if self.rank == dst:
            t.copy_(torch.stack(parts).sum(0).to(t.dtype))

    def exchange(self, sends, recvs):
        """Point to point: between two ranks the tensors are matched in list order."""
        self.sh.slots[self.rank] = sends
        self.sh.barrier.wait()
        taken = {}
        for src, buf in recvs:
            i = taken.get(src, 0)
            mine = [t for dst, t in self.sh.slots[src] if dst == self.rank]
            buf.copy_(mine[i].to(buf.device))
            taken[src] = i + 1
        self.sh.barrier.wait()


def run_ranks(world, fn):
    """fn(rank, coll) on `world` threads; returns the list of results, re-raising the first failure."""
    shared = ThreadCollectives.Shared(world)
    out = [None] * world
    err = [None] * world

    def body(r):
        try:
            out[r] = fn(r, ThreadCollectives(shared, r))
        except BaseException as e:  # noqa: BLE001 - reported below
            err[r] = e
            shared.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out
