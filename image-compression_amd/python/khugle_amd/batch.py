"""Batches of frames in host memory through the library's pipelined entry points (khg_batch_*): the host-side
counterpart of calling LoadBMPAndCompress / LoadComp (Main.cpp:180-334) in a loop."""
import ctypes as C

import numpy as np

from . import lib as _lib
from .codec import _check, peek


class Batch:
    """`depth` frames in flight on one GPU; pinned buffers (torch.Tensor.pin_memory()) are used in place."""

    def __init__(self, max_rows, max_cols, depth=4, device=0):
        self.L = _lib.load()
        self._b = C.c_void_p(0)
        _check(self.L.khg_batch_create(device, max_rows, max_cols, depth, C.byref(self._b)), "batch_create")

    def close(self):
        if self._b:
            self.L.khg_batch_destroy(self._b)
            self._b = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_ptrs(self, rgb_ptrs, rows, cols, lvl, out_ptrs, caps):
        """Raw form: lists of host addresses (planar RGB frames in, .comp buffers out).  Returns (lengths, statuses)."""
        n = len(rgb_ptrs)
        P = C.c_void_p * n
        Z = C.c_size_t * n
        lens = Z()
        st = (C.c_int * n)()
        rc = self.L.khg_batch_encode(self._b, n, P(*rgb_ptrs), rows, cols, lvl, P(*out_ptrs), Z(*caps), lens, st)
        return rc, list(lens), list(st)

    def decode_ptrs(self, comp_ptrs, lens, out_ptrs):
        n = len(comp_ptrs)
        P = C.c_void_p * n
        st = (C.c_int * n)()
        rc = self.L.khg_batch_decode(self._b, n, P(*comp_ptrs), (C.c_size_t * n)(*lens), P(*out_ptrs), st)
        return rc, list(st)

    def encode(self, frames, lvl):
        """frames: list of (3, H, W) uint8 numpy arrays of one size -> list of .comp byte strings."""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        _, h, w = frames[0].shape
        cap = int(self.L.khg_comp_bound(h, w))
        outs = [np.empty(cap, np.uint8) for _ in frames]
        rc, lens, st = self.encode_ptrs([f.ctypes.data for f in frames], h, w, lvl, [o.ctypes.data for o in outs],
                                        [cap] * len(frames))
        _check(rc, "batch_encode")
        return [o[:n].tobytes() for o, n in zip(outs, lens)]

    def decode(self, comps):
        """comps: list of .comp byte strings (sizes may differ) -> list of (3, H, W) uint8 numpy arrays."""
        bufs = [np.frombuffer(c, np.uint8) for c in comps]
        hds = [peek(c) for c in comps]
        outs = [np.empty((3, hd.h, hd.w), np.uint8) for hd in hds]
        rc, st = self.decode_ptrs([b.ctypes.data for b in bufs], [b.size for b in bufs], [o.ctypes.data for o in outs])
        _check(rc, "batch_decode")
        return outs
