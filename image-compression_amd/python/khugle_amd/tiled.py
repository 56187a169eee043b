"""One image tiled by rows over several GPUs (one process per GPU): the host layer that runs the collectives
between the per-GPU steps of the C ABI (khg_tile_* in include/khugle_hip.h).

Rank t of n owns image rows [rows*t/n, rows*(t+1)/n).  Data movement between ranks:
  * exchange_halos : 24 rows to each neighbour, point to point (the DWT halos)
  * histogram      : all-reduce SUM of the symbol counts, all-reduce MIN of the first raster indices
                     -> every rank builds the same table, tree and codes
  * chunk bits     : all-reduce SUM -> every rank knows the bit offset of every chunk
  * file bytes     : reduce SUM of the ranks' disjoint shares to the rank that keeps the file
  * decode         : the entropy decode is split over the ranks (each decodes 1/n of every plane's stream):
                     all-reduce SUM of the workgroup records -> every rank chains all spans; then decoded symbols
                     go point to point from the ranks that hold them to the ranks whose rows need them
Two hosts run this schedule.  The library's own (khg_tile_encode / khg_tile_decode on a khg_comm: RCCL called from
C++, every exchange in place on the context's buffers) is what Comm / encode_tiled_native / decode_tiled_native
below call: the path for GPUs.  encode_tiled / decode_tiled run the same steps from Python over a torch.distributed
process group; with gloo and the kernel simulator behind the same backend interface they are how the CPU tests
cover the schedule, and on GPUs (backend "nccl") they serve as the cross-check of the native host.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import lib as _lib
from .codec import _check, _ptr, _stream

HALO = 24          # image rows a tile needs beyond its own on each side
NBINS = 16384
_ABSENT = 0x7FFFFFFF


def tile_rows(rows, rank, world):
    """[first, last) image rows owned by `rank`."""
    if rows % (8 * world) != 0:
        raise ValueError("rows must be a multiple of 8 * ranks")
    return rows // world * rank, rows // world * (rank + 1)


def input_rows(rows, rank, world):
    """[first, last) image rows a rank's input strip must hold: its tile plus the halo, clipped to the image."""
    r0, r1 = tile_rows(rows, rank, world)
    return max(0, r0 - HALO), min(rows, r1 + HALO)


def exchange_halos(own, rank, world, group=None):
    """own: (3, R, W) uint8 tensor with the rank's own rows.  Returns the strip with the neighbours' edge rows
    attached (top rank and bottom rank have none on the outer side)."""
    _, R, W = own.shape
    if R < HALO:
        raise ValueError("tiles must be at least %d rows high" % HALO)
    up = torch.empty((3, HALO, W), dtype=own.dtype, device=own.device) if rank > 0 else None
    down = torch.empty((3, HALO, W), dtype=own.dtype, device=own.device) if rank < world - 1 else None
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, own[:, :HALO].contiguous(), rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, up, rank - 1, group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, own[:, R - HALO:].contiguous(), rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, down, rank + 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    parts = [p for p in (up, own, down) if p is not None]
    return torch.cat(parts, dim=1).contiguous()


class HipTileBackend:
    """The per-GPU steps on a khugle_amd.Codec (device tensors, torch's current stream)."""

    def __init__(self, codec):
        self.c = codec
        self.L = codec.L
        self.dev = torch.device("cuda", codec.device)

    def comp_bound(self, rows, cols):
        return int(self.L.khg_comp_bound(rows, cols))

    def forward(self, strip, rows, cols, lvl, rank, world):
        assert strip.is_cuda and strip.dtype == torch.uint8 and strip.is_contiguous()
        i0, i1 = input_rows(rows, rank, world)
        assert tuple(strip.shape) == (3, i1 - i0, cols), "strip must hold image rows %d..%d" % (i0, i1)
        self.shape = (rows, cols, world)
        _check(self.L.khg_tile_forward_dev(self.c._ctx, _ptr(strip[0]), _ptr(strip[1]), _ptr(strip[2]), rows, cols, lvl,
                                           rank, world, _stream()), "tile_forward")

    def hist_get(self):
        cnt = torch.empty(3 * NBINS, dtype=torch.int32, device=self.dev)
        first = torch.empty(3 * NBINS, dtype=torch.int32, device=self.dev)
        _check(self.L.khg_tile_hist_get(self.c._ctx, _ptr(cnt), _ptr(first), _stream()), "tile_hist_get")
        return cnt, first

    def hist_set(self, cnt, first):
        _check(self.L.khg_tile_hist_set(self.c._ctx, _ptr(cnt), _ptr(first), _stream()), "tile_hist_set")

    def lens(self):
        _check(self.L.khg_tile_lens_dev(self.c._ctx, _stream()), "tile_lens")

    def bits_get(self):
        rows, cols, world = self.shape
        bits = torch.empty(self.L.khg_tile_chunk_count(rows, cols, world), dtype=torch.int32, device=self.dev)
        _check(self.L.khg_tile_bits_get(self.c._ctx, _ptr(bits), _stream()), "tile_bits_get")
        return bits

    def bits_set(self, bits):
        _check(self.L.khg_tile_bits_set(self.c._ctx, _ptr(bits), _stream()), "tile_bits_set")

    def chunk_off_get(self):
        rows, cols, world = self.shape
        off = torch.empty(self.L.khg_tile_chunk_count(rows, cols, world), dtype=torch.int64, device=self.dev)
        _check(self.L.khg_tile_chunk_off_get(self.c._ctx, _ptr(off), _stream()), "tile_chunk_off_get")
        return off

    def pack(self, cap):
        share = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        _check(self.L.khg_tile_pack_dev(self.c._ctx, _ptr(share), cap, _stream()), "tile_pack")
        n = C.c_size_t(0)
        status = self.L.khg_encode_result(self.c._ctx, C.byref(n))  # device-side status of this rank's tile: not raised here
        return share, n.value, status

    def decode(self, comp, length, rows, cols, rank, world):
        r0, r1 = tile_rows(rows, rank, world)
        out = torch.empty((3, r1 - r0, cols), dtype=torch.uint8, device=self.dev)
        _check(self.L.khg_tile_decode_dev(self.c._ctx, _ptr(comp), length, rows, cols, rank, world, _ptr(out[0]),
                                          _ptr(out[1]), _ptr(out[2]), _stream()), "tile_decode")
        st = self.c.last_status_after_decode()
        if st != 0:
            raise _lib.KhgError(st, "tile_decode")
        return out

    # ---- decode with the entropy decode split over the ranks
    def dec_map(self, comp, length, rows, cols, rank, world):
        self.shape = (rows, cols, world)
        self.rank = rank
        _check(self.L.khg_tile_decode_map_dev(self.c._ctx, _ptr(comp), length, rows, cols, rank, world, _stream()),
               "tile_decode_map")

    def blk_get(self):
        words = torch.empty(self.L.khg_tile_blk_words(self.c._ctx), dtype=torch.int32, device=self.dev)
        if words.numel():
            _check(self.L.khg_tile_blk_get(self.c._ctx, _ptr(words), _stream()), "tile_blk_get")
        return words

    def blk_set(self, words):
        if words.numel():
            _check(self.L.khg_tile_blk_set(self.c._ctx, _ptr(words), _stream()), "tile_blk_set")

    def dec_write(self):
        world = self.shape[2]
        bounds = (C.c_uint32 * (3 * (world + 1)))()
        _check(self.L.khg_tile_decode_write_dev(self.c._ctx, bounds, _stream()), "tile_decode_write")
        return list(bounds)

    def need_ranges(self, rows, cols, rank, world, plane):
        r = (C.c_uint32 * 8)()
        n = self.L.khg_tile_need_ranges(rows, cols, rank, world, plane, r)
        if n < 0:
            raise _lib.KhgError(n, "tile_need_ranges")
        return [(r[2 * i], r[2 * i + 1]) for i in range(n)]

    def sym_get(self, plane, first, count):
        t = torch.empty(count, dtype=torch.int16, device=self.dev)
        _check(self.L.khg_tile_sym_get(self.c._ctx, plane, first, count, _ptr(t), _stream()), "tile_sym_get")
        return t

    def sym_set(self, plane, first, t):
        _check(self.L.khg_tile_sym_set(self.c._ctx, plane, first, t.numel(), _ptr(t), _stream()), "tile_sym_set")

    def sym_buffer(self, count):
        return torch.empty(count, dtype=torch.int16, device=self.dev)

    def dec_inverse(self):
        rows, cols, world = self.shape
        r0, r1 = tile_rows(rows, self.rank, world)
        out = torch.empty((3, r1 - r0, cols), dtype=torch.uint8, device=self.dev)
        _check(self.L.khg_tile_decode_inverse_dev(self.c._ctx, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream()),
               "tile_decode_inverse")
        return out, self.c.last_status_after_decode()


class DistCollectives:
    """The three collectives over a torch.distributed process group (RCCL on GPUs, gloo on CPUs)."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def all_reduce_sum(self, t):
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_min(self, t):
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)

    def reduce_sum(self, t, dst):
        if self.on:
            dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=self.group)

    def exchange(self, sends, recvs):
        """sends: [(dst rank, tensor)], recvs: [(src rank, tensor)]; between any two ranks the tensors are matched in
        list order."""
        if not self.on:
            return
        ops = [dist.P2POp(dist.isend, t, dst, self.group) for dst, t in sends]
        ops += [dist.P2POp(dist.irecv, t, src, self.group) for src, t in recvs]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


def _word_runs(sel, w0, w1, head=None, tail=None):
    """Word ranges [s, e) of the file that the chunks picked by `sel` touch: one per stretch of consecutive picked
    chunks, in file order, cut so that no word appears twice.  head / tail: a range put in front / behind."""
    prev = torch.zeros_like(sel)
    prev[1:] = sel[:-1]
    nxt = torch.zeros_like(sel)
    nxt[:-1] = sel[1:]
    s, e = w0[sel & ~prev], w1[sel & ~nxt]
    if head is not None:
        s, e = torch.cat([s.new_tensor([head[0]]), s]), torch.cat([e.new_tensor([head[1]]), e])
    if tail is not None:
        s, e = torch.cat([s, s.new_tensor([tail[0]])]), torch.cat([e, e.new_tensor([tail[1]])])
    if s.numel():
        before = torch.cat([e.new_zeros(1), e[:-1]])  # two stretches of one rank can meet in a word
        s = torch.maximum(s, before)
    keep = e > s
    return s[keep], e[keep]


def _run_words(s, e):
    """The word indices of the ranges [s, e), in order."""
    lens = e - s
    total = int(lens.sum().item()) if lens.numel() else 0
    if total == 0:
        return torch.zeros(0, dtype=torch.int64, device=s.device)
    first = torch.cumsum(lens, 0) - lens
    return torch.repeat_interleave(s - first, lens) + torch.arange(total, dtype=torch.int64, device=s.device)


def gather_shares(words, own, bits, off, rank, world, coll, dst):
    """The file from the ranks' shares with every byte moved once.  words: this rank's share as int32 words (zero
    where it owns nothing); own: which chunks are this rank's; bits, off: every chunk's bit count and bit offset
    in the file (the same on all ranks).  A rank sends the words its chunks touch -- rank 0 also the header's and the
    tables' -- and `dst` adds what it receives in place: where two ranks' chunks meet inside a word both send it,
    and their bits are disjoint.  Returns the words of the file on `dst`, None elsewhere."""
    nwords = words.numel()
    filled = bits > 0
    owner = torch.where(own & filled, torch.full_like(bits, rank + 1), torch.zeros_like(bits))
    coll.reduce_sum(owner, dst)  # chunk -> rank + 1 (only dst looks at it)
    w0 = off >> 5
    w1 = (off + bits.to(torch.int64) + 31) >> 5
    stream_end = int((off + bits.to(torch.int64)).max().item()) if bits.numel() else 36 * 8

    def indices(r, sel):
        head = (0, 9) if r == 0 else None  # Writer.h:9-23: the 36-byte header
        tail = (stream_end >> 5, nwords) if r == 0 else None  # the streams' last word, pad bits, tables
        return _run_words(*_word_runs(sel, w0, w1, head, tail))

    if rank != dst:
        idx = indices(rank, own & filled)
        coll.exchange([(dst, words[idx].contiguous())], [])
        return None
    idxs = [indices(r, (owner == r + 1) if r != rank else (own & filled)) for r in range(world)]
    bufs = [torch.empty(idxs[r].numel(), dtype=words.dtype, device=words.device) for r in range(world)]
    coll.exchange([], [(r, bufs[r]) for r in range(world) if r != rank])
    out = torch.zeros_like(words)
    for r in range(world):
        out.index_add_(0, idxs[r], words[idxs[r]] if r == rank else bufs[r])
    return out


def _agreed(coll, device, rank, what, step):
    """Runs step() and makes its failure every rank's failure.  A step that raises KhgError on one rank only (a busy
    context, a launch that failed, no memory) must not leave the other ranks waiting in the next collective: the
    ranks exchange the step's code (all-reduce MIN, codes are negative) and raise together.  Returns step()'s value."""
    code, value = 0, None
    try:
        value = step()
    except _lib.KhgError as e:
        code = e.code
    st = torch.tensor([code], dtype=torch.int32, device=device)
    coll.all_reduce_min(st)
    worst = int(st.item())
    if worst != 0:
        raise _lib.KhgError(worst, "%s (rank %d saw status %d)" % (what, rank, code))
    return value


def encode_tiled(backend, strip, rows, cols, lvl, rank, world, coll=None, dst=0, merge="reduce"):
    """All ranks call this with their strip (image rows input_rows(rows, rank, world)).  Returns the bytes of the
    .comp file as a uint8 tensor on rank `dst` (None elsewhere) -- the same bytes a single GPU produces.
    merge: "reduce" adds the ranks' whole zero-padded shares up in one collective of the file's size; "gather" moves
    every byte of the file once, point to point (gather_shares: an eighth of the link traffic on eight ranks, paid
    for with index arithmetic on every rank).
    Every group of steps between two collectives is agreed on (_agreed): a failure on any rank raises on all."""
    coll = coll or DistCollectives()
    dev = strip.device

    def first():
        backend.forward(strip, rows, cols, lvl, rank, world)
        return backend.hist_get()

    cnt, first_idx = _agreed(coll, dev, rank, "encode_tiled: forward", first)
    # first raster indices: 0xFFFFFFFF (= -1 as int32) marks a symbol this tile never saw; raster indices stay
    # below 2^31 (checked by the library), so for the MIN reduce the mark becomes the largest int32
    first_idx = torch.where(first_idx < 0, torch.full_like(first_idx, _ABSENT), first_idx)
    coll.all_reduce_sum(cnt)
    coll.all_reduce_min(first_idx)
    first_idx = torch.where(first_idx == _ABSENT, torch.full_like(first_idx, -1), first_idx)

    def second():
        backend.hist_set(cnt, first_idx)
        backend.lens()
        return backend.bits_get()

    bits = _agreed(coll, dev, rank, "encode_tiled: code lengths", second)
    own = bits != 0  # before the all-reduce only this rank's chunks have a bit count
    coll.all_reduce_sum(bits)

    def third():
        # The device-side status (a symbol out of range, a stream of 2^31 bits) may show on one rank only, like a
        # host-side failure: it joins the agreement, so that all ranks raise together instead of one raising and
        # the others waiting in the byte reduce for ever.
        backend.bits_set(bits)
        share, total, status = backend.pack(backend.comp_bound(rows, cols))
        if status != 0:
            raise _lib.KhgError(status, "tile_pack")
        return share, total, backend.chunk_off_get() if merge == "gather" else None

    share, total, off = _agreed(coll, dev, rank, "encode_tiled: pack", third)
    words = share[: (total + 3) & ~3].view(torch.int32)
    if merge == "gather":
        words = gather_shares(words, own, bits, off, rank, world, coll, dst)
    else:
        coll.reduce_sum(words, dst)  # the shares have no set bit in common: their sum is their OR
    if rank != dst:
        return None
    return words.view(torch.uint8)[:total]


def symbol_pieces(bounds, need, world):
    """Who sends what to whom after the split entropy decode.  bounds[p][t] = raster index of the first symbol of
    plane p that rank t holds (bounds[p][world] = symbols decoded); need[t][p] = raster ranges rank t's inverse
    transform reads.  Returns [(src, dst, plane, first, count)] in the one order every rank uses."""
    out = []
    for dst in range(world):
        for p in range(3):
            for a, b in need[dst][p]:
                for src in range(world):
                    if src == dst:
                        continue
                    lo, hi = max(a, bounds[p][src]), min(b, bounds[p][src + 1])
                    if lo < hi:
                        out.append((src, dst, p, lo, hi - lo))
    return out


def decode_tiled(backend, comp, length, rows, cols, rank, world, coll=None, split=None):
    """comp: the whole file on every rank.  Returns the rank's rows of the picture, (3, rows/world, cols) uint8.
    split (default: world > 1): every rank entropy-decodes 1/world of each plane's stream and the symbols are
    exchanged; otherwise every rank decodes the whole stream."""
    if split is None:
        split = world > 1
    if not split:
        return backend.decode(comp, length, rows, cols, rank, world)
    coll = coll or DistCollectives()
    dev = comp.device

    def first():
        backend.dec_map(comp, length, rows, cols, rank, world)
        return backend.blk_get()

    words = _agreed(coll, dev, rank, "decode_tiled: map", first)
    coll.all_reduce_sum(words)  # other ranks' records are zero here: the sum is the union

    def second():
        backend.blk_set(words)
        flat = backend.dec_write()  # identical on every rank (the chain runs on the summed records)
        bounds = [flat[p * (world + 1):(p + 1) * (world + 1)] for p in range(3)]
        need = [[backend.need_ranges(rows, cols, t, world, p) for p in range(3)] for t in range(world)]
        sends, recvs, landing = [], [], []
        for src, dst, p, first_sym, count in symbol_pieces(bounds, need, world):
            if src == rank:
                sends.append((dst, backend.sym_get(p, first_sym, count)))
            elif dst == rank:
                buf = backend.sym_buffer(count)
                recvs.append((src, buf))
                landing.append((p, first_sym, buf))
        return sends, recvs, landing

    sends, recvs, landing = _agreed(coll, dev, rank, "decode_tiled: write", second)
    coll.exchange(sends, recvs)

    def third():
        for p, first_sym, buf in landing:
            backend.sym_set(p, first_sym, buf)
        out, status = backend.dec_inverse()
        if status != 0:  # a damaged stream may show on the rank that decoded that part only
            raise _lib.KhgError(status, "tile_decode_inverse")
        return out

    return _agreed(coll, dev, rank, "decode_tiled: inverse", third)


# ---- the library's own host for the tiled mode

UNIQUE_ID_BYTES = 128


class Comm:
    """A khg_comm.  Comm.rccl(rank, world, device, group): every rank of the torch.distributed group calls it; the
    group only carries rank 0's RCCL id to the others (any backend).  Comm.local_group(n) / Comm.local(group, rank):
    ranks that are threads of this process on one GPU."""

    def __init__(self, handle, rank, world):
        self._h, self.rank, self.world = handle, rank, world

    @classmethod
    def rccl(cls, rank, world, device, group=None):
        L = _lib.load()
        ident = torch.zeros(UNIQUE_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            _check(L.khg_comm_unique_id(C.c_void_p(ident.data_ptr()), UNIQUE_ID_BYTES), "comm_unique_id")
        if world > 1:
            box = [ident.numpy().tobytes()]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = torch.frombuffer(bytearray(box[0]), dtype=torch.uint8)
        h = C.c_void_p()
        _check(L.khg_comm_create_rccl(C.byref(h), rank, world, C.c_void_p(ident.data_ptr()), UNIQUE_ID_BYTES, device),
               "comm_create_rccl")
        return cls(h, rank, world)

    @staticmethod
    def local_group(world):
        g = _lib.load().khg_comm_local_group(world)
        if not g:
            raise ValueError("a local group has 1 to 8 ranks")
        return g

    @classmethod
    def local(cls, group, rank, world):
        h = C.c_void_p()
        _check(_lib.load().khg_comm_create_local(C.byref(h), group, rank), "comm_create_local")
        return cls(h, rank, world)

    def close(self):
        if self._h:
            _lib.load().khg_comm_destroy(self._h)
            self._h = None


def exchange_halos_native(codec, comm, own, rows):
    """own: (3, rows/world, cols) uint8 device tensor -> the strip encode_tiled_native takes."""
    _, R, W = own.shape
    i0, i1 = input_rows(rows, comm.rank, comm.world)
    strip = torch.empty((3, i1 - i0, W), dtype=torch.uint8, device=own.device)
    _check(codec.L.khg_tile_exchange_halos(comm._h, _ptr(own[0]), _ptr(own[1]), _ptr(own[2]), rows, W, _ptr(strip[0]),
                                           _ptr(strip[1]), _ptr(strip[2]), _stream()), "tile_exchange_halos")
    return strip


def encode_tiled_native(codec, comm, strip, rows, cols, lvl, dst=0, merge="reduce"):
    """encode_tiled on the library's host: returns the file's bytes (uint8 device tensor) on rank dst, None elsewhere."""
    assert strip.is_cuda and strip.dtype == torch.uint8 and strip.is_contiguous()
    cap = int(codec.L.khg_comp_bound(rows, cols))
    out = torch.empty(cap, dtype=torch.uint8, device=strip.device)
    n = C.c_size_t(0)
    _check(codec.L.khg_tile_encode(codec._ctx, comm._h, _ptr(strip[0]), _ptr(strip[1]), _ptr(strip[2]), rows, cols, lvl,
                                   _ptr(out), cap, C.byref(n), dst, {"reduce": 0, "gather": 1}[merge], _stream()),
           "tile_encode")
    return out[: n.value] if comm.rank == dst else None


def decode_tiled_native(codec, comm, comp, length, rows, cols):
    """decode_tiled (split entropy decode) on the library's host: the rank's rows, (3, rows/world, cols) uint8."""
    r0, r1 = tile_rows(rows, comm.rank, comm.world)
    out = torch.empty((3, r1 - r0, cols), dtype=torch.uint8, device=comp.device)
    _check(codec.L.khg_tile_decode(codec._ctx, comm._h, _ptr(comp), length, rows, cols, _ptr(out[0]), _ptr(out[1]),
                                   _ptr(out[2]), _stream()), "tile_decode")
    return out
