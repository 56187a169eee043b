"""Loader of the test-only kernel simulator (tests/sim/khg_sim.cpp, built with -DKHG_SIM)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "sim", "khg_sim.cpp")
SO = os.path.join(HERE, "sim", "libkhg_sim.so")
CSRC = os.path.join(ROOT, "image-compression_amd", "csrc")

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_lib = None


def build(force=False):
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        # KHG_PS_BATCH_BITS: the single-pass packer's LDS batches hold 20000 bits here instead of 114688, so that the
        # entries of ordinary pictures are assembled in one batch (detail rows) and in several (rows with LL symbols)
        # KHG_CHAIN_GROUPS: the long-stream form of the decoder's chain (KHGSIM_CHAIN_SPLIT=1) takes batches of 32
        # workgroup maps instead of 256, so that pictures of a megapixel already span several
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I" + CSRC,
                               "-DKHG_PS_BATCH_BITS=20000", "-DKHG_CHAIN_GROUPS=2", "-o", SO, SRC])


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(SO)
        L.khgsim_encode.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t,
                                    C.POINTER(C.c_size_t)]
        L.khgsim_encode_bgr.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t,
                                        C.POINTER(C.c_size_t)]
        L.khgsim_forward.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.khgsim_decode.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _u8p, _u8p, _u8p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.POINTER(C.c_int)]
        L.khgsim_decode_bgr.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t]
        L.khgsim_ssim.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_void_p]
        L.khgsim_ssim.restype = C.c_double
        L.khgsim_encode_indexed.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t,
                                            C.POINTER(C.c_size_t)]
        L.khgsim_decode_indexed.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _u8p, _u8p, _u8p, C.c_int]
        L.khgsim_downsample.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.khgsim_vis_coeffs.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, _u8p, _u8p, _u8p]
        L.khgsim_vis_dequant.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _u8p, _u8p, _u8p]
        L.khgsim_vis_steps.argtypes = [C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p, C.c_void_p]
        L.khgsim_vis_bars.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong), _u8p, _u8p, _u8p]
        L.khgsim_ordered_mean.argtypes = [np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_size_t,
                                          C.c_void_p]
        L.khgsim_ordered_mean.restype = C.c_double
        L.khgsim_synth.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
        L.khgsim_synth.restype = None
        _f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.khgsim_div_const.argtypes = [_f64p, C.c_size_t, C.c_double, _f64p]
        L.khgsim_div_const.restype = None
        L.khgsim_round_quot.argtypes = [_f64p, _f64p, C.c_size_t, _f64p, _u8p]
        L.khgsim_round_quot.restype = None
        _vp = C.c_void_p
        L.khgsim_encode_ycbcr.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t,
                                          C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_void_p]
        L.khgsim_decode_ycbcr.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _f64p, _f64p, _f64p]
        L.khgsim_ctx_create.argtypes = [C.c_int, C.c_int]
        L.khgsim_ctx_create.restype = _vp
        L.khgsim_ctx_destroy.argtypes = [_vp]
        L.khgsim_ctx_destroy.restype = None
        L.khgsim_ctx_poison.argtypes = [_vp, C.c_int]
        L.khgsim_ctx_poison.restype = None
        L.khgsim_ctx_encode.argtypes = [_vp, _u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t,
                                        C.POINTER(C.c_size_t)]
        L.khgsim_ctx_decode.argtypes = [_vp, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p, _u8p, _u8p]
        L.khgsim_tile_create.argtypes = [C.c_int, C.c_int]
        L.khgsim_tile_create.restype = _vp
        L.khgsim_tile_destroy.argtypes = [_vp]
        L.khgsim_tile_destroy.restype = None
        L.khgsim_tile_chunk_count.argtypes = [C.c_int, C.c_int, C.c_int]
        L.khgsim_tile_chunk_count.restype = C.c_size_t
        L.khgsim_tile_forward.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        for f in ("hist_get", "hist_set"):
            getattr(L, "khgsim_tile_" + f).argtypes = [_vp, _vp, _vp]
            getattr(L, "khgsim_tile_" + f).restype = None
        L.khgsim_tile_lens.argtypes = [_vp]
        L.khgsim_tile_lens.restype = None
        for f in ("bits_get", "bits_set", "chunk_off_get"):
            getattr(L, "khgsim_tile_" + f).argtypes = [_vp, _vp]
            getattr(L, "khgsim_tile_" + f).restype = None
        L.khgsim_tile_pack.argtypes = [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.khgsim_tile_decode.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]
        L.khgsim_tile_decode_map.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.khgsim_tile_blk_words.argtypes = [_vp]
        L.khgsim_tile_blk_words.restype = C.c_size_t
        for f in ("blk_get", "blk_set", "decode_write"):
            getattr(L, "khgsim_tile_" + f).argtypes = [_vp, _vp]
        L.khgsim_tile_need_ranges.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
        for f in ("sym_get", "sym_set"):
            getattr(L, "khgsim_tile_" + f).argtypes = [_vp, C.c_int, C.c_uint32, C.c_uint32, _vp]
            getattr(L, "khgsim_tile_" + f).restype = None
        L.khgsim_tile_decode_inverse.argtypes = [_vp, _vp, _vp, _vp]
        _lib = L
    return _lib


def encode(rgb, lvl, cap=None):
    _, h, w = rgb.shape
    if cap is None:
        cap = (36 + h * w * 3 + 3 * 16384 * 8 + 3) // 4 * 4
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib().khgsim_encode(rgb[0], rgb[1], rgb[2], h, w, lvl, out, cap, C.byref(n))
    return rc, out[:n.value].tobytes() if rc == 0 else b""


def pack_symbols(qy, qcb, qcr, lvl):
    """The entropy coder alone (table, single-pass packer, header) on the caller's int16 symbol planes."""
    h, w = qy.shape
    cap = (36 + h * w * 12 + 3 * 16384 * 8 + 3) // 4 * 4
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    q = [np.ascontiguousarray(a, np.int16) for a in (qy, qcb, qcr)]
    L = lib()
    L.khgsim_pack_symbols.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [_u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    rc = L.khgsim_pack_symbols(q[0].ctypes.data, q[1].ctypes.data, q[2].ctypes.data, h, w, lvl, out, cap, C.byref(n))
    return rc, out[:n.value].tobytes() if rc == 0 else b""


def bgr_pixels(rgb):
    """The pixel array of the 24-bit BMP that holds `rgb` ((3, H, W) u8): B, G, R per pixel, rows bottom-up, rows
    padded to a multiple of 4 bytes (pad bytes filled with 0xAA: nothing may depend on them)."""
    _, h, w = rgb.shape
    stride = (3 * w + 3) // 4 * 4
    rows = np.full((h, stride), 0xAA, np.uint8)
    rows[:, :3 * w] = rgb[::-1][:, ::-1].transpose(1, 2, 0).reshape(h, 3 * w)  # channel order B, G, R; bottom-up
    return rows.reshape(-1), stride


def encode_bgr(rgb, lvl):
    _, h, w = rgb.shape
    pixels, stride = bgr_pixels(rgb)
    cap = (36 + h * w * 3 + 3 * 16384 * 8 + 3) // 4 * 4
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib().khgsim_encode_bgr(pixels, stride, h, w, lvl, out, cap, C.byref(n))
    return rc, out[:n.value].tobytes() if rc == 0 else b""


def forward(rgb, lvl):
    _, h, w = rgb.shape
    y = np.zeros((h, w)); cb = np.zeros((h // 2, w // 2)); cr = np.zeros((h // 2, w // 2))
    qy = np.zeros((h, w), np.int16); qcb = np.zeros((h // 2, w // 2), np.int16); qcr = np.zeros((h // 2, w // 2), np.int16)
    rc = lib().khgsim_forward(rgb[0], rgb[1], rgb[2], h, w, lvl, *[a.ctypes.data for a in (y, cb, cr, qy, qcb, qcr)])
    return rc, (y, cb, cr), (qy, qcb, qcr)


def decode(comp, rows, cols):
    buf = np.frombuffer(comp, np.uint8).copy()
    out = np.zeros((3, rows, cols), np.uint8)
    qy = np.zeros((rows, cols), np.int16); qcb = np.zeros((rows // 2, cols // 2), np.int16)
    qcr = np.zeros((rows // 2, cols // 2), np.int16)
    passes = C.c_int(0)
    rc = lib().khgsim_decode(buf, buf.size, rows, cols, out[0], out[1], out[2], qy.ctypes.data, qcb.ctypes.data,
                             qcr.ctypes.data, C.byref(passes))
    return rc, out, (qy, qcb, qcr), passes.value


def decode_bgr(comp, rows, cols, stride=None, fill=0x5C):
    """The decoded picture as a 24-bit BMP pixel array (rows of `stride` bytes, pre-filled with `fill`)."""
    buf = np.frombuffer(comp, np.uint8).copy()
    stride = stride or 3 * cols
    pixels = np.full(rows * stride, fill, np.uint8)
    rc = lib().khgsim_decode_bgr(buf, buf.size, rows, cols, pixels, stride)
    return rc, pixels


def ssim(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().khgsim_ssim(a, b, a.shape[1], a.shape[0], None)


def encode_indexed(rgb, lvl):
    """Format 2: the format-1 file followed by its seek index."""
    _, h, w = rgb.shape
    cap = (36 + h * w * 3 + 3 * 16384 * 8 + 3) // 4 * 4 + 4096 + h * w
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib().khgsim_encode_indexed(rgb[0], rgb[1], rgb[2], h, w, lvl, out, cap, C.byref(n))
    return rc, out[:n.value].tobytes() if rc == 0 else b""


def decode_indexed(comp, rows, cols, poison=0xA5):
    buf = np.frombuffer(comp, np.uint8).copy()
    out = np.zeros((3, rows, cols), np.uint8)
    rc = lib().khgsim_decode_indexed(buf, buf.size, rows, cols, out[0], out[1], out[2], poison)
    return rc, out


def downsample(plane, dst_h=256, dst_w=256):
    plane = np.ascontiguousarray(plane, np.uint8)
    out = np.zeros((dst_h, dst_w), np.uint8)
    lib().khgsim_downsample(plane, plane.shape[0], plane.shape[1], out, dst_h, dst_w)
    return out


def vis_coeffs(rgb):
    out = np.zeros_like(rgb)
    rc = lib().khgsim_vis_coeffs(rgb[0], rgb[1], rgb[2], rgb.shape[1], rgb.shape[2], out[0], out[1], out[2])
    return rc, out


def vis_dequant(comp, rows, cols):
    buf = np.frombuffer(comp, np.uint8).copy()
    out = np.zeros((3, rows, cols), np.uint8)
    rc = lib().khgsim_vis_dequant(buf, buf.size, rows, cols, out[0], out[1], out[2])
    return rc, out


def vis_steps(rows, cols, lvl):
    out = np.full((3, rows, cols), 7, np.uint8)
    lo_hi = np.zeros(2)
    rc = lib().khgsim_vis_steps(rows, cols, lvl, out[0], out[1], out[2], lo_hi.ctypes.data)
    return rc, out, (float(lo_hi[0]), float(lo_hi[1]))


def vis_bars(rows, cols, lens):
    out = np.full((3, rows, cols), 7, np.uint8)
    arr = (C.c_longlong * 3)(*[int(v) for v in lens])
    rc = lib().khgsim_vis_bars(rows, cols, arr, out[0], out[1], out[2])
    return rc, out


def ordered_mean(x):
    """(mean, [integer windows, scanned windows, values added singly]) of the mean kernel on the doubles x."""
    x = np.ascontiguousarray(x, np.float64)
    stat = np.zeros(3, np.uint64)
    m = lib().khgsim_ordered_mean(x, x.size, stat.ctypes.data)
    return m, [int(v) for v in stat]


def synth(rows, cols, seed, index):
    out = np.zeros((3, rows, cols), np.uint8)
    lib().khgsim_synth(out[0], out[1], out[2], rows, cols, seed, index)
    return out


def plan_groups(groups, slots):
    """groups: [(pw, span, weight)] -> (workgroups, [(seg_rows, nseg, nstrip_wg, wg0)]) from khg_plan_groups."""
    L = lib()
    n = len(groups)
    gin = (C.c_int * (3 * n))(*[v for g in groups for v in g])
    out = (C.c_int * (4 * n))()
    L.khgsim_plan_groups.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    wgs = L.khgsim_plan_groups(gin, n, slots, out)
    return wgs, [tuple(out[4 * i:4 * i + 4]) for i in range(n)]


def segment_rows(begin, end, nseg, seg):
    r0, r1 = C.c_int(0), C.c_int(0)
    L = lib()
    L.khgsim_segment_rows.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2
    L.khgsim_segment_rows.restype = None
    L.khgsim_segment_rows(begin, end, nseg, seg, C.byref(r0), C.byref(r1))
    return r0.value, r1.value


def div_const(x, c):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib().khgsim_div_const(x, x.size, c, out)
    return out


def round_quot(x, d):
    x = np.ascontiguousarray(x, np.float64); d = np.ascontiguousarray(d, np.float64)
    out = np.empty_like(x); fast = np.empty(x.size, np.uint8)
    lib().khgsim_round_quot(x, d, x.size, out, fast)
    return out, fast


class SimCtx:
    """One long-lived simulator workspace (what a khg_ctx is on the GPU): jobs of any size up to its maximum."""

    def __init__(self, rows, cols):
        self.h = lib().khgsim_ctx_create(rows, cols)
        assert self.h

    def __del__(self):
        try:
            lib().khgsim_ctx_destroy(self.h)
        except Exception:
            pass

    def poison(self, byte=0xA5):
        lib().khgsim_ctx_poison(self.h, byte)

    def encode(self, rgb, lvl):
        _, h, w = rgb.shape
        cap = (36 + h * w * 3 + 3 * 16384 * 8 + 3) // 4 * 4
        out = np.zeros(cap, np.uint8)
        n = C.c_size_t(0)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        rc = lib().khgsim_ctx_encode(self.h, rgb[0], rgb[1], rgb[2], h, w, lvl, out, cap, C.byref(n))
        return rc, out[:n.value].tobytes() if rc == 0 else b""

    def decode(self, comp, rows, cols):
        buf = np.frombuffer(comp, np.uint8).copy()
        out = np.zeros((3, rows, cols), np.uint8)
        rc = lib().khgsim_ctx_decode(self.h, buf, buf.size, rows, cols, out[0], out[1], out[2])
        return rc, out


class SimTileBackend:
    """khugle_amd.tiled backend on the kernel simulator (CPU tensors): one rank of a row-tiled image."""

    def __init__(self, rows, cols):
        import torch
        self.torch = torch
        self.h = lib().khgsim_tile_create(rows, cols)
        assert self.h
        self.rows, self.cols = rows, cols

    def __del__(self):
        try:
            lib().khgsim_tile_destroy(self.h)
        except Exception:
            pass

    def comp_bound(self, rows, cols):
        return (36 + rows * cols * 3 // 2 * 15 // 8 + 3 * 16384 * 8 + 64) & ~3

    def forward(self, strip, rows, cols, lvl, rank, world):
        assert strip.dtype == self.torch.uint8 and strip.is_contiguous()
        self.world = world
        rc = lib().khgsim_tile_forward(self.h, strip[0].data_ptr(), strip[1].data_ptr(), strip[2].data_ptr(), rows,
                                       cols, lvl, rank, world)
        assert rc == 0, rc

    def hist_get(self):
        t = self.torch
        cnt = t.empty(3 * 16384, dtype=t.int32); first = t.empty(3 * 16384, dtype=t.int32)
        lib().khgsim_tile_hist_get(self.h, cnt.data_ptr(), first.data_ptr())
        return cnt, first

    def hist_set(self, cnt, first):
        lib().khgsim_tile_hist_set(self.h, cnt.contiguous().data_ptr(), first.contiguous().data_ptr())

    def lens(self):
        lib().khgsim_tile_lens(self.h)

    def bits_get(self):
        bits = self.torch.empty(lib().khgsim_tile_chunk_count(self.rows, self.cols, self.world), dtype=self.torch.int32)
        lib().khgsim_tile_bits_get(self.h, bits.data_ptr())
        return bits

    def bits_set(self, bits):
        lib().khgsim_tile_bits_set(self.h, bits.contiguous().data_ptr())

    def chunk_off_get(self):
        off = self.torch.empty(lib().khgsim_tile_chunk_count(self.rows, self.cols, self.world), dtype=self.torch.int64)
        lib().khgsim_tile_chunk_off_get(self.h, off.data_ptr())
        return off

    def pack(self, cap):
        share = self.torch.empty(cap, dtype=self.torch.uint8)
        total = C.c_size_t(0)
        rc = lib().khgsim_tile_pack(self.h, share.data_ptr(), cap, C.byref(total))
        return share, total.value, rc

    def decode(self, comp, length, rows, cols, rank, world):
        out = self.torch.empty((3, rows // world, cols), dtype=self.torch.uint8)
        rc = lib().khgsim_tile_decode(self.h, comp.data_ptr(), length, rows, cols, rank, world, out[0].data_ptr(),
                                      out[1].data_ptr(), out[2].data_ptr())
        assert rc == 0, rc
        return out

    # ---- decode with the entropy decode split over the ranks
    def dec_map(self, comp, length, rows, cols, rank, world):
        self.world, self.rank = world, rank
        rc = lib().khgsim_tile_decode_map(self.h, comp.data_ptr(), length, rows, cols, rank, world)
        assert rc == 0, rc

    def blk_get(self):
        L = lib()
        words = self.torch.empty(L.khgsim_tile_blk_words(self.h), dtype=self.torch.int32)
        L.khgsim_tile_blk_get(self.h, words.data_ptr())
        return words

    def blk_set(self, words):
        lib().khgsim_tile_blk_set(self.h, words.contiguous().data_ptr())

    def dec_write(self):
        bounds = (C.c_uint32 * (3 * (self.world + 1)))()
        self.write_status = lib().khgsim_tile_decode_write(self.h, bounds)
        return list(bounds)

    def need_ranges(self, rows, cols, rank, world, plane):
        r = (C.c_uint32 * 8)()
        n = lib().khgsim_tile_need_ranges(rows, cols, rank, world, plane, r)
        return [(r[2 * i], r[2 * i + 1]) for i in range(n)]

    def sym_get(self, plane, first, count):
        t = self.torch.empty(count, dtype=self.torch.int16)
        lib().khgsim_tile_sym_get(self.h, plane, first, count, t.data_ptr())
        return t

    def sym_set(self, plane, first, t):
        lib().khgsim_tile_sym_set(self.h, plane, first, t.numel(), t.contiguous().data_ptr())

    def sym_buffer(self, count):
        return self.torch.empty(count, dtype=self.torch.int16)

    def dec_inverse(self):
        out = self.torch.empty((3, self.rows // self.world, self.cols), dtype=self.torch.uint8)
        rc = lib().khgsim_tile_decode_inverse(self.h, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        return out, rc


def encode_ycbcr(y, cb, cr, lvl, want_coef=False):
    y = np.ascontiguousarray(y, np.float64); cb = np.ascontiguousarray(cb, np.float64); cr = np.ascontiguousarray(cr, np.float64)
    rows, cols = y.shape
    cap = (36 + rows * cols * 3 + 3 * 16384 * 8 + 64) & ~3
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    co = [np.zeros_like(p) for p in (y, cb, cr)] if want_coef else [None] * 3
    rc = lib().khgsim_encode_ycbcr(y, cb, cr, rows, cols, lvl, out, cap, C.byref(n),
                                   *[c.ctypes.data if c is not None else None for c in co])
    return rc, out[:n.value].tobytes(), co


def decode_ycbcr(comp, rows, cols):
    buf = np.frombuffer(comp, np.uint8).copy()
    y = np.zeros((rows, cols)); cb = np.zeros((rows // 2, cols // 2)); cr = np.zeros((rows // 2, cols // 2))
    rc = lib().khgsim_decode_ycbcr(buf, buf.size, rows, cols, y, cb, cr)
    return rc, y, cb, cr
