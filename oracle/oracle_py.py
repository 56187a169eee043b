"""ctypes bindings for the CPU checker (oracle/khugle_oracle.c and oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(HERE, "libkhugle_oracle.so")
_REF_SO = os.path.join(HERE, "_ref", "libkhugle_ref.so")

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


class Header(C.Structure):
    _fields_ = [("y_len", C.c_int32), ("cb_len", C.c_int32), ("cr_len", C.c_int32),
                ("rle_y_len", C.c_int32), ("rle_cb_len", C.c_int32), ("rle_cr_len", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32),
                ("y_pad", C.c_uint8), ("cb_pad", C.c_uint8), ("cr_pad", C.c_uint8), ("lvl", C.c_uint8)]


def build(force=False):
    """Compile the C restatement (and oracle/_ref when the reference checkout is present)."""
    if force or not os.path.exists(_ORACLE_SO) or \
            os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "khugle_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_ORACLE_SO)
        L.khgo_encode.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.c_int]
        L.khgo_decode.argtypes = [_u8p, C.c_size_t, _u8p, _u8p, _u8p]
        L.khgo_peek.argtypes = [_u8p, C.c_size_t, C.POINTER(Header)]
        L.khgo_forward_planes.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, _f64p, _f64p, _f64p]
        L.khgo_symbols.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p]
        L.khgo_decode_planes.argtypes = [_u8p, C.c_size_t, _i32p, _i32p, _i32p, _f64p, _f64p, _f64p]
        L.khgo_symbol_table.argtypes = [_i32p, C.c_size_t, _i32p, _i32p, C.c_int]
        L.khgo_symbol_table.restype = C.c_long
        L.khgo_build_codes.argtypes = [_i32p, _i32p, C.c_long, _u64p, _u8p]
        L.khgo_rgb_to_ycbcr.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, _f64p, _f64p, _f64p]
        L.khgo_rgb_to_ycbcr.restype = None
        L.khgo_ycbcr_to_rgb.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_int, _u8p, _u8p, _u8p]
        L.khgo_ycbcr_to_rgb.restype = None
        L.khgo_fwt2d.argtypes = [_f64p, C.c_int, C.c_int, C.c_int]
        L.khgo_iwt2d.argtypes = [_f64p, C.c_int, C.c_int, C.c_int]
        L.khgo_divisor_plane.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _f64p]
        L.khgo_divisor_plane.restype = None
        L.khgo_ssim.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        L.khgo_ssim.restype = C.c_double
        L.khgo_ssim_blocks.argtypes = [_u8p, _u8p, C.c_int, C.c_int, _f64p]
        L.khgo_ssim_blocks.restype = C.c_long
        L.khgo_psnr.argtypes = [_u8p] * 6 + [C.c_int, C.c_int]
        L.khgo_psnr.restype = C.c_double
        L.khgo_vis_steps.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_char_p, C.c_char_p]
        L.khgo_downsample.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.khgo_vis_bars.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong), _u8p, _u8p, _u8p]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    """The reference's own headers, compiled (oracle/_ref).  None when not built."""
    global _ref
    if _ref is None and have_ref():
        R = C.CDLL(_REF_SO)
        R.kref_fwt2d.argtypes = [_f64p, C.c_int, C.c_int]
        R.kref_fwt2d.restype = None
        R.kref_iwt2d.argtypes = [_f64p, C.c_int, C.c_int]
        R.kref_iwt2d.restype = None
        R.kref_rle.argtypes = [_i32p, C.c_int, C.c_int, _i32p, _i32p]
        R.kref_rle.restype = C.c_long
        R.kref_codes.argtypes = [_i32p, _i32p, C.c_long, C.c_char_p, C.c_long]
        R.kref_codes.restype = C.c_long
        R.kref_encode_file.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_char_p]
        R.kref_decode_file.argtypes = [C.c_char_p, _u8p, _u8p, _u8p, C.POINTER(C.c_int)]
        R.kref_ssim.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        R.kref_ssim.restype = C.c_double
        _ref = R
    return _ref


def _planes(rgb):
    """rgb: (3, H, W) uint8 planar."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.ndim == 3 and rgb.shape[0] == 3
    return rgb[0], rgb[1], rgb[2], rgb.shape[1], rgb.shape[2]


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("khugle oracle error %d" % code)
        self.code = code


def encode(rgb, lvl, faithful=False):
    r, g, b, h, w = _planes(rgb)
    cap = 36 + 3 * h * w * 8 + 64  # generous: the oracle reports the size it needed
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib().khgo_encode(r, g, b, h, w, lvl, out, cap, C.byref(n), int(faithful))
    if rc:
        raise OracleError(rc)
    return out[:n.value].tobytes()


def peek(comp):
    buf = np.frombuffer(comp, np.uint8)
    hd = Header()
    rc = lib().khgo_peek(np.ascontiguousarray(buf), buf.size, C.byref(hd))
    if rc:
        raise OracleError(rc)
    return hd


def decode(comp):
    hd = peek(comp)
    buf = np.ascontiguousarray(np.frombuffer(comp, np.uint8))
    out = np.zeros((3, hd.h, hd.w), np.uint8)
    rc = lib().khgo_decode(buf, buf.size, out[0], out[1], out[2])
    if rc:
        raise OracleError(rc)
    return out


def rgb_to_ycbcr(rgb):
    """Colorspace.h:5-23 on u8 planes: Y full size, Cb / Cr at even rows and columns."""
    r, g, b, h, w = _planes(rgb)
    y = np.zeros((h, w)); cb = np.zeros((h // 2, w // 2)); cr = np.zeros((h // 2, w // 2))
    lib().khgo_rgb_to_ycbcr(r, g, b, h, w, y, cb, cr)
    return y, cb, cr


def forward_planes(rgb):
    r, g, b, h, w = _planes(rgb)
    y = np.zeros((h, w)); cb = np.zeros((h // 2, w // 2)); cr = np.zeros((h // 2, w // 2))
    rc = lib().khgo_forward_planes(r, g, b, h, w, y, cb, cr)
    if rc:
        raise OracleError(rc)
    return y, cb, cr


def symbols(rgb, lvl):
    r, g, b, h, w = _planes(rgb)
    qy = np.zeros((h, w), np.int32); qcb = np.zeros((h // 2, w // 2), np.int32); qcr = np.zeros((h // 2, w // 2), np.int32)
    rc = lib().khgo_symbols(r, g, b, h, w, lvl, qy, qcb, qcr)
    if rc:
        raise OracleError(rc)
    return qy, qcb, qcr


def decode_planes(comp):
    hd = peek(comp)
    buf = np.ascontiguousarray(np.frombuffer(comp, np.uint8))
    h, w = hd.h, hd.w
    qy = np.zeros((h, w), np.int32); qcb = np.zeros((h // 2, w // 2), np.int32); qcr = np.zeros((h // 2, w // 2), np.int32)
    y = np.zeros((h, w)); cb = np.zeros((h // 2, w // 2)); cr = np.zeros((h // 2, w // 2))
    rc = lib().khgo_decode_planes(buf, buf.size, qy, qcb, qcr, y, cb, cr)
    if rc:
        raise OracleError(rc)
    return (qy, qcb, qcr), (y, cb, cr)


def symbol_table(q, faithful=False):
    q = np.ascontiguousarray(q, np.int32).ravel()
    counts = np.zeros(q.size, np.int32); syms = np.zeros(q.size, np.int32)
    k = lib().khgo_symbol_table(q, q.size, counts, syms, int(faithful))
    if k < 0:
        raise OracleError(k)
    return counts[:k].copy(), syms[:k].copy()


def build_codes(counts, syms):
    counts = np.ascontiguousarray(counts, np.int32); syms = np.ascontiguousarray(syms, np.int32)
    codes = np.zeros(counts.size, np.uint64); lens = np.zeros(counts.size, np.uint8)
    rc = lib().khgo_build_codes(counts, syms, counts.size, codes, lens)
    if rc:
        raise OracleError(rc)
    return codes, lens


def ssim(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().khgo_ssim(a, b, a.shape[1], a.shape[0])


def ssim_blocks(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    out = np.zeros((a.shape[0] // 8) * (a.shape[1] // 8))
    n = lib().khgo_ssim_blocks(a, b, a.shape[1], a.shape[0], out)
    return out[:n]


def psnr(rgb_in, rgb_out):
    i = np.ascontiguousarray(rgb_in, np.uint8); o = np.ascontiguousarray(rgb_out, np.uint8)
    return lib().khgo_psnr(i[0], i[1], i[2], o[0], o[1], o[2], i.shape[2], i.shape[1])


def ycbcr_to_rgb(y, cb, cr):
    """YCbCr2RGB (Colorspace.h:25-36) and the conversion to unsigned char, on any planes."""
    y = np.ascontiguousarray(y, np.float64); cb = np.ascontiguousarray(cb, np.float64)
    cr = np.ascontiguousarray(cr, np.float64)
    h, w = y.shape
    out = np.zeros((3, h, w), np.uint8)
    lib().khgo_ycbcr_to_rgb(y, cb, cr, h, w, out[0], out[1], out[2])
    return out


def vis_coeffs(rgb):
    """Compress.h:108-141: the DWT coefficient planes through the colour conversion."""
    return ycbcr_to_rgb(*forward_planes(rgb))


def vis_dequant(comp):
    """Decompress.h:49-126, 234-262: the dequantised coefficients (before IWT2D) through the colour conversion."""
    hd = peek(comp)
    (qy, qcb, qcr), _ = decode_planes(comp)
    dy = np.zeros((hd.h, hd.w)); dc = np.zeros((hd.h // 2, hd.w // 2))
    lib().khgo_divisor_plane(hd.h, hd.w, hd.lvl, 0, dy)
    lib().khgo_divisor_plane(hd.h, hd.w, hd.lvl, 1, dc)
    return ycbcr_to_rgb(qy * dy, qcb * dc, qcr * dc)  # Decompress.h:105: one product per coefficient


def vis_steps(h, w, lvl, tables=(0, 0, 0)):
    """(picture, "Step Size ..." line, "RLE Size ..." line) of Compress.h:245-300."""
    out = np.zeros((3, h, w), np.uint8)
    s1 = C.create_string_buffer(100); s2 = C.create_string_buffer(100)
    rc = lib().khgo_vis_steps(h, w, lvl, tables[0], tables[1], tables[2], out[0].ctypes.data, out[1].ctypes.data,
                              out[2].ctypes.data, s1, s2)
    if rc:
        raise OracleError(rc)
    return out, s1.value.decode(), s2.value.decode()


def downsample(plane, dst_h=256, dst_w=256):
    plane = np.ascontiguousarray(plane, np.uint8)
    out = np.zeros((dst_h, dst_w), np.uint8)
    lib().khgo_downsample(plane, plane.shape[0], plane.shape[1], out, dst_h, dst_w)
    return out


def vis_bars(h, w, lens):
    out = np.zeros((3, h, w), np.uint8)
    arr = (C.c_longlong * 3)(*[int(v) for v in lens])
    lib().khgo_vis_bars(h, w, arr, out[0], out[1], out[2])
    return out


# ---- format 2 (no counterpart in the reference): the seek index behind a format-1 file, restated from its
# definition -- entry s of a plane names the first symbol whose codeword starts at or behind bit 256 * s of the
# plane's stream -- with the codeword positions taken from a plain cumulative sum of code lengths.
IDX_SUB_BITS = 256


def index_trailer(comp):
    """The bytes that format 2 appends to the format-1 file `comp` (the zero gap up to a multiple of 8 included)."""
    import struct
    hd = peek(comp)
    buf = np.frombuffer(comp, np.uint8)
    (qy, qcb, qcr), _ = decode_planes(comp)
    ks = [hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len]
    bits = [hd.y_len - hd.y_pad, hd.cb_len - hd.cb_pad, hd.cr_len - hd.cr_pad]
    table_off = 36 + (hd.y_len + hd.cb_len + hd.cr_len) // 8
    firsts, offs, nsyms, nents = [], [], [], []
    for p, q in enumerate((qy, qcb, qcr)):
        tab = np.frombuffer(buf[table_off:table_off + 8 * ks[p]].tobytes(), "<i4").reshape(ks[p], 2)
        table_off += 8 * ks[p]
        counts, syms = tab[:, 0].copy(), tab[:, 1].copy()
        _, lens = build_codes(counts, syms)
        len_of = dict(zip(syms.tolist(), lens.tolist()))
        flat = q.ravel()
        nsym = int(counts.sum()) if bits[p] else 0  # an empty stream decodes to nothing (flat pictures)
        nsym = min(nsym, flat.size)
        sl = np.array([len_of[v] for v in flat[:nsym].tolist()], np.int64)
        starts = np.concatenate([[0], np.cumsum(sl)[:-1]]) if nsym else np.zeros(0, np.int64)
        assert (int(sl.sum()) if nsym else 0) == bits[p]
        nent = (bits[p] + IDX_SUB_BITS - 1) // IDX_SUB_BITS
        first = np.full(nent, nsym, np.uint32)
        off = np.full(nent, 255, np.uint8)
        for e in range(nent):
            i = int(np.searchsorted(starts, e * IDX_SUB_BITS, side="left"))
            if i < nsym and starts[i] < (e + 1) * IDX_SUB_BITS:
                first[e] = i
                off[e] = starts[i] - e * IDX_SUB_BITS
        firsts.append(first); offs.append(off); nsyms.append(nsym); nents.append(nent)
    ne = sum(nents)
    head = struct.pack("<18I", 0x3249484B, 1, IDX_SUB_BITS, hd.h, hd.w, hd.lvl, bits[0], 0, bits[1], 0, bits[2], 0,
                       *nsyms, *nents)
    body = b"".join(f.astype("<u4").tobytes() for f in firsts) + b"".join(o.tobytes() for o in offs)
    body += b"\0" * ((4 - ne % 4) % 4)
    total = len(head) + len(body) + 8
    gap = b"\0" * ((8 - len(comp) % 8) % 8)
    return gap + head + body + struct.pack("<II", total, 0x4B484932)
