"""Python harness over the C ABI: device buffers and streams come from torch, every codec
operation runs in libkhugle_hip.so.  Counterpart of the reference application's
LoadBMPAndCompress / LoadComp / SaveAsComp callers (Main.cpp:180-345)."""
import ctypes as C

import numpy as np

from . import lib as _lib


def _check(rc, what=""):
    if rc != 0:
        raise _lib.KhgError(rc, what)


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Codec:
    """One khg_ctx: workspace for images up to max_rows x max_cols on one GPU."""

    def __init__(self, max_rows, max_cols, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("khugle_amd needs a HIP device (there is no CPU fallback)")
        self.L = _lib.load()
        self.device = device
        self.torch = torch
        torch.cuda.set_device(device)
        self._ctx = C.c_void_p(0)
        _check(self.L.khg_ctx_create(device, max_rows, max_cols, C.byref(self._ctx)), "ctx_create")
        self.max_rows, self.max_cols = max_rows, max_cols

    def close(self):
        if self._ctx:
            self.L.khg_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device-resident API (asynchronous on torch's current stream)
    def comp_bound(self, rows, cols):
        return self.L.khg_comp_bound(rows, cols)

    def encode_dev(self, rgb, lvl, out):
        """rgb: (3, H, W) uint8 cuda tensor; out: uint8 cuda tensor receiving the .comp bytes."""
        assert rgb.is_cuda and rgb.dtype == self.torch.uint8 and rgb.is_contiguous() and rgb.dim() == 3
        assert out.is_cuda and out.dtype == self.torch.uint8 and out.is_contiguous()
        _, h, w = rgb.shape
        _check(self.L.khg_encode_dev(self._ctx, _ptr(rgb[0]), _ptr(rgb[1]), _ptr(rgb[2]), h, w, lvl,
                                     _ptr(out), out.numel(), _stream()), "encode_dev")

    def encode_bgr_dev(self, pixels, stride, rows, cols, lvl, out):
        """pixels: the pixel array of a 24-bit BMP as a uint8 cuda tensor (B, G, R per pixel, rows bottom-up, `stride`
        bytes per row); out as for encode_dev.  The transform kernel reads the file's bytes itself."""
        assert pixels.is_cuda and pixels.dtype == self.torch.uint8 and pixels.is_contiguous()
        assert pixels.numel() >= stride * rows
        assert out.is_cuda and out.dtype == self.torch.uint8 and out.is_contiguous()
        _check(self.L.khg_encode_bgr_dev(self._ctx, _ptr(pixels), stride, rows, cols, lvl, _ptr(out), out.numel(),
                                         _stream()), "encode_bgr_dev")

    def encode_result(self):
        n = C.c_size_t(0)
        _check(self.L.khg_encode_result(self._ctx, C.byref(n)), "encode_result")
        return n.value

    def decode_dev(self, comp, length, rows, cols, rgb_out):
        assert comp.is_cuda and comp.dtype == self.torch.uint8 and comp.is_contiguous()
        assert rgb_out.is_cuda and rgb_out.dtype == self.torch.uint8 and rgb_out.is_contiguous()
        assert tuple(rgb_out.shape) == (3, rows, cols)
        _check(self.L.khg_decode_dev(self._ctx, _ptr(comp), length, rows, cols, _ptr(rgb_out[0]), _ptr(rgb_out[1]),
                                     _ptr(rgb_out[2]), _stream()), "decode_dev")

    def decode_bgr_dev(self, comp, length, rows, cols, pixels, stride):
        """Decode into the pixel array of a 24-bit BMP (uint8 cuda tensor of at least stride * rows bytes: B, G, R per
        pixel, rows bottom-up): SaveBmp's pixel loop (KhuGleSignal.cpp:285-293) happens in the last kernel."""
        assert comp.is_cuda and comp.dtype == self.torch.uint8 and comp.is_contiguous()
        assert pixels.is_cuda and pixels.dtype == self.torch.uint8 and pixels.is_contiguous()
        assert pixels.numel() >= stride * rows
        _check(self.L.khg_decode_bgr_dev(self._ctx, _ptr(comp), length, rows, cols, _ptr(pixels), stride, _stream()),
               "decode_bgr_dev")

    # ---- format 2: the same file followed by a seek index
    def set_format(self, fmt):
        """1: the reference's file (default).  2: followed by a seek index (encode buffers then need comp_bound() +
        index_bound() bytes; decode_indexed_dev reads such files)."""
        _check(self.L.khg_ctx_set_format(self._ctx, int(fmt)), "ctx_set_format")

    def index_bound(self, rows, cols):
        return int(self.L.khg_index_bound(rows, cols))

    def decode_indexed_dev(self, comp, length, rows, cols, rgb_out):
        assert comp.is_cuda and comp.dtype == self.torch.uint8 and comp.is_contiguous()
        assert rgb_out.is_cuda and rgb_out.dtype == self.torch.uint8 and rgb_out.is_contiguous()
        assert tuple(rgb_out.shape) == (3, rows, cols)
        _check(self.L.khg_decode_indexed_dev(self._ctx, _ptr(comp), length, rows, cols, _ptr(rgb_out[0]),
                                             _ptr(rgb_out[1]), _ptr(rgb_out[2]), _stream()), "decode_indexed_dev")

    # ---- what the application shows beside a result: two status lines and three pictures per direction
    def status_strings(self, rows, cols, lvl, tables):
        """("Step Size (min, max) : ...", "RLE Size (Y, Cb, Cr) : ...") of Compress.h:294-300 / Decompress.h:265-271;
        tables = the three table sizes (the header's rle_*_len)."""
        s1 = C.create_string_buffer(100)
        s2 = C.create_string_buffer(100)
        _check(self.L.khg_status_strings(rows, cols, lvl, int(tables[0]), int(tables[1]), int(tables[2]), s1, 100, s2,
                                         100), "status_strings")
        return s1.value.decode(), s2.value.decode()

    def vis_coeffs_dev(self, rgb):
        """Compress.h:108-141: the picture's DWT coefficient planes through YCbCr2RGB -> (3, H, W) uint8 cuda tensor."""
        assert rgb.is_cuda and rgb.dtype == self.torch.uint8 and rgb.is_contiguous()
        _, h, w = rgb.shape
        out = self.torch.empty_like(rgb)
        _check(self.L.khg_vis_coeffs_dev(self._ctx, _ptr(rgb[0]), _ptr(rgb[1]), _ptr(rgb[2]), h, w, _ptr(out[0]),
                                         _ptr(out[1]), _ptr(out[2]), _stream()), "vis_coeffs_dev")
        return out

    def vis_dequant_dev(self, comp, length, rows, cols):
        """Decompress.h:49-126, 234-262: a file's dequantised coefficients through YCbCr2RGB (no inverse transform)."""
        assert comp.is_cuda and comp.dtype == self.torch.uint8 and comp.is_contiguous()
        out = self.torch.empty((3, rows, cols), dtype=self.torch.uint8, device=comp.device)
        _check(self.L.khg_vis_dequant_dev(self._ctx, _ptr(comp), length, rows, cols, _ptr(out[0]), _ptr(out[1]),
                                          _ptr(out[2]), _stream()), "vis_dequant_dev")
        _check(self.last_status(), "vis_dequant_dev")
        return out

    def vis_steps_dev(self, rows, cols, lvl):
        """Compress.h:245-288: the quantiser's step sizes in red."""
        out = self.torch.empty((3, rows, cols), dtype=self.torch.uint8, device=self.torch.device("cuda", self.device))
        _check(self.L.khg_vis_steps_dev(rows, cols, lvl, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream()),
               "vis_steps_dev")
        return out

    def vis_bars_dev(self, rows, cols, len_bytes):
        """Compress.h:330-391 / Decompress.h:128-188: the three stream lengths (bytes) as bars."""
        out = self.torch.empty((3, rows, cols), dtype=self.torch.uint8, device=self.torch.device("cuda", self.device))
        arr = (C.c_longlong * 3)(*[int(v) for v in len_bytes])
        _check(self.L.khg_vis_bars_dev(rows, cols, arr, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream()),
               "vis_bars_dev")
        return out

    def downsample_dev(self, plane, dst_rows=256, dst_cols=256):
        """DownSampleImage (DownSample.h:25-37) of one (H, W) uint8 cuda plane."""
        assert plane.is_cuda and plane.dtype == self.torch.uint8 and plane.is_contiguous() and plane.dim() == 2
        out = self.torch.empty((dst_rows, dst_cols), dtype=self.torch.uint8, device=plane.device)
        _check(self.L.khg_downsample_dev(_ptr(plane), plane.shape[0], plane.shape[1], _ptr(out), dst_rows, dst_cols,
                                         _stream()), "downsample_dev")
        return out

    def last_status(self):
        return self.L.khg_last_status(self._ctx)

    def last_status_after_decode(self):
        """Waits for the context's last job and returns its device-side status (0 = ok)."""
        return self.L.khg_last_status(self._ctx)

    def debug_poison(self, byte=0xA5):
        """Test hook: fill the context's whole device workspace with `byte` (stale data of another job)."""
        _check(self.L.khg_debug_poison(self._ctx, byte), "debug_poison")

    def ssim_dev(self, a, b):
        """SSIM of two (H, W) uint8 cuda planes (the reference passes the R channel)."""
        h, w = a.shape
        v = C.c_double(0)
        _check(self.L.khg_ssim_dev(self._ctx, _ptr(a), _ptr(b), w, h, C.byref(v), _stream()), "ssim")
        return v.value

    def psnr_dev(self, rgb_in, rgb_out):
        _, h, w = rgb_in.shape
        v = C.c_double(0)
        _check(self.L.khg_psnr_dev(self._ctx, _ptr(rgb_in[0]), _ptr(rgb_in[1]), _ptr(rgb_in[2]), _ptr(rgb_out[0]),
                                   _ptr(rgb_out[1]), _ptr(rgb_out[2]), w, h, C.byref(v), _stream()), "psnr")
        return v.value

    def forward_planes_dev(self, rgb):
        t = self.torch
        _, h, w = rgb.shape
        y = t.empty((h, w), dtype=t.float64, device=rgb.device)
        cb = t.empty((h // 2, w // 2), dtype=t.float64, device=rgb.device)
        cr = t.empty((h // 2, w // 2), dtype=t.float64, device=rgb.device)
        _check(self.L.khg_forward_planes_dev(self._ctx, _ptr(rgb[0]), _ptr(rgb[1]), _ptr(rgb[2]), h, w, _ptr(y),
                                             _ptr(cb), _ptr(cr), _stream()), "forward_planes")
        return y, cb, cr

    def symbols_dev(self, rgb, lvl):
        t = self.torch
        _, h, w = rgb.shape
        qy = t.empty((h, w), dtype=t.int16, device=rgb.device)
        qcb = t.empty((h // 2, w // 2), dtype=t.int16, device=rgb.device)
        qcr = t.empty((h // 2, w // 2), dtype=t.int16, device=rgb.device)
        _check(self.L.khg_symbols_dev(self._ctx, _ptr(rgb[0]), _ptr(rgb[1]), _ptr(rgb[2]), h, w, lvl, _ptr(qy),
                                      _ptr(qcb), _ptr(qcr), _stream()), "symbols")
        return qy, qcb, qcr

    def profile_luma_forward(self, enable=True):
        """Start (or stop) timing the luma forward launch inside encode_dev with HIP events; resets the tally."""
        _check(self.L.khg_profile_luma_forward(self._ctx, 1 if enable else 0), "profile_luma_forward")

    def profile_read(self):
        """(event-to-event ms, device-clock ms, launches) summed since profile_luma_forward()."""
        ms = C.c_double(0)
        dev = C.c_double(0)
        n = C.c_int(0)
        _check(self.L.khg_profile_read(self._ctx, C.byref(ms), C.byref(dev), C.byref(n)), "profile_read")
        return ms.value, dev.value, n.value

    def time_luma_forward_dev(self, rgb, lvl, reps=10):
        """Mean duration (ms) of the level-1 luma forward kernel, HIP events on the current stream."""
        _, h, w = rgb.shape
        ms = C.c_float(0)
        _check(self.L.khg_time_luma_forward_dev(self._ctx, _ptr(rgb[0]), _ptr(rgb[1]), _ptr(rgb[2]), h, w, lvl, reps,
                                                C.byref(ms), _stream()), "time_luma_forward")
        return ms.value

    def synth_dev(self, rows, cols, seed, index, out=None):
        t = self.torch
        if out is None:
            out = t.empty((3, rows, cols), dtype=t.uint8, device="cuda:%d" % self.device)
        _check(self.L.khg_synth_dev(_ptr(out[0]), _ptr(out[1]), _ptr(out[2]), rows, cols, seed, index, _stream()),
               "synth")
        return out

    def synth_rows_dev(self, row0, rows, cols, seed, index):
        """Rows row0 .. row0+rows-1 of synthetic frame `index` of a `cols`-wide image, (3, rows, cols) uint8."""
        t = self.torch
        out = t.empty((3, rows, cols), dtype=t.uint8, device="cuda:%d" % self.device)
        _check(self.L.khg_synth_rows_dev(_ptr(out[0]), _ptr(out[1]), _ptr(out[2]), row0, rows, cols, seed, index,
                                         _stream()), "synth_rows")
        return out

    # ---- host-buffer API (synchronous; the library stages through its own device buffers)
    def encode(self, rgb, lvl):
        """rgb: (3, H, W) uint8 numpy -> bytes of the .comp file."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        _, h, w = rgb.shape
        cap = self.L.khg_comp_bound(h, w)
        if cap == 0:
            raise _lib.KhgError(_lib.ERR_DIMS, "encode")
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        _check(self.L.khg_encode(self._ctx, rgb[0].ctypes.data, rgb[1].ctypes.data, rgb[2].ctypes.data, h, w, lvl,
                                 out.ctypes.data, cap, C.byref(n)), "encode")
        return out[:n.value].tobytes()

    def encode_ycbcr(self, y, cb, cr, lvl, write_back=False):
        """CompressImage's own inputs: f64 planes (Y HxW, Cb / Cr H/2 x W/2) -> bytes of the .comp file.
        write_back=True also returns the DWT coefficient planes CompressImage leaves in its inputs."""
        y = np.array(y, np.float64, order="C"); cb = np.array(cb, np.float64, order="C"); cr = np.array(cr, np.float64, order="C")
        h, w = y.shape
        cap = self.L.khg_comp_bound(h, w)
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        _check(self.L.khg_encode_ycbcr(self._ctx, y.ctypes.data, cb.ctypes.data, cr.ctypes.data, h, w, lvl,
                                       out.ctypes.data, cap, C.byref(n), 1 if write_back else 0), "encode_ycbcr")
        comp = out[:n.value].tobytes()
        return (comp, (y, cb, cr)) if write_back else comp

    def decode_ycbcr(self, comp):
        """DecompressImage's own outputs: bytes of a .comp file -> f64 planes Y, Cb, Cr."""
        buf = np.frombuffer(comp, np.uint8)
        hd = peek(comp)
        y = np.empty((hd.h, hd.w)); cb = np.empty((hd.h // 2, hd.w // 2)); cr = np.empty((hd.h // 2, hd.w // 2))
        _check(self.L.khg_decode_ycbcr(self._ctx, buf.ctypes.data, buf.size, y.ctypes.data, cb.ctypes.data,
                                       cr.ctypes.data), "decode_ycbcr")
        return y, cb, cr

    def decode(self, comp):
        """bytes of a .comp file -> (3, H, W) uint8 numpy."""
        buf = np.frombuffer(comp, np.uint8)
        hd = peek(comp)
        out = np.empty((3, hd.h, hd.w), np.uint8)
        _check(self.L.khg_decode(self._ctx, buf.ctypes.data, buf.size, out[0].ctypes.data, out[1].ctypes.data,
                                 out[2].ctypes.data), "decode")
        return out


def peek(comp):
    """Header of a .comp byte string (validated)."""
    buf = np.frombuffer(comp, np.uint8)
    hd = _lib.Header()
    _check(_lib.load().khg_peek(buf.ctypes.data, buf.size, C.byref(hd)), "peek")
    return hd
