"""ctypes loader for libkhugle_hip.so (the C ABI declared in include/khugle_hip.h).

The library is built in-tree by image-compression_amd/csrc/Makefile (hipcc, gfx950).  There is
no CPU fallback: if the library is missing, or no HIP device is usable, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc"))
LIB_PATH = os.path.join(CSRC, "libkhugle_hip.so")

OK, ERR_DIMS, ERR_CAP, ERR_FORMAT, ERR_RANGE, ERR_NOMEM, ERR_HIP, ERR_ARG, ERR_BUSY = 0, -1, -2, -3, -4, -5, -6, -7, -8


class KhgError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = "status %d" % code
        try:
            msg = load().khg_status_string(code).decode()
        except Exception:
            pass
        super().__init__("khugle_hip%s: %s (%d)" % (" " + what if what else "", msg, code))


class Header(C.Structure):
    _fields_ = [("y_len", C.c_int32), ("cb_len", C.c_int32), ("cr_len", C.c_int32),
                ("rle_y_len", C.c_int32), ("rle_cb_len", C.c_int32), ("rle_cr_len", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32),
                ("y_pad", C.c_uint8), ("cb_pad", C.c_uint8), ("cr_pad", C.c_uint8), ("lvl", C.c_uint8)]


# every symbol include/khugle_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("khg_ctx_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("khg_ctx_destroy", None, [_P]),
    ("khg_status_string", C.c_char_p, [C.c_int]),
    ("khg_comp_bound", C.c_size_t, [C.c_int, C.c_int]),
    ("khg_encode", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("khg_encode_dev", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    ("khg_ctx_set_format", C.c_int, [_P, C.c_int]),
    ("khg_index_bound", C.c_size_t, [C.c_int, C.c_int]),
    ("khg_decode_indexed_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_status_strings", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                     C.c_char_p, C.c_size_t]),
    ("khg_vis_coeffs_dev", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_vis_dequant_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_vis_steps_dev", C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_vis_bars_dev", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_longlong), _P, _P, _P, _P]),
    ("khg_downsample_dev", C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    ("khg_ordered_mean_dev", C.c_int, [_P, C.c_size_t, _P, _P]),
    ("khg_decode_bgr_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    ("khg_encode_bgr_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    ("khg_encode_result", C.c_int, [_P, C.POINTER(C.c_size_t)]),
    ("khg_peek", C.c_int, [_P, C.c_size_t, C.POINTER(Header)]),
    ("khg_decode", C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
    ("khg_decode_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_last_status", C.c_int, [_P]),
    ("khg_encode_ycbcr", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.c_int]),
    ("khg_decode_ycbcr", C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
    ("khg_encode_ycbcr_dev", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P, _P, _P, _P]),
    ("khg_decode_ycbcr_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_batch_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("khg_batch_destroy", None, [_P]),
    ("khg_batch_encode", C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_batch_decode", C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    ("khg_tile_input_rows", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("khg_tile_chunk_count", C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    ("khg_tile_forward_dev", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("khg_tile_hist_get", C.c_int, [_P, _P, _P, _P]),
    ("khg_tile_hist_set", C.c_int, [_P, _P, _P, _P]),
    ("khg_tile_lens_dev", C.c_int, [_P, _P]),
    ("khg_tile_bits_get", C.c_int, [_P, _P, _P]),
    ("khg_tile_chunk_off_get", C.c_int, [_P, _P, _P]),
    ("khg_tile_bits_set", C.c_int, [_P, _P, _P]),
    ("khg_tile_pack_dev", C.c_int, [_P, _P, C.c_size_t, _P]),
    ("khg_tile_decode_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_tile_decode_map_dev", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("khg_tile_blk_words", C.c_size_t, [_P]),
    ("khg_tile_blk_get", C.c_int, [_P, _P, _P]),
    ("khg_tile_blk_set", C.c_int, [_P, _P, _P]),
    ("khg_tile_decode_write_dev", C.c_int, [_P, _P, _P]),
    ("khg_tile_need_ranges", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("khg_tile_sym_get", C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32, _P, _P]),
    ("khg_tile_sym_set", C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32, _P, _P]),
    ("khg_tile_decode_inverse_dev", C.c_int, [_P, _P, _P, _P, _P]),
    ("khg_comm_unique_id", C.c_int, [_P, C.c_size_t]),
    ("khg_comm_create_rccl", C.c_int, [C.POINTER(_P), C.c_int, C.c_int, _P, C.c_size_t, C.c_int]),
    ("khg_comm_wrap_rccl", C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int]),
    ("khg_comm_local_group", _P, [C.c_int]),
    ("khg_comm_create_local", C.c_int, [C.POINTER(_P), _P, C.c_int]),
    ("khg_comm_destroy", None, [_P]),
    ("khg_tile_exchange_halos", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_tile_encode", C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.c_int, C.c_int, _P]),
    ("khg_tile_decode", C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_profile_luma_forward", C.c_int, [_P, C.c_int]),
    ("khg_profile_read", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("khg_ssim_dev", C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_double), _P]),
    ("khg_ssim_static_dev", C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    ("khg_ssim_static_prepare", C.c_int, []),
    ("khg_psnr_dev", C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_double), _P]),
    ("khg_forward_planes_dev", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_symbols_dev", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("khg_time_luma_forward_dev", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    ("khg_debug_poison", C.c_int, [_P, C.c_int]),
    ("khg_synth_rows_dev", C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _P]),
    ("khg_synth_dev", C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _P]),
]

_lib = None


def load():
    """Load the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C %s` (hipcc, gfx950); "
                              "there is no CPU fallback" % (LIB_PATH, CSRC))
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
