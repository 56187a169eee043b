"""BMP ingest/egest for the codec harness.

Host-side file I/O only (no codec arithmetic).  Produces and consumes the planar,
top-down (3, H, W) uint8 R/G/B layout that the reference's CKhuGleSignal holds in
m_Red/m_Green/m_Blue after ReadBmp (KhuGleSignal.cpp:134-275): uncompressed 24-bpp
and 1/4/8-bpp palettised files, rows padded to 4 bytes, stored bottom-up, pixel data
at bfOffBits, palette directly after the 40-byte info header.  write_bmp mirrors
SaveBmp (KhuGleSignal.cpp:277-334): 24-bpp, 2000 pixels per metre.
"""
import struct

import numpy as np


def read_bmp(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"BM":
        raise ValueError("%s: not a BMP file" % path)
    off_bits = struct.unpack_from("<I", data, 10)[0]
    width, height = struct.unpack_from("<ii", data, 18)
    bit_count = struct.unpack_from("<H", data, 28)[0]
    compression = struct.unpack_from("<I", data, 30)[0]
    clr_used = struct.unpack_from("<I", data, 46)[0]
    if compression != 0:
        raise ValueError("%s: compressed BMP not supported" % path)
    if width <= 0 or height <= 0:
        raise ValueError("%s: unsupported dimensions" % path)
    if bit_count == 24:
        stride = (width * 3 + 3) // 4 * 4
        rows = np.frombuffer(data, np.uint8, stride * height, off_bits).reshape(height, stride)
        bgr = rows[::-1, :width * 3].reshape(height, width, 3)
        return np.ascontiguousarray(bgr[:, :, ::-1].transpose(2, 0, 1))
    if bit_count in (1, 4, 8):
        ncol = clr_used if clr_used else 1 << bit_count
        pal = np.frombuffer(data, np.uint8, ncol * 4, 54).reshape(ncol, 4)  # B, G, R, reserved
        stride = ((width * bit_count + 7) // 8 + 3) // 4 * 4
        rows = np.frombuffer(data, np.uint8, stride * height, off_bits).reshape(height, stride)
        if bit_count == 8:
            idx = rows[:, :width]
        else:
            per = 8 // bit_count
            shifts = np.arange(per - 1, -1, -1, dtype=np.uint8) * bit_count
            idx = ((rows[:, :, None] >> shifts) & ((1 << bit_count) - 1)).reshape(height, -1)[:, :width]
        idx = idx[::-1]
        return np.ascontiguousarray(np.stack([pal[idx, 2], pal[idx, 1], pal[idx, 0]]))
    raise ValueError("%s: %d-bpp BMP not supported" % (path, bit_count))


def bmp24_pixels(path):
    """The pixel array of an uncompressed 24-bit BMP exactly as it lies in the file -- (bytes, stride, width, height),
    B, G, R per pixel, rows bottom-up, `stride` bytes per row -- or None for any other kind of BMP.  This is what
    Codec.encode_bgr_dev takes: ReadBmp's conversion loop (KhuGleSignal.cpp:264-272) then happens on the GPU."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"BM":
        raise ValueError("%s: not a BMP file" % path)
    off_bits = struct.unpack_from("<I", data, 10)[0]
    width, height = struct.unpack_from("<ii", data, 18)
    bit_count = struct.unpack_from("<H", data, 28)[0]
    compression = struct.unpack_from("<I", data, 30)[0]
    if bit_count != 24 or compression != 0 or width <= 0 or height <= 0:
        return None
    stride = (width * 3 + 3) // 4 * 4
    if off_bits + stride * height > len(data):
        raise ValueError("%s: truncated BMP" % path)
    return np.frombuffer(data, np.uint8, stride * height, off_bits), stride, width, height


def bmp24_headers(width, height):
    """The two headers SaveBmp puts in front of the pixel array (KhuGleSignal.cpp:295-319)."""
    size = (width * 3 + 3) // 4 * 4 * height
    header = struct.pack("<2sIHHI", b"BM", 54 + size, 0, 0, 54)
    info = struct.pack("<IiiHHIIiiII", 40, width, height, 1, 24, 0, size, 2000, 2000, 0, 0)
    return header + info


def write_bmp24_pixels(path, pixels, width, height):
    """A 24-bit BMP from its pixel array as Codec.decode_bgr_dev leaves it (B, G, R, rows bottom-up, rows of
    (3 * width + 3) // 4 * 4 bytes)."""
    pixels = np.asarray(pixels, np.uint8).reshape(-1)
    if pixels.size != (width * 3 + 3) // 4 * 4 * height:
        raise ValueError("pixel array does not match %dx%d" % (width, height))
    with open(path, "wb") as f:
        f.write(bmp24_headers(width, height))
        f.write(pixels.tobytes())


def write_bmp(path, rgb):
    rgb = np.asarray(rgb, np.uint8)
    _, height, width = rgb.shape
    stride = (width * 3 + 3) // 4 * 4
    rows = np.zeros((height, stride), np.uint8)
    bgr = rgb[::-1].transpose(1, 2, 0)  # (H, W, [B, G, R]), top-down
    rows[:, :width * 3] = bgr[::-1].reshape(height, width * 3)  # bottom-up
    write_bmp24_pixels(path, rows, width, height)
