"""Counterpart of the reference's two user-facing operations, CImageProcessing::LoadBMPAndCompress
(Main.cpp:180-279) and CImageProcessing::LoadComp (Main.cpp:281-334), without the GUI: read a BMP, encode it,
decode it again, and report the figures the application shows under "PSNR  SSIM  FILE SIZE".

All codec and metric work runs on the GPU through the C ABI; this module only moves files and formats text.
"""
import os

import numpy as np

from .bmp import bmp24_pixels, read_bmp, write_bmp, write_bmp24_pixels
from .codec import Codec, peek

PRESETS = {"normal": 0, "high": 1, "extreme": 2}
HEADER_STR = "PSNR         SSIM        FILE SIZE"  # Main.cpp:255


class CompressReport:
    """What LoadBMPAndCompress leaves behind: the file image, the round-tripped picture and the status line."""

    def __init__(self, comp, decoded, psnr, ssim, original_file_size):
        self.comp = comp
        self.decoded = decoded
        self.psnr = psnr
        self.ssim = ssim
        self.file_size = len(comp)
        self.original_file_size = original_file_size

    @property
    def value_str(self):
        # Main.cpp:257-260
        return "%6.3f     %4.3f     %6.3fKB (%.3f %% of Original)" % (
            self.psnr, self.ssim, self.file_size / 1024, self.file_size / self.original_file_size * 100)


def _codec_for(h, w, codec):
    if codec is not None:
        return codec
    return Codec(h, w)


def load_bmp_and_compress(path, lvl, codec=None):
    """Main.cpp:180-279.  lvl: 0 Normal, 1 High, 2 Extreme (or the names)."""
    lvl = PRESETS.get(str(lvl).lower(), lvl)
    rgb = read_bmp(path)
    _, h, w = rgb.shape
    c = _codec_for(h, w, codec)
    t = c.torch
    raw = bmp24_pixels(path)
    if raw is not None:  # 24-bit file: its pixel array goes to the GPU as it is (B, G, R, bottom-up)
        pixels, stride, _, _ = raw
        d_pix = t.from_numpy(pixels.copy()).cuda(c.device)
        d_comp = t.zeros(c.comp_bound(h, w), dtype=t.uint8, device=d_pix.device)
        c.encode_bgr_dev(d_pix, stride, h, w, lvl, d_comp)
        comp = d_comp[:c.encode_result()].cpu().numpy().tobytes()
    else:  # palettised: expanded to planar R, G, B on the host
        comp = c.encode(rgb, lvl)
    decoded = c.decode(comp)  # the application decodes its own result to show it and to measure it
    d_in = t.from_numpy(np.ascontiguousarray(rgb)).cuda(c.device)
    d_out = t.from_numpy(decoded).cuda(c.device)
    psnr = c.psnr_dev(d_in, d_out)          # GetPsnr on R, G, B (Main.cpp:241-248)
    ssim = c.ssim_dev(d_in[0], d_out[0])    # ComputeSSIM on the R channel only (Main.cpp:250-253)
    return CompressReport(comp, decoded, psnr, ssim, os.path.getsize(path))


def save_comp(report, path):
    """Menu "Save compressed image": WriteAll (Writer.h:43-69)."""
    with open(path, "wb") as f:
        f.write(report.comp)


def load_comp(path, codec=None):
    """Main.cpp:281-334: returns the (3, H, W) u8 picture and the status line ("%6.3fKB")."""
    with open(path, "rb") as f:
        comp = f.read()
    hd = peek(comp)  # MeasureImageSize (Reader.h:97-110) plus validation
    c = _codec_for(hd.h, hd.w, codec)
    rgb = c.decode(comp)
    return rgb, "%6.3fKB" % (len(comp) / 1024)


def comp_to_bmp(comp_path, bmp_path, codec=None):
    """LoadComp followed by the menu's "Save BMP" (SaveBmp, KhuGleSignal.cpp:277-334) without the planar picture in
    between: the last decode kernel writes the file's pixel array, the host adds the two headers.  Returns LoadComp's
    status line."""
    with open(comp_path, "rb") as f:
        comp = f.read()
    hd = peek(comp)
    c = _codec_for(hd.h, hd.w, codec)
    t = c.torch
    padded = np.zeros((len(comp) + 3) // 4 * 4, np.uint8)
    padded[:len(comp)] = np.frombuffer(comp, np.uint8)
    d_comp = t.from_numpy(padded).cuda(c.device)
    stride = 3 * hd.w  # a multiple of 4: widths are multiples of 8
    d_pix = t.empty(stride * hd.h, dtype=t.uint8, device=d_comp.device)
    c.decode_bgr_dev(d_comp, len(comp), hd.h, hd.w, d_pix, stride)
    rc = c.last_status()
    if rc:
        raise RuntimeError("decode_bgr_dev: status %d" % rc)
    write_bmp24_pixels(bmp_path, d_pix.cpu().numpy(), hd.w, hd.h)
    return "%6.3fKB" % (len(comp) / 1024)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m khugle_amd", description="bmp <-> comp on an MI355X")
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("compress", help="BMP -> .comp (prints the application's PSNR / SSIM / size line)")
    a.add_argument("bmp")
    a.add_argument("comp")
    a.add_argument("--preset", default="high", choices=sorted(PRESETS))
    a.add_argument("--save-decoded", help="also write the round-tripped picture as a 24-bit BMP")
    b = sub.add_parser("decompress", help=".comp -> BMP")
    b.add_argument("comp")
    b.add_argument("bmp")
    args = ap.parse_args(argv)
    if args.cmd == "compress":
        rep = load_bmp_and_compress(args.bmp, args.preset)
        save_comp(rep, args.comp)
        if args.save_decoded:
            write_bmp(args.save_decoded, rep.decoded)
        print(HEADER_STR)
        print(rep.value_str)
    else:
        line = comp_to_bmp(args.comp, args.bmp)
        print("FILE SIZE")
        print(line)
    return 0
