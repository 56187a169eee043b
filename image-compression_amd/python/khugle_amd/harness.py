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

    def __init__(self, comp, decoded, psnr, ssim, original_file_size, step_str="", rle_str="", pictures=None):
        self.comp = comp
        self.decoded = decoded
        self.psnr = psnr
        self.ssim = ssim
        self.file_size = len(comp)
        self.original_file_size = original_file_size
        self.step_str = step_str    # Compress.h:294-296
        self.rle_str = rle_str      # Compress.h:299-300
        self.pictures = pictures    # visualize=True: {"coefficients", "steps", "bars"} -> (3, H, W) u8

    @property
    def value_str(self):
        # Main.cpp:257-260
        return "%6.3f     %4.3f     %6.3fKB (%.3f %% of Original)" % (
            self.psnr, self.ssim, self.file_size / 1024, self.file_size / self.original_file_size * 100)


def _codec_for(h, w, codec):
    if codec is not None:
        return codec
    return Codec(h, w)


def _pictures_of(c, comp, rows, cols, coefficients, padded):
    """The three pictures beside a result: the coefficient picture given, the step sizes, the stream lengths.
    padded: CompressImage draws the lengths with the pad bits, DecompressImage without."""
    hd = peek(comp)
    bits = [hd.y_len, hd.cb_len, hd.cr_len]
    pads = [0, 0, 0] if padded else [hd.y_pad, hd.cb_pad, hd.cr_pad]
    lens = [(b - p) // 8 for b, p in zip(bits, pads)]
    return {"coefficients": coefficients.cpu().numpy(),
            "steps": c.vis_steps_dev(rows, cols, hd.lvl).cpu().numpy(),
            "bars": c.vis_bars_dev(rows, cols, lens).cpu().numpy()}


def load_bmp_and_compress(path, lvl, codec=None, visualize=False):
    """Main.cpp:180-279.  lvl: 0 Normal, 1 High, 2 Extreme (or the names).  visualize: also the three pictures the
    application puts beside the result (Compress.h:116-141, 245-288, 330-391)."""
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
    hd = peek(comp)
    step_str, rle_str = c.status_strings(h, w, lvl, (hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len))
    pictures = _pictures_of(c, comp, h, w, c.vis_coeffs_dev(d_in), padded=True) if visualize else None
    return CompressReport(comp, decoded, psnr, ssim, os.path.getsize(path), step_str, rle_str, pictures)


def save_comp(report, path):
    """Menu "Save compressed image": WriteAll (Writer.h:43-69)."""
    with open(path, "wb") as f:
        f.write(report.comp)


def load_comp(path, codec=None, visualize=False):
    """Main.cpp:281-334: returns the (3, H, W) u8 picture and the status line ("%6.3fKB"); with visualize also
    {"step_str", "rle_str", "pictures"}: DecompressImage's two lines and three pictures (Decompress.h:128-271)."""
    with open(path, "rb") as f:
        comp = f.read()
    hd = peek(comp)  # MeasureImageSize (Reader.h:97-110) plus validation
    c = _codec_for(hd.h, hd.w, codec)
    rgb = c.decode(comp)
    line = "%6.3fKB" % (len(comp) / 1024)
    if not visualize:
        return rgb, line
    t = c.torch
    padded = np.zeros((len(comp) + 3) // 4 * 4, np.uint8)
    padded[:len(comp)] = np.frombuffer(comp, np.uint8)
    d_comp = t.from_numpy(padded).cuda(c.device)
    step_str, rle_str = c.status_strings(hd.h, hd.w, hd.lvl, (hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len))
    pictures = _pictures_of(c, comp, hd.h, hd.w, c.vis_dequant_dev(d_comp, len(comp), hd.h, hd.w), padded=False)
    return rgb, line, {"step_str": step_str, "rle_str": rle_str, "pictures": pictures}


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
    a.add_argument("--pictures", help="directory for the three pictures the application shows beside the result")
    b = sub.add_parser("decompress", help=".comp -> BMP")
    b.add_argument("comp")
    b.add_argument("bmp")
    b.add_argument("--pictures", help="directory for the three pictures the application shows beside the result")
    args = ap.parse_args(argv)

    def save_pictures(directory, pictures):
        os.makedirs(directory, exist_ok=True)
        for name, rgb in pictures.items():
            write_bmp(os.path.join(directory, name + ".bmp"), rgb)

    if args.cmd == "compress":
        rep = load_bmp_and_compress(args.bmp, args.preset, visualize=bool(args.pictures))
        save_comp(rep, args.comp)
        if args.save_decoded:
            write_bmp(args.save_decoded, rep.decoded)
        if args.pictures:
            save_pictures(args.pictures, rep.pictures)
        print(HEADER_STR)
        print(rep.value_str)
        print(rep.step_str)
        print(rep.rle_str)
    else:
        line = comp_to_bmp(args.comp, args.bmp)
        print("FILE SIZE")
        print(line)
        if args.pictures:
            _, _, extra = load_comp(args.comp, visualize=True)
            save_pictures(args.pictures, extra["pictures"])
            print(extra["step_str"])
            print(extra["rle_str"])
    return 0
