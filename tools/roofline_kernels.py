"""Per-kernel roofline table from rocprofv3 output (tools/gpu_roofline.sh collects it):
    python tools/roofline_kernels.py KERNEL_STATS.csv FETCH_COUNTERS.csv WRITE_COUNTERS.csv COMP_BYTES OUT.json
For every kernel of the codec: algorithmic bytes per launch (compulsory reads + writes of that launch on a
7680 x 4320 frame, High preset), average duration, achieved GB/s against the 8 TB/s HBM peak, and the HBM bytes the
counters saw per launch (FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled, the gfx950 correction for
wide streaming reads -- the calibration kernels listed in the output show how well that holds here)."""
import csv
import json
import sys
from collections import defaultdict

PEAK = 8000.0  # GB/s


def model(px, comp):
    """Algorithmic bytes per launch; px = luma pixels, comp = .comp bytes."""
    c = float(comp)
    walk = c * 1.25  # walk tables: 256 subsequences x (longest code ~20) x 4 B per 16 KiB of stream
    return {
        "khg_fwd_kernel<0>": (8.25 * px, "level 1 of Y, Cb, Cr: 3 RGB read + 3 LL f64 + 2.25 symbols written"),
        "khg_fwd_kernel<2>": (3.75 * px, "level 2 of Y, Cb, Cr: 3 LL f64 read + 0.75 symbols written"),
        "khg_lens_kernel": (3.0 * px, "symbols read"),
        "khg_pack_kernel": (3.0 * px + c, "symbols read + stream written"),
        "khg_packscan_kernel": (3.0 * px + c, "symbols read + stream written (single pass: no measuring pass)"),
        "khg_dec_map_kernel<32>": (c + walk, "stream read + walk tables written"),
        "khg_dec_write_kernel": (c + c / 16 + 3.0 * px, "stream + walk column read, symbols written"),
        "khg_inv_kernel<0>": (3.75 * px, "level 2: 0.75 symbols read + 3 LL f64 written"),
        "khg_inv_kernel<1>": (5.75 * px, "Cb, Cr level 1: 0.75 symbols + 1 LL read, 4 f64 planes written"),
        "khg_inv_kernel<2>": (10.5 * px, "luma level 1 + RGB: 1.5 symbols + 2 LL + 4 chroma read, 3 RGB written"),
        "khg_inv_kernel<4>": (8.25 * px, "level 1 of Y, Cb, Cr + RGB in one launch: 2.25 symbols + 3 LL f64 read, 3 RGB written"),
        "khg_ssim_kernel": (2.0 * px + px / 8.0, "two planes read, one f64 per block written"),
        "khg_sse_kernel": (6.0 * px, "six planes read"),
        "khg_synth_kernel": (3.0 * px, "three planes written (calibration)"),
    }


def short(name):
    """khg_fwd_kernel<0, 2, 1, false>(KhgFwdArgs) -> khg_fwd_kernel<0>: the first template argument says which
    launch it is (level, source); the others are tuning variants of the same launch."""
    n = name.replace("void ", "").split("(")[0]
    if n.startswith("khg_fwd_kernel<") and "," in n:
        n = n.split(",")[0] + ">"
    return n


def main():
    stats, fetch, write, comp, out = sys.argv[1:6]
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    from source_stamp import stamp
    px = 4320 * 7680
    dur = {}
    for r in csv.DictReader(open(stats)):
        dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    ctr = {}
    for path, cname in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        acc, n = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == cname:
                acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
                n[short(r["Kernel_Name"])] += 1
        ctr[cname] = {k: acc[k] / n[k] for k in acc}
    m = model(px, int(comp))
    table = {}
    for k in sorted(dur, key=lambda k: -dur[k][0] * dur[k][1]):
        if not k.startswith("khg"):
            continue
        us, calls = dur[k]
        row = {"avg_us": round(us, 1), "calls": calls}
        f, w = ctr["FETCH_SIZE"].get(k), ctr["WRITE_SIZE"].get(k)
        if f is not None and w is not None:
            row["fetch_bytes"] = int(2 * f * 1024)
            row["write_bytes"] = int(w * 1024)
            row["traffic_bytes"] = row["fetch_bytes"] + row["write_bytes"]
        if k in m:
            b, what = m[k]
            row["algorithmic_bytes"] = int(b)
            row["what"] = what
            row["achieved_gbps"] = round(b / us / 1e3, 1)
            row["frac"] = round(b / us / 1e3 / PEAK, 4)
        else:
            row["what"] = "latency class (one or a few workgroups): not priced against HBM"
        table[k] = row
    json.dump({"taken_at": stamp(),
               "frame": "7680x4320 synthetic, High preset, one stream", "comp_bytes": int(comp), "peak_gbps": PEAK,
               "counter_note": "fetch_bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction), write_bytes = WRITE_SIZE x 1024, "
                               "each from its own rocprofv3 --pmc pass",
               "kernels": table}, open(out, "w"), indent=1)
    for k, r in table.items():
        print("%-28s %8.1f us  alg %12s  traffic %12s  frac %s" % (k, r["avg_us"], r.get("algorithmic_bytes", "-"),
                                                                 r.get("traffic_bytes", "-"), r.get("frac", "-")))


main()
