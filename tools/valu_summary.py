"""VALU pricing of the level-1 forward launch from the logs of tools/gpu_valu.sh:
    python tools/valu_summary.py LOGDIR OUT.json
ceiling_us = VALU wave-instructions of the launch / 1024 SIMDs x cycles one wave64 f64 instruction holds a SIMD's
issue slot (measured: the mean of v_add_f64, v_mul_f64, v_fma_f64 with four waves per SIMD) / the shader clock held
inside the launch's workgroups.  frac = ceiling_us / the launch's duration."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import stamp

SIMDS = 256 * 4


def main():
    d, out = sys.argv[1:3]
    rates = {}
    for line in open(os.path.join(d, "rates.log")):
        m = re.match(r"(\S+(?: \S+)?)\s+([\d.]+) ms\s+([\d.]+) ns per wave64 instruction and SIMD =\s+([\d.]+) cycles at the "
                     r"([\d.]+) GHz held", line)
        if m:
            rates[m.group(1)] = {"ns": float(m.group(3)), "cycles": float(m.group(4)), "ghz": float(m.group(5))}
    f64 = [rates[k] for k in ("v_add_f64", "v_mul_f64", "v_fma_f64")]
    cyc = sum(r["cycles"] for r in f64) / 3
    ns = sum(r["ns"] for r in f64) / 3
    probe_ghz = sum(r["ghz"] for r in f64) / 3
    clock = open(os.path.join(d, "clock.log")).read()
    fwd_ghz = float(re.search(r"shader clock held inside the workgroups: median ([\d.]+) GHz", clock).group(1))
    alone = open(os.path.join(d, "alone.log")).read()
    us = float(re.search(r"median ([\d.]+)", alone).group(1))
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "pmc", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "khg_fwd_kernel<0" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
    pmc = {k: acc[k] / n[k] for k in acc}
    insts = pmc["SQ_INSTS_VALU"]
    ceiling_us = insts / SIMDS * cyc / (fwd_ghz * 1e3)
    json.dump({
        "taken_at": stamp(),
        "kernel": "khg_fwd_kernel<0> on one 7680x4320 frame, High preset, alone on the GPU",
        "valu_wave_insts": int(insts),
        "other_wave_insts": {k[9:].lower(): int(v) for k, v in pmc.items() if k.startswith("SQ_INSTS_") and k != "SQ_INSTS_VALU"},
        "valu_busy_frac_of_wave_cycles": round(pmc["SQ_ACTIVE_INST_VALU"] * 4 / pmc["SQ_WAVE_CYCLES"], 3)
        if pmc.get("SQ_WAVE_CYCLES") else None,
        "f64_probe": {"cycles_per_wave_inst_per_simd": round(cyc, 2), "ns_per_wave_inst_per_simd": round(ns, 3),
                      "shader_ghz_held": round(probe_ghz, 3), "waves_per_simd": 4,
                      "per_instruction": {k: rates[k] for k in sorted(rates)},
                      "method": "tools/probe_rates.hip: 8 independent chains of one instruction per wave, 4 waves per "
                                "SIMD on every CU; s_memtime against s_memrealtime inside the kernel gives the clock"},
        "shader_ghz_held_in_kernel": fwd_ghz,
        "launch_us": us,
        "ceiling_us": round(ceiling_us, 1),
        "frac": round(ceiling_us / us, 3),
        "nominal_ns_at_2p4ghz_4_cycles": round(4 / 2.4, 3),
        "gap": "both: a wave64 f64 instruction holds the issue slot %.2f cycles, not 4 (issue), and the forward launch "
               "holds %.2f GHz where the probe holds %.2f GHz and the data sheet says 2.4 (clock)" % (cyc, fwd_ghz, probe_ghz),
    }, open(out, "w"), indent=1)
    print(open(out).read())


main()
