#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the KhuGle codec path on MI355X.

Workload (BASELINE.json config "Synthetic 7680x4320 RGB batch of 1024 frames, High preset,
batch-sharded over 1/2/4/8 GPUs"): every rank owns 128 distinct synthetic 8K frames (1024 / 8),
already resident in HBM.  One step = encode all of them to .comp bytes, then decode all of them
back to RGB (device to device).  Frames are independent, so ranks exchange nothing on the data
path (weak scaling); torch.distributed (RCCL) only provides the barrier and the max-over-ranks.

Prints ONE JSON line on rank 0.  `value` = megapixels that went through encode AND decode per
second, whole job.  `roofline` prices the longest data-parallel kernel (the level-1 forward launch:
colour conversion, DWT level 1, quantiser and histogram of Y, Cb and Cr) against the 8 TB/s HBM
peak with timing taken live; `cpu_baseline` times the reference's own DWT / Huffman / file code
(oracle/_ref, built from the reference's headers) on the host cores on a bounded sample: one
5760x5760 frame (the square 8K-class shape the reference can round-trip) per core.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "image-compression_amd", "python"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

ROWS, COLS, LVL, SEED = 4320, 7680, 1, 2025
HBM_PEAK_GBS = 8000.0
CPU_POOL_CAP = 16  # CPUs a one-GPU job may use on the GPU boxes
PROFILE_TAG = "r03"  # profiles/<tag>_kernels.json, profiles/<tag>_valu.json
# Level-1 forward launch (colour + DWT level 1 + quantiser + histogram of Y, Cb and Cr in one kernel, the longest
# data-parallel kernel of the path), algorithmic bytes per luma pixel:
#   3     u8 R, G, B read once
# + 3     LL quadrants written as f64 for level 2 (8 B on a quarter of the 1.5 samples per pixel)
# + 2.25  int16 symbols of the three detail sub-bands (2 B on three quarters of the 1.5 samples)
L1_FWD_BYTES_PER_PX = 8.25
# the same launch priced by the pass-structured model (one kernel per stage, f64 planes between the stages): colour
# conversion (3 read + 12 written), DWT level 1 (12 + 12) and quantisation of the detail bands (9 read + 2.25 written)
L1_FWD_STAGED_BYTES_PER_PX = 15 + 24 + 11.25


def _cpu_worker(args):
    """One host process: encode+decode `frames` square frames with the reference's own code."""
    side, seed, first, frames, lvl = args
    import ctypes as C
    import tempfile
    import numpy as np
    import oracle_py as O
    from khugle_amd.synth import synth_frame
    R = O.ref()
    path = os.path.join(tempfile.gettempdir(), "khg_cpu_%d.comp" % os.getpid()).encode()
    imgs = [synth_frame(side, side, seed, first + i) for i in range(frames)]
    t0 = time.perf_counter()
    te = td = 0.0
    for rgb in imgs:
        a = time.perf_counter()
        if R is not None:
            R.kref_encode_file(rgb[0], rgb[1], rgb[2], side, lvl, path)
        else:
            comp = O.encode(rgb, lvl, faithful=True)
        b = time.perf_counter()
        if R is not None:
            out = np.zeros_like(rgb)
            n = C.c_int(0)
            R.kref_decode_file(path, out[0], out[1], out[2], C.byref(n))
        else:
            O.decode(comp)
        c = time.perf_counter()
        te += b - a
        td += c - b
    try:
        os.unlink(path.decode())
    except OSError:
        pass
    return time.perf_counter() - t0, te, td


def cpu_baseline(side=5760, frames_per_core=1):
    import multiprocessing as mp
    import oracle_py as O
    # One process per core of this job's CPU share.  os.cpu_count() reports the whole machine; the GPU boxes give a
    # one-GPU job 16 CPUs and export that figure as OMP_NUM_THREADS, so the pool is capped there (CPU_POOL_CAP when
    # the variable is absent).
    host_cores = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or CPU_POOL_CAP
    cores = max(1, min(host_cores, share))
    O.build()
    kind = "reference" if O.have_ref() else "port"
    jobs = [(side, SEED, i * frames_per_core, frames_per_core, LVL) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    mp_total = cores * frames_per_core * side * side / 1e6
    per_core_enc = side * side / 1e6 / (sum(r[1] for r in res) / (cores * frames_per_core))
    per_core_dec = side * side / 1e6 / (sum(r[2] for r in res) / (cores * frames_per_core))
    what = ("reference DWT.h/Huffman.h/Writer.h/Reader.h compiled as they are + restated colour/quantiser"
            if kind == "reference" else "CPU restatement with the reference's O(N*K) table search")
    return {"value": round(mp_total / busy, 3), "unit": "MP/s encode+decode", "cores": cores, "kind": kind,
            "host_cores": host_cores,
            "cores_note": "pool = min(os.cpu_count() = %d, this job's CPU share = %d)" % (host_cores, share),
            "frames_timed": cores * frames_per_core,
            "sample": "%d frames of %dx%d (same generator, seed %d, High preset; the reference cannot "
                      "round-trip non-square frames), one process per core, %s; per core: encode %.2f MP/s, "
                      "decode %.2f MP/s; pool wall %.1f s" % (cores * frames_per_core, side, side, SEED, what,
                                                               per_core_enc, per_core_dec, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=128, help="frames per GPU per step")
    ap.add_argument("--streams", type=int, default=24, help="contexts/streams in flight per GPU")
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-square", action="store_true", help="skip the short 5760x5760 pass")
    ap.add_argument("--no-presets", action="store_true", help="skip the short Normal / Extreme passes")
    ap.add_argument("--no-format2", action="store_true", help="skip the short pass with the opt-in seek index")
    ap.add_argument("--no-host-pipeline", action="store_true", help="skip the short pass over frames in host memory")
    ap.add_argument("--no-isolated-kernel", action="store_true",
                    help="skip timing the dominant kernel alone on an idle GPU (20 launches after the timed region)")
    ap.add_argument("--hw-queues", type=int, default=0,
                    help="HIP hardware queues for this process (GPU_MAX_HW_QUEUES); 0 = one per stream, at most 16")
    args = ap.parse_args()
    # One process per GPU.  Under torch.distributed.run the ranks exist already; started plainly with --gpus N > 1
    # this process becomes the parent that starts N ranks (and never touches the GPU itself).
    from khugle_amd.launch import ranks_or_spawn
    rank, local_rank, world = ranks_or_spawn(args.gpus)
    # Streams that share a hardware queue run their kernels strictly one after another, single-workgroup
    # table/tree kernels included; with a queue per stream those hide behind other frames' wide kernels.
    # Read by the HIP runtime when it initialises, so it is set before torch is imported.
    os.environ["GPU_MAX_HW_QUEUES"] = str(max(1, min(16, args.hw_queues or args.streams)))

    import numpy as np
    import torch
    import torch.distributed as dist
    import khugle_amd
    from khugle_amd.shard import frame_range, max_over_ranks, whole_job_rate

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the codec has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus

    rows, cols, F, S = args.rows, args.cols, args.frames, max(1, min(args.streams, args.frames))
    px = rows * cols
    ctxs = [khugle_amd.Codec(rows, cols, device=local_rank) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    # inputs resident in HBM before the timed region: frames rank*F .. rank*F+F-1 of the synthetic batch
    frames = torch.empty((F, 3, rows, cols), dtype=torch.uint8, device=dev)
    for f, g in enumerate(frame_range(rank, world, F)):
        ctxs[0].synth_dev(rows, cols, SEED, g, out=frames[f])
    cap = (2 * px + (1 << 20)) & ~3  # 2 bytes per pixel: ~7x the High-preset size, checked by the encoder
    comps = torch.zeros((F, cap), dtype=torch.uint8, device=dev)
    decs = torch.empty((S, 3, rows, cols), dtype=torch.uint8, device=dev)
    lens = [0] * F
    torch.cuda.synchronize()

    def encode_pass():
        for f in range(F):
            c = f % S
            if f >= S:
                lens[f - S] = ctxs[c].encode_result()
            with torch.cuda.stream(streams[c]):
                ctxs[c].encode_dev(frames[f], LVL, comps[f])
        for f in range(max(0, F - S), F):
            lens[f] = ctxs[f % S].encode_result()

    def decode_pass():
        for f in range(F):
            c = f % S
            if f >= S and ctxs[c].last_status() != 0:
                raise RuntimeError("decode failed")
            with torch.cuda.stream(streams[c]):
                ctxs[c].decode_dev(comps[f], lens[f], rows, cols, decs[c])
        for c in range(S):
            if ctxs[c].last_status() != 0:
                raise RuntimeError("decode failed")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        encode_pass()
        decode_pass()
    for c in ctxs:
        c.profile_luma_forward(True)  # HIP events around the dominant kernel, on the stream it is launched on
    barrier()
    t_enc = t_dec = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a = time.perf_counter()
        encode_pass()
        torch.cuda.synchronize()
        b = time.perf_counter()
        decode_pass()
        torch.cuda.synchronize()
        c = time.perf_counter()
        t_enc += b - a
        t_dec += c - b
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, t_enc, t_dec = max_over_ranks([elapsed, t_enc, t_dec], dev)

    # dominant kernel: average duration of its launches inside the timed region (other streams' kernels share
    # the GPU with it there), from the events khg_encode_dev recorded around each launch
    prof = [c.profile_read() for c in ctxs]
    k_launches = sum(n for _, _, n in prof)
    k_event_ms = sum(ms for ms, _, _ in prof) / max(1, k_launches)
    k_ms = sum(dv for _, dv, _ in prof) / max(1, k_launches)
    k_alone_ms = None
    if not args.no_isolated_kernel:
        with torch.cuda.stream(streams[0]):
            k_alone_ms = ctxs[0].time_luma_forward_dev(frames[0], LVL, reps=20)

    # the same generator on a square 8K-class frame (5760 x 5760 = 33.2 MP): the shape the reference itself can
    # round-trip, so this rate is the one whose small siblings are byte-checked against the reference's files
    square = None
    if not args.no_square:
        side, SF, SS = 5760, 96, S
        sctx = [khugle_amd.Codec(side, side, device=local_rank) for _ in range(SS)]
        sfr = torch.empty((SF, 3, side, side), dtype=torch.uint8, device=dev)
        for f, g in enumerate(frame_range(rank, world, SF)):
            sctx[0].synth_dev(side, side, SEED, g, out=sfr[f])
        scomp = torch.zeros((SF, (2 * side * side + (1 << 20)) & ~3), dtype=torch.uint8, device=dev)
        sdec = torch.empty((SS, 3, side, side), dtype=torch.uint8, device=dev)
        slen = [0] * SF
        torch.cuda.synchronize()  # frames and buffers were filled on the default stream

        def square_pass():
            for f in range(SF):
                c = f % SS
                if f >= SS:
                    slen[f - SS] = sctx[c].encode_result()
                with torch.cuda.stream(streams[c]):
                    sctx[c].encode_dev(sfr[f], LVL, scomp[f])
            for f in range(max(0, SF - SS), SF):
                slen[f] = sctx[f % SS].encode_result()
            for f in range(SF):
                c = f % SS
                if f >= SS:
                    st_ = sctx[c].last_status()
                    if st_ != 0:
                        raise RuntimeError("decode of square frame %d failed: status %d" % (f - SS, st_))
                with torch.cuda.stream(streams[c]):
                    sctx[c].decode_dev(scomp[f], slen[f], side, side, sdec[c])
            for c in range(SS):
                st_ = sctx[c].last_status()
                if st_ != 0:
                    raise RuntimeError("decode of a square frame failed: status %d" % st_)

        square_pass()
        barrier()
        ts = time.perf_counter()
        square_pass()
        barrier()
        st, = max_over_ranks([time.perf_counter() - ts], dev)
        square = {"workload": "%d frames of %dx%d per GPU, same generator and preset, %d streams, one pass"
                              % (SF, side, side, SS),
                  "value": round(whole_job_rate(world, SF, 1, side * side, st), 1), "unit": "MP/s"}
        del sctx, sfr, scomp, sdec

    dec_last = decs[(F - 1) % S].clone()  # the timed decode's picture of the last frame, for the parity spot check
    # the other two presets of the reference (Compress.h:19-41), one pass each way over the first frames: Normal
    # quantises least, so its tables are the largest (the tree kernels' time grows with the table) and its files the
    # longest.  Never part of `value`: the headline configuration is High.
    presets = None
    if not args.no_presets:
        PF = min(F, 96)
        presets = {}
        for name, lvl in (("Normal", 0), ("Extreme", 2)):
            plen = [0] * PF

            def enc_p():
                for f in range(PF):
                    c = f % S
                    if f >= S:
                        plen[f - S] = ctxs[c].encode_result()
                    with torch.cuda.stream(streams[c]):
                        ctxs[c].encode_dev(frames[f], lvl, comps[f])
                for f in range(max(0, PF - S), PF):
                    plen[f] = ctxs[f % S].encode_result()

            def dec_p():
                for f in range(PF):
                    c = f % S
                    if f >= S and ctxs[c].last_status() != 0:
                        raise RuntimeError("decode failed (preset %s)" % name)
                    with torch.cuda.stream(streams[c]):
                        ctxs[c].decode_dev(comps[f], plen[f], rows, cols, decs[c])
                for c in range(min(S, PF)):
                    if ctxs[c].last_status() != 0:
                        raise RuntimeError("decode failed (preset %s)" % name)

            enc_p(); dec_p()
            barrier()
            pa = time.perf_counter()
            enc_p()
            torch.cuda.synchronize()
            pb = time.perf_counter()
            dec_p()
            torch.cuda.synchronize()
            pc = time.perf_counter()
            barrier()
            pe, pd = max_over_ranks([pb - pa, pc - pb], dev)
            hd = khugle_amd.peek(comps[0, :plen[0]].cpu().numpy().tobytes())
            presets[name] = {"value": round(whole_job_rate(world, PF, 1, px, pe + pd), 1), "unit": "MP/s",
                             "encode_mp_s": round(whole_job_rate(world, PF, 1, px, pe), 1),
                             "decode_mp_s": round(whole_job_rate(world, PF, 1, px, pd), 1),
                             "bits_per_pixel": round(8.0 * sum(plen) / (PF * px), 4),
                             "table_entries": [hd.rle_y_len, hd.rle_cb_len, hd.rle_cr_len],
                             "frames_per_gpu": PF}
        # the timed frames are re-encoded below (format 2 or not), so comps and lens stay those of the High preset
        encode_pass()

    # the same frames in the opt-in format 2 (the format-1 file followed by a seek index, csrc/khg_index.h): one pass
    # each way.  Never part of `value`: the headline is the reference's own format.
    format2 = None
    if not args.no_format2:
        lens2 = [0] * F

        def encode2_pass():
            for f in range(F):
                c = f % S
                if f >= S:
                    lens2[f - S] = ctxs[c].encode_result()
                with torch.cuda.stream(streams[c]):
                    ctxs[c].encode_dev(frames[f], LVL, comps[f])
            for f in range(max(0, F - S), F):
                lens2[f] = ctxs[f % S].encode_result()

        def decode2_pass():
            for f in range(F):
                c = f % S
                if f >= S and ctxs[c].last_status() != 0:
                    raise RuntimeError("indexed decode failed")
                with torch.cuda.stream(streams[c]):
                    ctxs[c].decode_indexed_dev(comps[f], lens2[f], rows, cols, decs[c])
            for c in range(S):
                if ctxs[c].last_status() != 0:
                    raise RuntimeError("indexed decode failed")

        for c in ctxs:
            c.set_format(2)
        encode2_pass(); decode2_pass()
        barrier()
        a2 = time.perf_counter()
        encode2_pass()
        torch.cuda.synchronize()
        b2 = time.perf_counter()
        decode2_pass()
        torch.cuda.synchronize()
        c2 = time.perf_counter()
        barrier()
        for c in ctxs:
            c.set_format(1)
        e2, d2 = max_over_ranks([b2 - a2, c2 - b2], dev)
        format2 = {"what": "the format-1 file followed by a seek index (opt-in, no counterpart in the reference); "
                           "one pass each way over the same frames and streams",
                   "encode_mp_s": round(whole_job_rate(world, F, 1, px, e2), 1),
                   "decode_mp_s": round(whole_job_rate(world, F, 1, px, d2), 1),
                   "bytes_per_frame": int(sum(lens2) / F), "format1_bytes_per_frame": int(sum(lens) / F),
                   "same_pixels_as_format1": bool(torch.equal(decs[(F - 1) % S], dec_last))}

    # frames and files in (pinned) host memory through the library's pipelined batch entry points: the rate a host
    # application sees, PCIe copies included.  Never part of `value`.
    host = None
    if not args.no_host_pipeline:
        from khugle_amd.batch import Batch
        HF, HD = 16, 8
        hb = Batch(rows, cols, depth=HD, device=local_rank)
        h_frames = torch.empty((HF, 3, rows, cols), dtype=torch.uint8).pin_memory()
        h_frames.copy_(frames[:HF])
        hcap = int(hb.L.khg_comp_bound(rows, cols))
        h_comps = torch.empty((HF, hcap), dtype=torch.uint8).pin_memory()
        h_dec = torch.empty((HF, 3, rows, cols), dtype=torch.uint8).pin_memory()
        fp = [h_frames[i].data_ptr() for i in range(HF)]
        cp = [h_comps[i].data_ptr() for i in range(HF)]
        dp = [h_dec[i].data_ptr() for i in range(HF)]
        hb.encode_ptrs(fp, rows, cols, LVL, cp, [hcap] * HF)  # warm-up
        barrier()
        ta = time.perf_counter()
        rc_e, hl, _ = hb.encode_ptrs(fp, rows, cols, LVL, cp, [hcap] * HF)
        tb = time.perf_counter()
        rc_d, _ = hb.decode_ptrs(cp, hl, dp)
        tc = time.perf_counter()
        if rc_e or rc_d:
            raise RuntimeError("host pipeline failed: %d %d" % (rc_e, rc_d))
        if hl[0] != lens[0] or not torch.equal(h_comps[0, :hl[0]], comps[0, :lens[0]].cpu()):
            raise RuntimeError("host pipeline bytes differ from the device-resident path")
        he, hd_ = max_over_ranks([tb - ta, tc - tb], dev)
        host = {"workload": "%d frames per GPU in pinned host memory, %d contexts in flight, one pass" % (HF, HD),
                "value": round(whole_job_rate(world, HF, 1, px, he + hd_), 1), "unit": "MP/s",
                "encode_mp_s": round(whole_job_rate(world, HF, 1, px, he), 1),
                "decode_mp_s": round(whole_job_rate(world, HF, 1, px, hd_), 1),
                "pcie_bytes_per_frame_each_way": 3 * px + int(sum(hl) / HF)}
        hb.close()
        del h_frames, h_comps, h_dec

    if rank == 0:
        # parity spot check (untimed): the last frame's bytes and pixels against the CPU oracle
        parity = None
        try:
            import oracle_py as O
            f = F - 1
            rgb = frames[f].cpu().numpy()
            comp = comps[f, :lens[f]].cpu().numpy().tobytes()
            want = O.encode(rgb, LVL)
            dec = dec_last.cpu().numpy()
            parity = bool(comp == want) and bool(np.array_equal(dec, O.decode(want)))
        except Exception as e:  # the oracle is test infrastructure: never let it break the measurement
            parity = "unavailable: %s" % e
        # HBM bytes per launch from the counters and the per-kernel table: collected by tools/gpu_roofline.sh
        # (rocprofv3 --kernel-trace --stats and one --pmc pass each for FETCH_SIZE and WRITE_SIZE), committed
        traffic = per_kernel = profile_head = valu = None
        stale = None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_stamp import csrc_digest
        running = csrc_digest()
        tpath = os.path.join(ROOT, "profiles", PROFILE_TAG + "_kernels.json")
        if os.path.exists(tpath):
            prof_table = json.load(open(tpath))
            traffic = prof_table["kernels"].get("khg_fwd_kernel<0>", {}).get("traffic_bytes")
            per_kernel = {k: {f: v[f] for f in ("avg_us", "algorithmic_bytes", "traffic_bytes", "frac") if f in v}
                          for k, v in prof_table["kernels"].items()}
            at = prof_table.get("taken_at", {})
            profile_head = at.get("git_head")
            stale = at.get("csrc_sha256") != running
        vpath = os.path.join(ROOT, "profiles", PROFILE_TAG + "_valu.json")
        if os.path.exists(vpath):
            v = json.load(open(vpath))
            valu = {"wave_insts": v["valu_wave_insts"],
                    "cycles_per_f64_wave_inst_per_simd": v["f64_probe"]["cycles_per_wave_inst_per_simd"],
                    "ns_per_f64_wave_inst_per_simd": v["f64_probe"]["ns_per_wave_inst_per_simd"],
                    "shader_ghz_held_in_probe": v["f64_probe"]["shader_ghz_held"],
                    "shader_ghz_held_in_kernel": v["shader_ghz_held_in_kernel"],
                    "ceiling_us": v["ceiling_us"], "profiled_launch_us": v["launch_us"], "frac": v["frac"],
                    "gap_to_1p67_ns": v["gap"],
                    "profile_head": v.get("taken_at", {}).get("git_head"),
                    "stale": v.get("taken_at", {}).get("csrc_sha256") != running,
                    "source": "profiles/%s_valu.json (tools/gpu_valu.sh: tools/probe_rates.hip with s_memtime against "
                              "s_memrealtime, a KHG_TIMING build for the clock inside the launch, SQ_INSTS_VALU from "
                              "a --pmc pass)" % PROFILE_TAG}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        l1_bytes = L1_FWD_BYTES_PER_PX * px

        def priced(ms):
            gbs = l1_bytes / (ms * 1e-3) / 1e9
            return {"avg_launch_ms": round(ms, 4), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}

        # `achieved` / `frac`: the kernel's own duration, i.e. launches on an otherwise idle GPU timed with HIP events
        # on the stream they run on (what rocprofv3 reports for the one-stream profile committed under profiles/).
        # `in_timed_region`: the same kernel's launches inside the timed region, where 24 streams share the GPU and a
        # launch's span (first workgroup start to last workgroup end, device clock) also covers the time its
        # workgroups wait for CUs held by other frames' kernels -- a lower bound on the kernel's rate, not its cost.
        main = priced(k_alone_ms) if k_alone_ms is not None else priced(k_ms)
        roofline = {"bound": "hbm",
                    "kernel": "khg_fwd_kernel<0, 2, 1, false> (colour + DWT level 1 + quantiser + histogram of Y, Cb, "
                              "Cr; khg_fwd_kernel<0> in the tables)",
                    "achieved": main["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": main["frac"],
                    "traffic": traffic, "bytes_per_launch": l1_bytes, "avg_launch_ms": main["avg_launch_ms"],
                    "timing": ("20 launches on an idle GPU after the timed region, HIP events on the launch stream"
                               if k_alone_ms is not None else "launches inside the timed region (device clock span)"),
                    "in_timed_region": dict(priced(k_ms), launches_timed=k_launches,
                                            event_avg_launch_ms=round(k_event_ms, 4),
                                            note="span of the launches that ran inside the timed region next to the "
                                                 "other streams' kernels; event_avg_launch_ms also counts the wait "
                                                 "for the GPU"),
                    "staged_model": {"bytes_per_px": L1_FWD_STAGED_BYTES_PER_PX,
                                     "note": "what the stages fused into this launch would move as separate passes "
                                             "with f64 planes between them; the launch is priced at its compulsory "
                                             "8.25 B/px"},
                    "valu": valu,
                    "per_kernel_profile": per_kernel,
                    "per_kernel_profile_head": profile_head,
                    "stale": stale,
                    "per_kernel_profile_note": "one-stream 8K frame under rocprofv3 (profiles/%s_kernels.json, made by "
                                               "tools/gpu_roofline.sh): avg_us, algorithmic and counter bytes per "
                                               "launch, fraction of the 8 TB/s HBM peak; stale = the library's "
                                               "sources have changed since (digest over csrc/ and the public "
                                               "header)" % PROFILE_TAG}
        total_mp = world * F * args.steps * px / 1e6
        bpp = 8.0 * sum(lens) / (F * px)
        line = {
            "metric": "encode+decode megapixels/sec (node), 8K batch, High preset",
            "value": round(whole_job_rate(world, F, args.steps, px, elapsed), 1),
            "unit": "MP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "synthetic %dx%d RGB, %d frames per GPU per step (1024-frame batch sharded "
                                   "over 8 GPUs), High preset, encode to .comp then decode to RGB, device resident"
                                   % (cols, rows, F),
                       "frames_per_gpu": F, "streams_per_gpu": S, "seed": SEED, "bits_per_pixel": round(bpp, 4)},
            "encode_mp_s": round(total_mp / t_enc, 1),
            "decode_mp_s": round(total_mp / t_dec, 1),
            "parity_spot_check": parity,
            "roofline": roofline,
            "cpu_baseline": cpu,
            # the pass-structured traffic model of the path (63 + c bytes per pixel each way, c = .comp bytes
            # per pixel): what a stage-per-kernel pipeline would move; fusion keeps the real traffic below it
            "pipeline_model": {"bytes_per_px_each_way": round(63 + bpp / 8, 3),
                               "encode_gbps": round((63 + bpp / 8) * total_mp / t_enc / 1e3, 1),
                               "decode_gbps": round((63 + bpp / 8) * total_mp / t_dec / 1e3, 1),
                               "peak_gbps": HBM_PEAK_GBS * world},
            "presets": presets,
            "square_variant": square,
            "format2": format2,
            "host_pipeline": host,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
