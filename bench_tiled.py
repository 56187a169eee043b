#!/usr/bin/env python3
"""One large synthetic image tiled by rows over the GPUs of a node (BASELINE.json config 4), one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench_tiled.py --side 32768 --steps 3 --warmup 1

Each rank generates its tile of the frame (plus 24 halo rows per side) on its own GPU, so the timed region starts
with the inputs resident in HBM.  A step = tiled encode (DWT halos recomputed from the halo rows, histogram
all-reduced over RCCL, chunk bit counts all-reduced, shares of the file reduced to rank 0) followed by tiled decode
(every rank entropy-decodes 1/N of each plane's stream, the workgroup records are all-reduced, the decoded symbols go
point to point to the ranks whose rows need them, every rank reconstructs its rows).  Prints one JSON line on rank 0.

With one process (plain `python bench_tiled.py --side 8192`) it measures the degenerate single-tile case.
`--virtual N` rehearses N ranks as N contexts on ONE GPU (threads, collectives inside the process): no scaling
figure, but the whole tiled path runs at the full image size and is checked against the one-tile result.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

LVL, SEED = 1, 2025


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--virtual", type=int, default=0, help="rehearse this many ranks on one GPU (threads)")
    ap.add_argument("--merge", choices=("reduce", "gather"), default="reduce",
                    help="how the ranks' shares become the file (khugle_amd.tiled.encode_tiled)")
    ap.add_argument("--host", choices=("native", "python"), default="native",
                    help="native: the library's C++ host calls RCCL itself (khg_tile_encode / khg_tile_decode); "
                         "python: the same schedule from khugle_amd.tiled over torch.distributed")
    ap.add_argument("--gpus", type=int, default=0,
                    help="ranks (one process per GPU); started plainly, this process launches them itself")
    args = ap.parse_args()
    from khugle_amd.launch import ranks_or_spawn
    rank, local_rank, world = ranks_or_spawn(args.gpus or int(os.environ.get("WORLD_SIZE", "1")))

    import torch
    import torch.distributed as dist
    from khugle_amd import Codec, tiled

    if not torch.cuda.is_available():
        raise SystemExit("bench_tiled.py needs a HIP device (the codec has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    side = args.side
    px = side * side

    def run_rank(r, w, coll, codec, comm=None):
        """Generate the tile, then warm-up + timed steps; returns (seconds, comp bytes tensor or None, tile)."""
        i0, i1 = tiled.input_rows(side, r, w)
        strip = codec.synth_rows_dev(i0, i1 - i0, side, SEED, 0)
        be = tiled.HipTileBackend(codec)
        if comm is not None:
            encode = lambda: tiled.encode_tiled_native(codec, comm, strip, side, side, LVL, merge=args.merge)
            decode = lambda full, n: tiled.decode_tiled_native(codec, comm, full, n, side, side)
        else:
            encode = lambda: tiled.encode_tiled(be, strip, side, side, LVL, r, w, coll=coll, merge=args.merge)
            decode = lambda full, n: tiled.decode_tiled(be, full, n, side, side, r, w, coll=coll)
        torch.cuda.synchronize()
        comp = tile = None
        length = torch.zeros(1, dtype=torch.int64, device=dev)
        t_enc = t_dec = 0.0
        for it in range(args.warmup + args.steps):
            if coll is not None and hasattr(coll, "barrier"):
                coll.barrier()
            elif w > 1:
                dist.barrier()
            torch.cuda.synchronize()
            a = time.perf_counter()
            comp = encode()
            torch.cuda.synchronize()
            b = time.perf_counter()
            # the file goes back to every rank (decode needs the whole stream)
            if r == 0:
                length[0] = comp.numel()
            if coll is not None and hasattr(coll, "broadcast"):
                full, n = coll.broadcast(comp, length, r)
            elif w > 1:
                dist.broadcast(length, src=0)
                n = int(length.item())
                full = torch.zeros((n + 11) & ~3, dtype=torch.uint8, device=dev)
                if r == 0:
                    full[:n] = comp
                dist.broadcast(full, src=0)
            else:
                n = comp.numel()
                full = torch.zeros((n + 11) & ~3, dtype=torch.uint8, device=dev)
                full[:n] = comp
            torch.cuda.synchronize()
            c = time.perf_counter()
            tile = decode(full, n)
            torch.cuda.synchronize()
            d = time.perf_counter()
            if it >= args.warmup:
                t_enc += b - a
                t_dec += (d - c)
        return t_enc, t_dec, comp, tile

    if args.virtual > 1:
        from tiledlib import ThreadCollectives, run_ranks
        import threading
        w = args.virtual
        ctxs = [Codec(side, side, device=local_rank) for _ in range(w)]
        bar = threading.Barrier(w)
        box = {}

        class Coll(ThreadCollectives):
            def barrier(self):
                bar.wait()

            def broadcast(self, comp, length, r):
                if r == 0:
                    n = comp.numel()
                    full = torch.zeros((n + 11) & ~3, dtype=torch.uint8, device=dev)
                    full[:n] = comp
                    box["full"], box["n"] = full, n
                bar.wait()
                out = (box["full"], box["n"])
                bar.wait()
                return out

        comms = [None] * w
        if args.host == "native":
            group = tiled.Comm.local_group(w)
            comms = [tiled.Comm.local(group, r, w) for r in range(w)]
        res = run_ranks(w, lambda r, coll: run_rank(r, w, Coll(coll.sh, r), ctxs[r], comms[r]))
        t_enc = max(x[0] for x in res)
        t_dec = max(x[1] for x in res)
        comp = res[0][2]
        # check against the single-tile path
        one = Codec(side, side, device=local_rank)
        frame = one.synth_rows_dev(0, side, side, SEED, 0)
        out = torch.zeros(one.comp_bound(side, side), dtype=torch.uint8, device=dev)
        one.encode_dev(frame, LVL, out)
        n1 = one.encode_result()
        dec = torch.empty((3, side, side), dtype=torch.uint8, device=dev)
        one.decode_dev(out, n1, side, side, dec)
        ok = one.last_status() == 0 and n1 == comp.numel() and bool(torch.equal(out[:n1], comp))
        for r in range(w):
            r0, r1 = tiled.tile_rows(side, r, w)
            ok = ok and bool(torch.equal(res[r][3], dec[:, r0:r1]))
        mode, nranks = "virtual ranks on one GPU (threads)", w
        comp_len = comp.numel()
    else:
        codec = Codec(side, side, device=local_rank)
        comm = tiled.Comm.rccl(rank, world, local_rank) if args.host == "native" else None
        t_enc, t_dec, comp, tile = run_rank(rank, world, None, codec, comm)
        tt = torch.tensor([t_enc, t_dec], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_enc, t_dec = [float(x) for x in tt.cpu()]
        ok = None
        mode, nranks = "one process per GPU, RCCL", world
        comp_len = comp.numel() if rank == 0 else 0

    if rank == 0:
        mp = px * args.steps / 1e6
        print(json.dumps({
            "metric": "encode+decode megapixels/sec, one %dx%d image tiled by rows, High preset" % (side, side),
            "value": round(mp / (t_enc + t_dec), 1), "unit": "MP/s", "n_gpus": world, "tiles": nranks, "mode": mode,
            "steps": args.steps, "warmup": args.warmup, "encode_mp_s": round(mp / t_enc, 1),
            "decode_mp_s": round(mp / t_dec, 1), "comp_bytes": comp_len,
            "bits_per_pixel": round(8.0 * comp_len / px, 4), "identical_to_single_tile": ok, "merge": args.merge,
            "host": args.host, "scaling": "strong", "data": "synthetic", "dtype": "f64"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
