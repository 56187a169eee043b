"""Rank body for tests/test_launch.py: joins a gloo group through the environment the launcher set up, reduces
one number and lets rank 0 print what a benchmark's JSON line would carry."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))


def main():
    n = int(sys.argv[1])
    from khugle_amd.launch import ranks_or_spawn
    rank, local_rank, world = ranks_or_spawn(n, need_devices=False)
    if len(sys.argv) > 2 and sys.argv[2] in ("fail", "stubborn") and rank == 1:
        if sys.argv[2] == "stubborn":
            import time
            time.sleep(1.0)  # rank 0 has its handler in place by now
        sys.exit(7)
    if len(sys.argv) > 2 and sys.argv[2] == "stubborn":  # a rank that a collective or a GPU wait keeps from ending
        import signal
        import time
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(600)
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == n
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "max": float(t[0]), "local_rank": local_rank}), flush=True)


main()
