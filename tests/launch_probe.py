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
    if len(sys.argv) > 2 and sys.argv[2] == "fail" and rank == 1:
        sys.exit(7)
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
