"""Batch sharding over ranks (one process per GPU): frames are independent, so the data path has no collective.

The only communication is the barrier around the timed region and the max-over-ranks of the timings; both go
through torch.distributed (RCCL on GPUs, gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def frame_range(rank, world, frames_per_rank):
    """Global indices of the frames rank `rank` owns: a contiguous block, weak scaling (fixed work per rank)."""
    if not (0 <= rank < world) or frames_per_rank < 0:
        raise ValueError("bad shard (%d of %d, %d frames)" % (rank, world, frames_per_rank))
    return range(rank * frames_per_rank, (rank + 1) * frames_per_rank)


def max_over_ranks(values, device):
    """Element-wise maximum of a list of floats over all ranks (identity when not distributed)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.cpu()]


def whole_job_rate(world, frames_per_rank, steps, pixels_per_frame, elapsed_s):
    """Megapixels per second of the whole job: every rank's frames over the slowest rank's time."""
    return world * frames_per_rank * steps * pixels_per_frame / 1e6 / elapsed_s
