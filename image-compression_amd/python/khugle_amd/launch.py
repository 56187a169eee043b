"""One process per GPU: the launcher behind `bench.py --gpus N` and `bench_tiled.py --gpus N`.

A benchmark script calls ranks_or_spawn(n) first thing.  Under an external launcher (torch.distributed.run sets
RANK / WORLD_SIZE) it only checks that the launcher's world size is the requested one.  Started plainly with
n > 1 it becomes the parent: it starts n copies of the script, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, waits for them and exits with the worst exit code.  The parent runs no GPU work.
(Counting the devices may initialise the HIP runtime in the parent -- torch falls back to hipGetDeviceCount where
it has no amdsmi -- which is harmless here: the ranks are fresh processes started with Popen, not forks that would
inherit the runtime's state, and the parent never replaces itself with another program.)
When a rank fails, the others are told to stop (SIGTERM); a rank that has not gone after KILL_AFTER seconds -- it may
sit in a collective or a GPU wait that ignores the signal -- is killed, so the parent always exits.
"""
import os
import socket
import subprocess
import sys
import time


# seconds between SIGTERM and SIGKILL for the ranks that outlive a failed one
KILL_AFTER = float(os.environ.get("KHG_LAUNCH_KILL_AFTER", "10"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def device_count():
    import torch
    return torch.cuda.device_count()


def ranks_or_spawn(n, need_devices=True, argv=None):
    """Returns (rank, local_rank, world) in a rank process.  In the parent of a self-launched job it does not
    return: it exits with the job's exit code."""
    if n < 1:
        raise SystemExit("--gpus must be at least 1")
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ:  # a launcher started us
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != n:
            raise SystemExit("--gpus %d, but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a "
                             "number for a job of another size" % (n, world))
        return rank, local_rank, world
    if n == 1:
        return 0, 0, 1
    if need_devices:
        have = device_count()
        if have < n:
            raise SystemExit("--gpus %d asked for, but this machine has %d HIP device(s): refusing to run fewer "
                             "ranks than requested" % (n, have))
    argv = list(sys.argv if argv is None else argv)
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    worst = 0
    alive = list(procs)
    kill_at = None
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and worst == 0:
                worst = rc if rc > 0 else 1
                for q in alive:  # one rank failed: the others would wait in a collective for ever
                    q.terminate()
                kill_at = time.monotonic() + KILL_AFTER
        if kill_at is not None and alive and time.monotonic() >= kill_at:
            for q in alive:
                q.kill()
            kill_at = None
    raise SystemExit(worst)
