"""`--gpus N` starts N ranks (khugle_amd.launch): the self-launch path at world size 2 over gloo, the check against
an external launcher's world size, and bench.py refusing to run fewer ranks than it was asked for."""
import json
import os
import subprocess
import sys

from conftest import ROOT

PROBE = os.path.join(ROOT, "tests", "launch_probe.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_self_launch_runs_two_ranks():
    p = subprocess.run([sys.executable, PROBE, "2"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line == {"n_gpus": 2, "max": 2.0, "local_rank": 0}


def test_one_failing_rank_fails_the_job():
    p = subprocess.run([sys.executable, PROBE, "2", "fail"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_a_rank_that_ignores_sigterm_is_killed():
    import time
    env = dict(_clean_env(), KHG_LAUNCH_KILL_AFTER="1")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, PROBE, "2", "stubborn"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert time.monotonic() - t0 < 60


def test_world_size_of_an_external_launcher_must_match():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, PROBE, "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "refusing" in p.stderr


def test_bench_refuses_more_ranks_than_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this machine has two devices")
    for script in ("bench.py", "bench_tiled.py"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2"], env=_clean_env(),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and "refusing to run fewer ranks" in p.stderr, (script, p.stderr)
