"""bench.py started bare with --gpus N must start its own ranks (the driver launches N > 1 either way), relay exactly one JSON
line when they produce one, and hand their failure on when they do not.  No GPU here: the ranks refuse to run ("needs a HIP
device"), which is the failure path -- non-zero exit, nothing on stdout that could be mistaken for a result."""
import os
import subprocess
import sys

from conftest import ROOT


def test_bare_multi_gpu_start_spawns_ranks_and_hands_their_failure_on():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("a GPU is present: the ranks would run the real benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == "", r.stdout[-500:]
    assert "needs a HIP device" in r.stderr  # printed by the RANKS: they were started, through torch.distributed.run
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "FAILED" in r.stderr.upper()


def test_single_process_start_needs_no_launcher():
    """--gpus 1 (the default) never spawns: without a GPU it fails in this very process with the same message"""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs a HIP device" in r.stderr
