"""bench.py's rank handling, on a host without a GPU: it must never print a line whose n_gpus differs from
--gpus, and --gpus N without a launcher must start N fresh rank processes."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason="exercises the no-GPU exits of bench.py")


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing to print" in r.stderr and r.stdout.strip() == ""


def test_gpus_without_a_launcher_starts_that_many_ranks():
    r = _run(["--gpus", "3", "--steps", "1"])
    # every rank gets as far as the device check and fails loudly there: there is no CPU fallback
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") == 3 and r.stdout.strip() == ""
