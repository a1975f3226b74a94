"""The sharded decode step (HeadParallelDecoder.step, what `bench.py --gpus N` runs) with world_size 2 on ONE GPU:
both ranks on cuda:0, the all-gather carried by gloo through the host (RCCL refuses two ranks on one device).
Sharded output == unsharded output."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_match_unsharded():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hp_two_ranks_one_gpu.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "max|diff|" in r.stdout
