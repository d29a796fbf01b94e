"""The N>1 step sequence (hipGraph replay -> RCCL all-reduce -> Adam outside the graph; rank loss weights; the two-graph
overlapped exchange and its eager fallback) on ONE GPU through a 1-rank RCCL group -- tests/scripts/single_gpu_distributed.py."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_distributed_step_equals_single_rank_step_bit_for_bit():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scripts", "single_gpu_distributed.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
