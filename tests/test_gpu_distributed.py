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


def test_bench_spawns_its_own_rank_and_prints_a_valid_line():
    """`bench.py --gpus 1 --spawn`: the parent starts one child (1-rank RCCL group, all-reduce + Adam outside the graph) and relays
    its JSON line; the line carries what an N > 1 run reports (all-reduce timing, overlap on, RCCL ranks)."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-roofline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["value"] > 0 and line["unit"] == "molecules/s"
    assert "spawned" in line["config"] and "allreduce" in line["config"]["step"]
    assert line["allreduce_alone"]["us_per_call"] > 0 and line["nccl_ranks"]["world_size"] == 1
    assert line["overlap_on"]["ms_per_step"] > 0


def test_bench_two_ranks_report_strong_headline_and_weak_subobject():
    """The N > 1 line of bench.py with two REAL ranks (spawned by the parent) -- on one GPU, so over gloo instead of RCCL
    (FRAGNET_BENCH_BACKEND; RCCL refuses two ranks per device): headline = the global batch of 512 sharded over the ranks
    (strong), `weak` sub-object = 512 molecules per rank, the flat all-reduce timed alone, per-rank step times."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", FRAGNET_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--overlap", "off",
                        "--no-cpu-baseline", "--no-roofline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 3
    assert line["config"]["global_batch"] == 512 and line["config"]["per_gpu_batch"] == 256
    assert line["weak"]["scaling"] == "weak" and line["weak"]["global_batch"] == 1024 and line["weak"]["value"] > 0
    assert line["allreduce_alone"]["us_per_call"] > 0 and line["nccl_ranks"]["world_size"] == 2
    assert line["ms_per_step_fastest_rank"] <= line["ms_per_step"] + 1e-9
