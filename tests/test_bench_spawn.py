"""bench.py --gpus N without a launcher: the parent starts N child processes (never exec), hands each the torchrun environment
contract, relays rank 0's JSON line as its own last stdout line and exits non-zero when any rank fails.  The children here are
stub commands (no GPU in the CPU suite); on the GPU box `bench.py --gpus 1 --spawn` runs the real thing (tests/test_gpu_train.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OK_CHILD = ("import os, json, sys; r = int(os.environ['RANK']); "
            "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0 and os.environ['LOCAL_RANK'] == str(r); "
            "print('noise from rank', r); "
            "print(json.dumps({'rank': r, 'world': int(os.environ['WORLD_SIZE']), 'child': os.environ.get('FRAGNET_BENCH_CHILD')}))")
BAD_CHILD = ("import os, sys, time, json; r = int(os.environ['RANK']); "
             "print(json.dumps({'rank': r})); sys.stdout.flush(); "
             "sys.exit(7) if r == 1 else time.sleep(60)")


def _run(n, child, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--child-cmd", json.dumps([sys.executable, "-c", child]), *extra]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)


def test_parent_spawns_ranks_and_relays_rank0_json_last():
    r = _run(3, OK_CHILD)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    last = json.loads(lines[-1])
    assert last == {"rank": 0, "world": 3, "child": "1"}
    assert "noise from rank 0" in r.stdout                 # rank 0's other output is relayed, in front of the JSON line
    assert "noise from rank 1" not in r.stdout             # the other ranks' stdout goes to stderr
    assert "noise from rank 2" in r.stderr


def test_parent_fails_when_a_rank_fails_and_stops_the_others():
    r = _run(2, BAD_CHILD)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with 7" in r.stderr


def test_single_gpu_spawn_flag_starts_one_child():
    r = _run(1, OK_CHILD, extra=("--spawn",))
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.splitlines()[-1]) == {"rank": 0, "world": 1, "child": "1"}


def test_spawn_function_directly():
    sys.path.insert(0, ROOT)
    import bench
    rc = bench.spawn_ranks(2, [], child_cmd=[sys.executable, "-c", "import os, sys; sys.exit(0 if os.environ['WORLD_SIZE'] == '2' else 3)"])
    assert rc == 1                                          # both ranks exit 0 but rank 0 printed no JSON line
