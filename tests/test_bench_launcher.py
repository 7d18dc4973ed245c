"""bench.py's own N-rank launcher (`python bench.py --gpus N`, the driver's command line), on CPU.

The ranks of the real benchmark need a GPU, so here (a) a stand-in rank program exercises the launcher's
environment, rendezvous (gloo, world_size 2) and JSON relay, (b) the real bench.py is started with --gpus 2 on
this GPU-less box and must fail loudly -- never fall back to one rank or to a CPU path.  The real two-rank
run (the HIP shooter under both ranks) is tests/test_gpu_bench.py, marked gpu."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stubs", "rank_stub.py")


def _launch(args, script=STUB):
    code = ("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.spawn_ranks(%d, %r, script=%r))"
            % (ROOT, args[0], args[1], script))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_launcher_starts_n_ranks_and_relays_rank0_json():
    r = _launch((2, []))
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1 and r.stdout.strip() == line[0]      # stdout is the JSON line and nothing else
    assert "noise before the line" in r.stderr
    j = json.loads(line[0])
    assert j == {"n_gpus": 2, "sum": 3, "tens": 20, "local_rank": "0", "master": "127.0.0.1"}


def test_launcher_fails_when_any_rank_fails():
    r = _launch((2, ["--fail-rank", "1"]))
    assert r.returncode != 0
    assert "rank 1 exited with code 7" in r.stderr


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_gpus2_without_a_gpu_fails_loudly_in_every_rank(gpu_available):
    if gpu_available:
        pytest.skip("GPU present: the real two-rank run is tests/test_gpu_bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "no CPU fallback" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
