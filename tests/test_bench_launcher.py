"""CPU tier: `bench.py --gpus N` without a launcher around it starts its own N ranks (fresh children; the parent never
touches a GPU), hands each the rendezvous variables torch.distributed.run would set, relays rank 0's stdout, and ends the
job with a non-zero status as soon as one rank fails."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TLM_BENCH_LAUNCH_ECHO="1", **kw)
    return env


def test_bench_starts_its_own_ranks():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True,
                         timeout=120, env=_env())
    assert res.returncode == 0, res.stderr
    lines = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                              # rank 0's line only
    assert lines[0]["RANK"] == "0" and lines[0]["LOCAL_RANK"] == "0" and lines[0]["WORLD_SIZE"] == "3"
    assert lines[0]["MASTER_ADDR"] == "127.0.0.1" and int(lines[0]["MASTER_PORT"]) > 0


def test_bench_under_a_launcher_does_not_spawn():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, env=_env(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"))
    assert res.returncode == 0, res.stderr
    assert json.loads(res.stdout.strip())["RANK"] == "1"


def test_bench_ends_the_job_when_a_rank_fails():
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, env=_env(TLM_BENCH_LAUNCH_ECHO_FAIL="1"))
    assert res.returncode == 7
    assert time.time() - t0 < 25                        # rank 0 would have slept 30 s
    assert "rank 1 ended with status 7" in res.stderr
