"""telomeri_amd/csrc/cpu_share.h: the number of host threads worth starting = min(cores, affinity mask, cgroup CPU quota).

The GPU boxes show 256 cores and give a container 16 cores' worth of CPU time (cpu.max); thread counts taken from the core
count got the whole process tree throttled ("The CPU quota").  CPU-only test: a three-line program is compiled
against the header and its answer compared with the same rule restated here.
"""
import math
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def expected_share():
    n = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    quota = period = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota, period = int(q), int(p)
    except OSError:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except OSError:
            pass
    if quota and period and quota > 0 and period > 0:
        n = min(n, max(1, math.ceil(quota / period)))
    return max(1, n)


def test_cpu_share_follows_affinity_and_quota(tmp_path):
    src = tmp_path / "share.cpp"
    src.write_text('#include "cpu_share.h"\nint main() { std::printf("%u\\n", tlm_cpu_share()); return 0; }\n')
    exe = tmp_path / "share"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(REPO, "telomeri_amd", "csrc"), str(src), "-o", str(exe),
                    "-pthread"], check=True)
    base_env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "TLM_WORLD")}
    got = int(subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=base_env).stdout)
    assert got == expected_share()
    assert 1 <= got <= (os.cpu_count() or 1)
    # the ranks of a node share it
    if got >= 2:
        env = dict(os.environ, LOCAL_WORLD_SIZE="2")
        assert int(subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=env).stdout) == got // 2
        env = dict(os.environ, TLM_WORLD=str(4 * got))
        env.pop("LOCAL_WORLD_SIZE", None)
        assert int(subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=env).stdout) == 1
    # a narrower affinity mask narrows the answer
    one = sorted(os.sched_getaffinity(0))[:1]
    got1 = int(subprocess.run(["taskset", "-c", str(one[0]), str(exe)], check=True, capture_output=True, text=True).stdout)
    assert got1 == 1


def test_python_cpu_share_follows_the_same_rule(monkeypatch):
    from telomeri_amd import cpu_share
    for var in ("LOCAL_WORLD_SIZE", "TLM_WORLD"):
        monkeypatch.delenv(var, raising=False)
    assert cpu_share.cpu_share() == expected_share()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    assert cpu_share.cpu_share() == max(1, expected_share() // 2)
    assert isinstance(cpu_share.throttle_counters(), dict)
