"""The CPU share of this process: min(cores, affinity mask, cgroup CPU quota) -- csrc/cpu_share.h for Python callers.

A GPU box shows all 256 cores of its host and gives a container 16 cores' worth of CPU time (cgroup cpu.max).  Thread pools
sized from the core count (OpenMP inside numpy / torch: 256 workers that spin for a while after every parallel region) use
the quota up and the WHOLE process tree is stopped until the next 100 ms period -- the thread that feeds the device
included, wherever it happens to be (DESIGN.md §6 "The CPU quota").  `limit_thread_pools()` must run before numpy / torch
are imported."""
import math
import os


def cpu_share() -> int:
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = period = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota, period = int(q), int(p)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota and period and quota > 0 and period > 0:
        n = min(n, max(1, math.ceil(quota / period)))
    for var in ("LOCAL_WORLD_SIZE", "TLM_WORLD"):       # one process per GPU: the ranks of a node share all of the above
        v = os.environ.get(var)
        if v:
            try:
                if int(v) > 1:
                    n = max(1, n // int(v))
            except ValueError:
                pass
            break
    return max(1, n)


def limit_thread_pools() -> int:
    """Cap the OpenMP / BLAS pools of libraries imported AFTER this call at the CPU share; returns the share."""
    n = cpu_share()
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(var, str(n))
    return n


def throttle_counters() -> dict:
    """nr_throttled / throttled_usec of this cgroup (cpu.stat): how often and for how long the quota stopped the tree."""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for ln in open(path):
                k, _, v = ln.partition(" ")
                if k in ("nr_throttled", "throttled_usec", "throttled_time", "nr_periods"):
                    out[k] = int(v)
            break
        except (OSError, ValueError):
            continue
    return out
