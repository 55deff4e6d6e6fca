#!/bin/bash
# profiles/probes/cpu_quota.sh -- what CPU time the box gives us: affinity, cgroup quota, and whether a run of the host program was throttled
show() { for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpu/cpu.stat /sys/fs/cgroup/cpu,cpuacct/cpu.stat; do [ -r $f ] && { echo "== $f"; cat $f; }; done; }
echo "nproc $(nproc), online $(getconf _NPROCESSORS_ONLN)"; cat /proc/self/cgroup; show
bash $GRAFT_REPO_ROOT/profiles/probes/child_runs.sh 1 TLM_HOST_TIMING=1
show
