// telomeri_amd/csrc/cpu_share.h -- how many threads are worth starting (host side of the library and of the host program).
//
// Not the machine's core count: a container sees all 256 cores of an MI355X host but is given a CPU-time QUOTA (cgroup
// cpu.max; 16 cores' worth on the boxes this was measured on), and a process tree that runs more busy threads than that is
// stopped as a whole until the next 100 ms period -- the thread that feeds the device included.  Measured: with 32-64
// workers scanning the sequence files in the helper process, the device graph build of the other process took 43 to 115 ms
// (nr_throttled / throttled_usec of cpu.stat went up with every run).
#ifndef TLM_CPU_SHARE_H
#define TLM_CPU_SHARE_H

#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <thread>

inline unsigned tlm_cpu_share() {
    static const unsigned share = [] {
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned) CPU_COUNT(&set));
        long long quota = -1, period = 0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
            char q[32] = "";
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = std::atoll(q);
            std::fclose(f);
        } else {                                                                        // cgroup v1
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
                std::fclose(g);
            }
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(g, "%lld", &period) != 1) period = 0;
                std::fclose(g);
            }
        }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned) std::max<long long>(1, (quota + period - 1) / period));
        // one process per GPU: the ranks of a node share all of the above (LOCAL_WORLD_SIZE is set by torch.distributed.run,
        // TLM_WORLD by the host program's own launcher; both mean ranks of ONE node here)
        for (const char *var : {"LOCAL_WORLD_SIZE", "TLM_WORLD"})
            if (const char *v = std::getenv(var)) {
                const long ranks = std::atol(v);
                if (ranks > 1) n = std::max(1u, n / (unsigned) ranks);
                break;
            }
        return std::max(1u, n);
    }();
    return share;
}

#endif
