import sys, heapq, numpy as np, torch
sys.path.insert(0, "/root/repo")
from telomeri_amd import capi, synth
ctx = capi.Context(0)
ds = synth.make_config("athaliana", device="cuda")
g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
wp = capi.WalkParams.default()
ps = g.walk_mc(0, wp, 0, 50000)
st = ps.unit_stats()
steps = st["steps"].astype(np.float64)
print("units", len(steps), "mean", steps.mean(), "max", steps.max(), "p99", np.percentile(steps, 99), "kernel_ms", ps.counters.kernel_ms, "waves", ps.counters.waves)
for W in (8192, 4096):
    h = [0.0]*W; heapq.heapify(h)
    for s in steps:
        t = heapq.heappop(h); heapq.heappush(h, t+s)
    mk = max(h); print("workers", W, "makespan", mk, "ideal", steps.sum()/W, "ratio", mk/(steps.sum()/W))
# both metrics in one queue
ps2 = g.walk_mc(1, wp, 0, 50000); st2 = ps2.unit_stats()["steps"].astype(np.float64)
allsteps = np.concatenate([steps, st2])
h = [0.0]*8192; heapq.heapify(h)
for s in allsteps:
    t = heapq.heappop(h); heapq.heappush(h, t+s)
print("merged makespan ratio", max(h)/(allsteps.sum()/8192))
