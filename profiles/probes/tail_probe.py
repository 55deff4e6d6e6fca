"""How much of the Monte-Carlo launch is tail?  Per-walk step counts of the bench workload from the device, then a
greedy-queue simulation on 8192 workers: the present order (anchor-major), the anchors sorted by their mean walk length
(longest first, as a pilot of a few trials per anchor could estimate it), and the bound sum / workers."""
import sys, os, heapq, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from telomeri_amd import capi, synth
ctx = capi.Context(0)
ds = synth.make_config("athaliana", device="cuda")
g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
wp = capi.WalkParams.default()
a, b = g.walk_mc2(0, 1, wp, 0, 50000)
steps = np.concatenate([a.unit_stats()["steps"], b.unit_stats()["steps"]]).astype(np.float64)
print("walks", len(steps), "mean", steps.mean(), "p99", np.percentile(steps, 99), "max", steps.max(), "kernel_ms", a.counters.kernel_ms)
per_anchor = steps.reshape(50, 2000)          # 25 anchors x 2 metrics
means = per_anchor.mean(axis=1)
print("per (metric, anchor) mean steps: min %.0f max %.0f" % (means.min(), means.max()))

def makespan(order, W=8192):
    h = [0.0] * W
    heapq.heapify(h)
    for s in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + s)
    return max(h)

ideal = steps.sum() / 8192
print("ideal", ideal, "longest walk / ideal", steps.max() / ideal)
print("anchor-major order      : makespan / ideal = %.3f" % (makespan(steps) / ideal))
lpt = np.concatenate([per_anchor[i] for i in np.argsort(-means)])
print("anchors by mean, longest first: %.3f" % (makespan(lpt) / ideal))
pil = per_anchor[:, :16].mean(axis=1)          # what a 16-trial pilot would see
lpt2 = np.concatenate([per_anchor[i] for i in np.argsort(-pil)])
print("anchors by 16-trial pilot mean: %.3f" % (makespan(lpt2) / ideal))
print("walks sorted, longest first (hindsight): %.3f" % (makespan(np.sort(steps)[::-1]) / ideal))
