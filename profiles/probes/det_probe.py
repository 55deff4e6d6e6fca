"""DET units of the bench workload: how long is the longest walk, and is the launch bound by it?"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from telomeri_amd import capi, synth
ctx = capi.Context(0)
ds = synth.make_config("athaliana", device="cuda")
g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
anchors, det_off = g.anchor_table()
u1 = int(det_off[25])
for rep in range(2):
    pe, po = g.walk_det2(0, 1, 0, u1)
for ps in (pe, po):
    st = ps.unit_stats()
    steps = st["steps"].astype(np.float64)
    ev = st["evals"].astype(np.float64)
    print("units", len(steps), "steps: sum", steps.sum(), "mean", steps.mean(), "max", steps.max(), "p99", np.percentile(steps, 99),
          "evals", ev.sum(), "kernel_ms", ps.counters.kernel_ms, "waves", ps.counters.waves)
    if ps.counters.kernel_ms:
        print("  longest walk alone at this launch time:", ps.counters.kernel_ms * 1e3 / steps.max(), "us per step")
for wpc in (24, 16, 12, 8, 4):
    ctx.set_walk_tuning(waves_per_cu=wpc, path_capacity=0)
    for rep in range(2):
        pe, po = g.walk_det2(0, 1, 0, u1)
    print("waves_per_cu", wpc, "kernel_ms", pe.counters.kernel_ms, "waves", pe.counters.waves)
