"""DET units of a config: how long is the longest walk, is the launch bound by it, what does occupancy buy?
  python profiles/probes/det_probe.py [config] [anchors|all]"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from telomeri_amd import capi, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "athaliana"
na = sys.argv[2] if len(sys.argv) > 2 else "25"
ctx = capi.Context(0)
ds = synth.make_config(cfg, device="cuda")
g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
del ds
anchors, det_off = g.anchor_table()
u1 = int(det_off[-1] if na == "all" else det_off[int(na)])
for rep in range(2):
    pe, po = g.walk_det2(0, 1, 0, u1)
for ps in (pe, po):
    st = ps.unit_stats()
    steps = st["steps"].astype(np.float64)
    ev = st["evals"].astype(np.float64)
    print("units", len(steps), "steps: sum", steps.sum(), "mean", steps.mean(), "max", steps.max(), "p99", np.percentile(steps, 99),
          "p50", np.percentile(steps, 50), "evals", ev.sum(), "kernel_ms", ps.counters.kernel_ms, "busy_ms", ps.counters.kernel_busy_ms,
          "waves", ps.counters.waves, flush=True)
    if ps.counters.kernel_ms:
        print("  us per step per wave:", ps.counters.kernel_ms * 1e3 * ps.counters.waves / (2 * steps.sum()),
              "; longest walk at that pace:", ps.counters.kernel_ms * ps.counters.waves / (2 * steps.sum()) * steps.max(), "ms", flush=True)
for wpc in (() if os.environ.get("TLM_PROBE_NOSWEEP") else (24, 16, 12, 8, 4)):
    ctx.set_walk_tuning(waves_per_cu=wpc, path_capacity=0)
    for rep in range(2):
        pe, po = g.walk_det2(0, 1, 0, u1)
    print("waves_per_cu", wpc, "kernel_ms", pe.counters.kernel_ms, "waves", pe.counters.waves, flush=True)
