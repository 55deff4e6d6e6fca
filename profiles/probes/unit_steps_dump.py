"""Per-unit steps / evals of the whole config's four builders (one GPU) -> npz, the input of profiles/predict_scaling.py.
    python profiles/probes/unit_steps_dump.py <config> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from telomeri_amd import capi, synth  # noqa: E402

config, out = sys.argv[1], sys.argv[2]
ctx = capi.Context(0)
ds = synth.make_config(config, device="cuda:0")
g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
del ds
rb = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
wp = capi.WalkParams.default(rb_att=rb)
sets = list(g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp))
_, det_off = g.anchor_table()
res = {"det_off": det_off, "rebuild_attempts": np.array([rb])}
for name, ps in zip(("mc_es", "mc_os", "det_es", "det_os"), sets):
    st = ps.unit_stats()
    res[name + "_steps"] = st["steps"].astype(np.uint32)
    res[name + "_evals"] = st["evals"].astype(np.uint64)
    res[name + "_nodes"] = np.where(st["status"] == 1, st["n_nodes"], 0).astype(np.uint32)
    res[name + "_kernel_ms"] = np.array([ps.counters.kernel_ms, ps.counters.kernel_busy_ms])
np.savez_compressed(out, **res)
print("wrote", out, {k: (v.shape, int(v.sum()) if v.dtype.kind in "ui" else None) for k, v in res.items()})
