"""profiles/probes/hard_probe.py <config> [rb] -- first look at a hostile config on the GPU box: generator time, graph
figures (degrees, rows that are not compact), per-anchor Monte-Carlo work with `rb` trials, deterministic unit lengths,
kernel times.  Writes gpurun_out/<config>_probe.json."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from telomeri_amd import capi, synth  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "athaliana_hard"
rb = int(sys.argv[2]) if len(sys.argv) > 2 else 50
out = {"config": config, "rb": rb}
ctx = capi.Context(0)
t = time.time()
ds = synth.make_config(config, device="cuda")
torch.cuda.synchronize()
out["generate_s"] = time.time() - t
out["records"] = [len(ds.rr), len(ds.cr)]
out["torch_peak_GB"] = torch.cuda.max_memory_allocated() / 1e9
print(out, flush=True)
t = time.time()
g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
out["build_s"] = time.time() - t
gi = g.info()
out.update({"nodes": gi.n_nodes, "adj": gi.n_adj, "kept": gi.n_kept, "anchors": gi.n_anchors, "max_degree": gi.max_degree,
            "unsafe_nodes": list(gi.unsafe_nodes), "build_ms": gi.build_ms, "phases": gi.phases()})
clen = ds.contig_len.cpu().numpy()
out["contig_len"] = {"min": int(clen.min()), "max": int(clen.max()), "mean": float(clen.mean())}
del ds
torch.cuda.empty_cache()
print(out, flush=True)
csr = g.download(cold=False)
deg = np.diff(csr.row_ptr.astype(np.int64))
out["degree"] = {"mean": float(deg.mean()), "p50": float(np.median(deg)), "p99": float(np.quantile(deg, 0.99)),
                 "over_512": int((deg > 512).sum()), "over_768": int((deg > 768).sum()), "over_4096": int((deg > 4096).sum())}
anchors, det_off = g.anchor_table()
out["anchor_degree"] = {"min": int(deg[anchors].min()), "max": int(deg[anchors].max()), "mean": float(deg[anchors].mean())}
out["det_units"] = int(det_off[-1])
del csr
wp = capi.WalkParams.default(rb_att=rb)
ctx.set_walk_accounting(True)
for metric, nm in ((capi.METRIC_ES, "es"), (capi.METRIC_OS, "os")):
    t = time.time()
    ps = g.walk_mc(metric, wp)
    wall = time.time() - t
    st = ps.unit_stats()
    c = ps.counters
    per = st["evals"].astype(np.int64).reshape(-1, rb).sum(1)
    out["mc_" + nm] = {"wall_s": wall, "kernel_ms": c.kernel_ms, "units": int(st.shape[0]), "accepted": int(ps.n_paths), "evals": int(c.evals),
                       "steps": int(c.steps), "max_steps": int(st["steps"].max()), "evals_per_s": c.evals / (c.kernel_ms * 1e-3),
                       "per_anchor_evals": {"min": int(per.min()), "max": int(per.max()), "mean": float(per.mean()),
                                            "cv": float(per.std() / per.mean())},
                       "layout_bytes_per_eval": c.layout_bytes / max(1, c.evals),
                       "step_class": [int(x) for x in c.step_class],
                       "status": {int(k): int(v) for k, v in zip(*np.unique(st["status"], return_counts=True))}}
    np.save(os.path.join(ROOT, "gpurun_out", "%s_mc_%s_per_anchor.npy" % (config, nm)), per)
    print(nm, out["mc_" + nm], flush=True)
    del ps
ctx.set_walk_accounting(False)
for metric, nm in ((capi.METRIC_ES, "es"), (capi.METRIC_OS, "os")):
    t = time.time()
    pd = g.walk_det(metric)
    wall = time.time() - t
    st = pd.unit_stats()
    c = pd.counters
    out["det_" + nm] = {"wall_s": wall, "kernel_ms": c.kernel_ms, "units": int(st.shape[0]), "accepted": int(pd.n_paths), "evals": int(c.evals),
                        "steps": int(c.steps), "max_steps": int(st["steps"].max()), "p99_steps": float(np.quantile(st["steps"], 0.99)),
                        "evals_per_s": c.evals / (c.kernel_ms * 1e-3)}
    print(nm, out["det_" + nm], flush=True)
    del pd
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "%s_probe.json" % config), "w"), indent=1)
print("done")
