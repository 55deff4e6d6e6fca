"""What ONE rank of N walks under the sliced partition, on one GPU: rank 0's share of the units of a config
(tlm_dist_unit_share: slice 0 of every anchor's trials and first edges) through tlm_walk_builders_list, timed like a bench step
(no exchange).  The per-rank step a job of N GPUs would see if the links added nothing -- a measured input to the scaling
prediction of DESIGN.md §7, not a multi-GPU number.   python profiles/probes/rank_share_probe.py <config> <N> [<N> ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from telomeri_amd import capi, synth


def main():
    cfg = sys.argv[1]
    ctx = capi.Context(0)
    ds = synth.make_config(cfg, device="cuda:0")
    g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    del ds
    torch.cuda.empty_cache()
    wp = capi.WalkParams.default(2000, 60, 4_700_000)
    out = {"config": cfg, "rebuild_attempts": 2000, "shares": []}
    for world in [int(x) for x in sys.argv[2:]]:
        mc = g.unit_share(0, world, wp, False)
        det = g.unit_share(0, world, wp, True)
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            sets = list(g.walk_builders_list(capi.METRIC_ES, capi.METRIC_OS, wp, mc, det))
            torch.cuda.synchronize()
            ms = 1e3 * (time.time() - t0)
            mc_ms = sets[0].counters.kernel_ms
            mc_evals = int(sets[0].counters.evals) + int(sets[1].counters.evals)
            det_evals = int(sets[2].counters.evals) + int(sets[3].counters.evals)
            rec = {"world": world, "mc_units": int(mc.size), "det_units": int(det.size), "step_ms": ms, "walk_mc_ms": mc_ms,
                   "mc_evals": mc_evals, "det_evals": det_evals, "frac_8B_per_eval": 8.0 * mc_evals / (mc_ms * 1e-3) / 8e12,
                   "paths": int(sum(s.n_paths for s in sets))}
            if rep and (best is None or ms < best["step_ms"]):
                best = rec
            del sets
        out["shares"].append(best)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
