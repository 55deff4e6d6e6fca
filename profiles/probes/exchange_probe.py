#!/usr/bin/env python3
"""profiles/probes/exchange_probe.py [config] -- what the lazy exchange (tlm_pathset_gather_summaries) costs on the HOST side for
the path set of a whole config: one rank, a one-rank RCCL communicator, TLM_EXCHANGE_SELFTEST (the whole exchange path runs;
nothing crosses a link).  The host-side share -- staging, the D2H copy of the slabs, the paths_-order pass over every summary --
is the same at any world size, because every rank ends with ALL summaries."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TLM_EXCHANGE_SELFTEST"] = "1"

import torch
from telomeri_amd import capi, synth


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "athaliana"
    ctx = capi.Context(0)
    ds = synth.make_config(config, device="cuda:0")
    fp = capi.FilterParams.default()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    del ds
    torch.cuda.empty_cache()
    wp = capi.WalkParams.default(2000, 60, 4_700_000)
    sets = list(g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp))
    ctx.dist_init(0, 1, capi.dist_unique_id())
    out = {"config": config, "paths": int(sum(s.n_paths for s in sets)), "runs": []}
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        ga = capi.Gathered(sets)
        t1 = time.time()
        xi = ga.info
        out["runs"].append({"python_wall_ms": 1e3 * (t1 - t0), "wall_ms": float(xi.wall_ms), "device_ms": float(xi.device_ms)})
        del ga
    print(json.dumps(out))


if __name__ == "__main__":
    main()
