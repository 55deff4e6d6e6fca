"""How many lists hold the same neighbour twice (multi-edges: PathManager.cpp:340-347 pushes the FIRST entry with the chosen
neighbour, so a deterministic step has to look for it)?  Per config: share of nodes, share of the adjacency entries that sit in
such lists, by degree class.   python profiles/probes/multi_edge_probe.py <config> [...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from telomeri_amd import capi, synth


def main():
    ctx = capi.Context(0)
    out = {}
    for cfg in sys.argv[1:]:
        ds = synth.make_config(cfg, device="cuda:0")
        g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
        del ds
        torch.cuda.empty_cache()
        csr = g.download(cold=False)
        del g
        row_ptr = torch.from_numpy(csr.row_ptr.astype(np.int64)).cuda()
        nbr = torch.from_numpy(csr.nbr.astype(np.int64)).cuda()
        n = row_ptr.numel() - 1
        deg = row_ptr[1:] - row_ptr[:-1]
        owner = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
        key = (owner << 32) | nbr
        del owner, nbr
        key, _ = torch.sort(key)
        dup = key[1:] == key[:-1]
        rows = torch.unique(key[1:][dup] >> 32)
        flag = torch.zeros(n, dtype=torch.bool, device="cuda")
        flag[rows] = True
        res = {"nodes": int(n), "entries": int(deg.sum()), "nodes_with_a_repeated_neighbour": int(flag.sum()),
               "share_of_entries_in_such_lists": float(deg[flag].sum()) / float(deg.sum())}
        for lo, hi in ((0, 256), (256, 512), (512, 768), (768, 1 << 30)):
            m = (deg > lo) & (deg <= hi)
            res["deg_%d_%d" % (lo + 1, hi)] = {"nodes": int(m.sum()), "with_repeat": int((m & flag).sum())}
        out[cfg] = res
        del key, dup, rows, flag, deg, row_ptr
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
