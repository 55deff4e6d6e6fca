import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from telomeri_amd import capi
from oracle import pyoracle as po
import helpers as H
degrees = list(range(1, 40)) + list(range(120, 136)) + list(range(246, 300)) + list(range(370, 396)) \
    + list(range(500, 530)) + [640, 1023, 1024, 1025, 1536, 4095, 4097, 5000, 2, 3, 5, 8]
degrees = np.array(degrees * 2, np.int64)
args = T._csr_with_degrees(degrees, seed=11, zero_rows=[0, 7, 60, 80, 100, 130, 150])
ctx = capi.Context(0)
dg = ctx.graph_from_csr(*args)
og = po.Graph.from_csr(*args)
wp = capi.WalkParams.default(rb_att=150, bt_att=6, len_thr=300_000)
for rep in range(2):
    ps = dg.walk_mc(capi.METRIC_ES, wp)
    ops, octr, ostats = og.walk_mc(po.METRIC_ES, H.to_oracle_wp(wp), po.SEED_PER_WALK)
    st = ps.unit_stats()
    bad = [i for i in range(len(st)) if tuple(st[i]) != tuple(ostats[i])]
    print("rep", rep, "units", len(st), "differ", len(bad))
    for i in bad[:10]:
        print(i, "dev", st[i], "ora", ostats[i])
