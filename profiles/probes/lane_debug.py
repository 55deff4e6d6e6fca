"""Debug probe: the lane-width graph of tests/test_gpu_parity.py::test_walks_across_lane_widths, device vs oracle, every
per-unit field of the first mismatching units (and how many units differ per field)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from oracle import pyoracle as po
from telomeri_amd import capi
import helpers as H
ctx = capi.Context(0)
degrees = list(range(1, 40)) + list(range(120, 136)) + list(range(246, 300)) + list(range(370, 396)) \
    + list(range(500, 530)) + [640, 1023, 1024, 1025, 1536, 4095, 4097, 5000, 2, 3, 5, 8]
degrees = np.array(degrees * 2, np.int64)
args = T._csr_with_degrees(degrees, seed=11, zero_rows=[0, 7, 60, 80, 100, 130, 150])
dg = ctx.graph_from_csr(*args)
og = po.Graph.from_csr(*args)
wp = capi.WalkParams.default(rb_att=150, bt_att=6, len_thr=300_000)
deg = np.diff(args[0].astype(np.int64))
for rep in range(2):
    ps = dg.walk_mc(capi.METRIC_ES, wp)
    ops, octr, ost = og.walk_mc(capi.METRIC_ES, H.to_oracle_wp(wp), po.SEED_PER_WALK)
    st = ps.unit_stats()
    bad = np.zeros(st.shape[0], bool)
    for f in ("status", "n_nodes", "steps", "evals", "draws", "length"):
        d = st[f] != ost[f]
        bad |= d
        print("rep", rep, f, "differs at", int(d.sum()), "of", st.shape[0], "units; first", np.nonzero(d)[0][:12])
    for u in np.nonzero(bad)[0][:6]:
        print(" unit", u, "device", st[u], "oracle", ost[u])
    print("waves", ps.counters.waves, "launches", ps.counters.kernel_launches)
    # where do the oracle's accepted paths go: max degree met on paths of bad vs good units
    good_units = set(np.nonzero(~bad)[0].tolist())
    mx_good, mx_bad = [], []
    for i in range(len(ops)):
        l, nodes, slots = ops.path(i)
        (mx_good if int(ops.unit[i]) in good_units else mx_bad).append(int(deg[nodes].max()))
    print(" oracle-accepted paths: good units max-degree-on-path quantiles", np.quantile(mx_good, [0, .5, 1]) if mx_good else None,
          "bad", np.quantile(mx_bad, [0, .5, 1]) if mx_bad else None)
