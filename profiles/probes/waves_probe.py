"""Resident persistent waves of the walk kernels on this GPU (occupancy check after LDS / register changes)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from telomeri_amd import capi, synth
ctx = capi.Context(0)
ds = synth.make_config("ecoli", device="cuda")
g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
wp = capi.WalkParams.default(rb_att=20000)
ps = g.walk_mc(0, wp, 0, 40000)
print("mc waves", ps.counters.waves, "kernel_ms", ps.counters.kernel_ms)
pd = g.walk_det(0)
print("det waves", pd.counters.waves, "kernel_ms", pd.counters.kernel_ms)
