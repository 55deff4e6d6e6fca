"""Graph build, repeated: device time of the 1st .. 4th build of the same A. thaliana-shaped input in one process."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from telomeri_amd import capi, synth
ctx = capi.Context(0)
ds = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "athaliana", device="cuda")
torch.cuda.synchronize()
fp = capi.FilterParams.default()
g = None
for i in range(4):
    t0 = time.time()
    if g is not None:
        del g
    t1 = time.time()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    t2 = time.time()
    print("build %d: destroy of the previous %.1f ms, build wall %.1f ms, device %.1f ms" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), g.info().build_ms))
