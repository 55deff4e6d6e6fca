"""One rank of a multi-rank exchange test on ONE GPU (tests/test_exchange_world.py starts several of these).

    python tests/exchange_worker.py <dataset> <rank> <world> <id file> <out dir>

Every rank builds the same graph on cuda:0, walks its share of the units (sliced or contiguous partition), then runs the
product's exchange -- tlm_pathset_allgather, tlm_pathset_gather_summaries + tlm_gathered_fetch -- through the library's own
communicator, with tests/fake_rccl underneath (TLM_RCCL_LIB, set by the parent).  What it received goes into
<out dir>/rank<r>.npz; a failure is written to rank<r>.err (code + message) and the exit status is 3.
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

RB = {"toy": 30, "quirks": 30, "hub": 80}


def hub_csr(seed: int = 11):
    """A random graph whose FIRST anchor owns a list of 70 000 entries (multi-edges): edge slots beyond 65 534 occur in its
    paths, so the exchange must fall back to 32-bit slots on the wire -- decided by ONE rank's data, agreed by all."""
    rng = np.random.default_rng(seed)
    n = 60
    anchors = np.zeros(n, np.uint8)
    anchors[[0, 31, 59]] = 1
    rows = []
    for i in range(n):
        d = 70_000 if i == 0 else int(rng.integers(3, 40))
        nb = rng.integers(1, n, size=d)
        nb = nb[nb != i]
        if i == 0:      # no anchor in the hub's list: no forced-anchor edge, no truncated unit list -- the whole list is walked from
            nb = nb[anchors[nb] == 0]
        rows.append(nb)
    row_ptr = np.zeros(n + 1, np.uint64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    nbr = np.concatenate(rows).astype(np.uint32)
    a = nbr.shape[0]
    es = rng.integers(1, 20000, size=a).astype(np.float32)
    os_ = rng.integers(1, 9000, size=a).astype(np.float32)
    q_start = rng.integers(3000, 9000, size=a).astype(np.uint32)
    t_start = rng.integers(0, 3000, size=a).astype(np.uint32)
    node_len = rng.integers(20000, 90000, size=n).astype(np.uint32)
    return row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors


def make_graph(ctx, dataset: str):
    from telomeri_amd import capi, synth
    import helpers as H
    if dataset == "hub":
        return ctx.graph_from_csr(*hub_csr()), capi.WalkParams.default(rb_att=RB[dataset], bt_att=5)
    if dataset == "quirks":      # runs of identical paths (contig<->contig edge), empty deterministic blocks
        return H.device_graph(ctx, H.quirks_dataset(), capi.FilterParams.default("min")), capi.WalkParams.default(rb_att=RB[dataset], bt_att=5)
    return H.device_graph(ctx, synth.make_config(dataset), capi.FilterParams.default()), capi.WalkParams.default(rb_att=RB[dataset])


def rank_sets(dg, wp, rank: int, world: int):
    """The four builder sets of one rank's share: its slices of every anchor's units (tlm_dist_unit_share, the hosts'
    default) or, with TLM_TEST_PARTITION=contiguous, its bundle of whole anchors (the unit ranges of round 3)."""
    from telomeri_amd import capi
    if os.environ.get("TLM_TEST_PARTITION", "sliced") == "sliced":
        return list(dg.walk_builders_list(capi.METRIC_ES, capi.METRIC_OS, wp, dg.unit_share(rank, world, wp, False),
                                          dg.unit_share(rank, world, wp, True)))
    a0, a1 = dg.anchor_range(rank, world)
    _, det_off = dg.anchor_table()
    rb = wp.rebuild_attempts
    return list(dg.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp, (a0 * rb, a1 * rb), (int(det_off[a0]), int(det_off[a1]))))


def pick_indices(n_paths: int) -> np.ndarray:
    """The request every rank makes of tlm_gathered_fetch: both ends, a repeat, an out-of-order spread over all owners."""
    if n_paths == 0:
        return np.zeros(0, np.uint64)
    idx = [n_paths - 1, 0, n_paths // 2, n_paths // 2] + list(range(n_paths - 1, -1, -max(1, n_paths // 23)))
    return np.array(idx, np.uint64)


def fault_run(capi, ctx, sets, rank: int, outdir: str) -> int:
    """With TLM_EXCHANGE_FAULT one rank pretends a local step failed.  Every rank makes the same calls -- the full exchange,
    the lazy one, a fetch, the fetch again -- and writes down how each ended: a failure anywhere must end the call on EVERY
    rank (none may hang), and once a communicator has been aborted later calls fail at once."""
    log = []

    def attempt(what, fn):
        try:
            r = fn()
            log.append("%s: ok" % what)
            return r
        except capi.TlmError as e:
            log.append("%s: %s" % (what, e))
            return None

    attempt("allgather", lambda: capi.PathSet.allgather(sets))
    g = attempt("gather_summaries", lambda: capi.Gathered(sets))
    if g is not None:
        e = g.export()
        pick = pick_indices(g.n_paths)
        attempt("fetch", lambda: g.fetch(pick, e["n_nodes"]))
        attempt("fetch again", lambda: g.fetch(pick, e["n_nodes"]))
    with open(os.path.join(outdir, "rank%d.log" % rank), "w") as f:
        f.write("\n".join(log) + "\n")
    return 0


def load_graph(ctx, files, fp_mode="avg"):
    """The sharded load's subject: the PAF files of a graph through tlm_paf_add_file (every rank of a communicator brings its
    run of pieces of each file and the ranks swap them), names, graph.  -> dict of host arrays + the handle's share info."""
    from telomeri_amd import capi
    paf = capi.PafTokens(ctx)
    recs = []
    for path, anchors in files:
        n, clean = paf.add_file(path, anchors)
        assert clean, path
        recs.append(n)
    names, is_ctg = paf.finish()
    cols = [paf.columns(f) for f in range(len(files))]
    g = paf.build_graph(capi.FilterParams.default(fp_mode))
    csr = g.download(cold=True)
    out = {"records": np.array(recs, np.int64), "names": np.array(names), "is_ctg": is_ctg}
    for f, c in enumerate(cols):
        out.update({"f%d_%s" % (f, k): v for k, v in c.items()})
    for k in ("row_ptr", "nbr", "q_start", "q_end", "t_start", "t_end", "strand", "node_len", "node_anchor"):
        out["csr_" + k] = getattr(csr, k)
    for k in ("os", "si", "es"):
        out["csr_" + k] = getattr(csr, k).view(np.uint32)
    info = paf.share_info()
    out["share"] = np.array([info.bytes_sent, info.bytes_received], np.int64)
    return out


def load_run(capi, ctx, rank: int, outdir: str) -> int:
    """dataset "load:<rr path>:<cr path>": the sharded load alone (tests/test_exchange_world.py::test_sharded_load_*)."""
    rr, cr = sys.argv[1].split(":")[1:3]
    if os.environ.get("TLM_TEST_OTHER_FILE_RANK") == str(rank):        # this rank is shown another file under the same role
        rr += ".other"
    try:
        out = load_graph(ctx, [(rr, False), (cr, True)])
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
        return 0
    except capi.TlmError as e:
        with open(os.path.join(outdir, "rank%d.err" % rank), "w") as f:
            f.write(str(e))
        return 3


def main() -> int:
    dataset, rank, world, idfile, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    from telomeri_amd import capi
    try:
        ctx = capi.Context(0)
        if rank == 0:
            uid = capi.dist_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                if time.time() - t0 > 120:
                    raise RuntimeError("no id file")
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        ctx.dist_init(rank, world, uid)
        if dataset.startswith("load:"):
            return load_run(capi, ctx, rank, outdir)
        dg, wp = make_graph(ctx, dataset)
        sets = rank_sets(dg, wp, rank, world)
        out = {"local_paths": np.array([s.n_paths for s in sets], np.int64)}
        if os.environ.get("TLM_EXCHANGE_FAULT"):
            return fault_run(capi, ctx, sets, rank, outdir)
        full, info = capi.PathSet.allgather(sets)
        h = full.to_host()
        out.update(full_unit=h.unit, full_length=h.length, full_off=h.off, full_nodes=h.nodes, full_slots=h.slots,
                   full_info=np.array([info.bytes_sent, info.bytes_received, info.slot_bits], np.int64))
        g = capi.Gathered(sets)
        e = g.export()
        out.update({"lazy_" + k: v for k, v in e.items()})
        out["lazy_info"] = np.array([g.info.bytes_sent, g.info.bytes_received, g.n_paths, g.n_local_paths], np.int64)
        pick = pick_indices(g.n_paths)
        nodes, slots = g.fetch(pick, e["n_nodes"])
        out.update(fetch_nodes=nodes, fetch_slots=slots)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
        del g, full
        ctx.dist_finalize()
        return 0
    except capi.TlmError as e:
        with open(os.path.join(outdir, "rank%d.err" % rank), "w") as f:
            f.write(str(e))
        return 3
    except Exception:
        with open(os.path.join(outdir, "rank%d.err" % rank), "w") as f:
            f.write("UNEXPECTED " + traceback.format_exc())
        return 4


if __name__ == "__main__":
    rc = main()
    sys.stdout.flush()
    os._exit(rc)        # a failed rank must not wait for destructors that talk to an aborted communicator
