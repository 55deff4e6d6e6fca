"""GPU parity proper: the HIP path, called through the C ABI, against the oracle (CPU restatement of
the reference, pinned by tests/test_oracle_*.py).  Bar: bit-exact -- node numbering, adjacency order,
fp32 bit patterns of SI/OS/ES, every accepted path (nodes, edge slots, length, order) and the
per-unit counters (steps, evals, draws) under the deterministic builder and per-walk-seeded MC.
"""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import pyoracle as po
from telomeri_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return capi.Context(0)


@pytest.fixture(scope="module")
def ctx_full():
    """A context whose walks always launch the FULL instantiations of the kernels (TLM_WALK_FULL_KERNEL is read once, when a
    context is made)."""
    import os
    os.environ["TLM_WALK_FULL_KERNEL"] = "1"
    try:
        return capi.Context(0)
    finally:
        del os.environ["TLM_WALK_FULL_KERNEL"]


def _walk_both(ctx, dg, og, metric, wp, *, mc, unit_begin=0, unit_end=None):
    if mc:
        ps = dg.walk_mc(metric, wp, unit_begin, unit_end)
        ops, octr, ostats = og.walk_mc(metric, H.to_oracle_wp(wp), po.SEED_PER_WALK, unit_begin, unit_end)
    else:
        ps = dg.walk_det(metric, unit_begin, unit_end)
        ops, octr, ostats = og.walk_det(metric, unit_begin, unit_end)
    H.assert_stats_equal(ps.unit_stats(), ostats)
    H.assert_paths_equal(ps.to_host(), ops)
    c = ps.counters
    assert (c.units, c.steps, c.evals, c.draws, c.accepted) == (octr.units, octr.steps, octr.evals, octr.draws,
                                                                octr.accepted)
    return ps, ops


# ---------------------------------------------------------------------------------------------
# a2 + a3

@pytest.mark.parametrize("mode,extra", [("avg", {}), ("min", {}), ("max", {}), ("sum", {}),
                                        ("avg", dict(min_oll=5000, min_olp=0.05, max_ohl=3000, max_ohp=0.1))])
def test_filter_score_bits(ctx, mode, extra):
    ds = synth.make_config("toy")
    rec = ds.rr.cat(ds.cr)
    fp = capi.FilterParams.default(mode, **extra)
    keep, si, os_, es, nk = ctx.filter_score(fp, rec.to("cuda"))
    cols = [getattr(rec, c).numpy() for c in H.REC_COLS[:7]]
    same = (rec.q_id == rec.t_id).numpy().astype(np.uint8)
    okeep, osi, oos, oes = po.filter_score(H.to_oracle_fp(fp), *cols, same)
    np.testing.assert_array_equal(keep.cpu().numpy(), okeep)
    assert nk == int(okeep.sum())
    for dev, ora in ((si, osi), (os_, oos), (es, oes)):
        np.testing.assert_array_equal(dev.cpu().numpy().view(np.uint32), ora.view(np.uint32))


def test_filter_score_edge_values(ctx):
    """zero-length sequences, negative overlaps, zero overlap (0/0), INT extremes: same bits as x86."""
    i32 = lambda v: torch.tensor(v, dtype=torch.int32)
    rec = synth.Records(q_id=i32([0, 1, 2, 3, 4, 5]), t_id=i32([1, 2, 3, 4, 5, 5]),
                        q_len=i32([0, 1000, 1000, 2_000_000_000, 5, 10]), q_start=i32([0, 500, 10, 0, 0, 0]),
                        q_end=i32([0, 400, 10, 1_000_000_000, 5, 10]), t_len=i32([0, 1000, 1000, 2_000_000_000, 7, 10]),
                        t_start=i32([0, 0, 20, 5, 1, 0]), t_end=i32([0, 900, 20, 1_000_000_005, 6, 10]),
                        matches=i32([0, 100, 0, 900_000_000, 5, 10]), strand=i32([0, 1, 0, 1, 0, 0]))
    for mode in ("avg", "min", "max", "sum"):
        fp = capi.FilterParams.default(mode)
        keep, si, os_, es, nk = ctx.filter_score(fp, rec.to("cuda"))
        cols = [getattr(rec, c).numpy() for c in H.REC_COLS[:7]]
        okeep, osi, oos, oes = po.filter_score(H.to_oracle_fp(fp), *cols, (rec.q_id == rec.t_id).numpy().astype(np.uint8))
        np.testing.assert_array_equal(keep.cpu().numpy(), okeep, err_msg=mode)
        for dev, ora in ((si, osi), (os_, oos), (es, oes)):
            np.testing.assert_array_equal(dev.cpu().numpy().view(np.uint32), ora.view(np.uint32), err_msg=mode)


def test_filter_score_empty(ctx):
    ds = synth.make_config("tiny")
    rec = ds.rr.take(torch.zeros(0, dtype=torch.long))
    keep, si, os_, es, nk = ctx.filter_score(capi.FilterParams.default(), rec.to("cuda"))
    assert nk == 0 and keep.numel() == 0


# ---------------------------------------------------------------------------------------------
# a4

@pytest.mark.parametrize("name,mode", [("tiny", "avg"), ("toy", "avg"), ("toy", "min"), ("sparse", "avg"),
                                       ("dense", "avg")])
def test_graph_build_matches_oracle(ctx, name, mode):
    ds = synth.make_config(name)
    fp = capi.FilterParams.default(mode)
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    H.assert_csr_equal(dg.download(), og.csr())
    gi = dg.info()
    assert gi.n_records == len(ds.rr) + len(ds.cr)
    assert gi.n_kept * 2 == gi.n_adj
    ids = dg.download().node_name_id
    assert [og.node_ext_id(i) for i in range(len(ids))] == ids.tolist()


def test_graph_build_quirks(ctx):
    ds = H.quirks_dataset()
    for mode in ("avg", "min"):
        fp = capi.FilterParams.default(mode)
        dg = H.device_graph(ctx, ds, fp)
        og = H.oracle_graph(ds, H.to_oracle_fp(fp))
        H.assert_csr_equal(dg.download(), og.csr())
        assert dg.det_units() == og.det_units()


def test_graph_build_single_file_and_empty(ctx):
    ds = synth.make_config("tiny")
    fp = capi.FilterParams.default()
    dg = H.device_graph(ctx, ds, fp, files=[(ds.cr, True)])
    og = H.oracle_graph(ds, H.to_oracle_fp(fp), files=[(ds.cr, True)])
    H.assert_csr_equal(dg.download(), og.csr())
    # a filter that keeps nothing: an empty graph, no walk units, empty path sets
    fp0 = capi.FilterParams.default("avg", min_oll=10 ** 9)
    dg0 = H.device_graph(ctx, ds, fp0)
    assert dg0.info().n_nodes == 0 and dg0.info().n_adj == 0
    wp = capi.WalkParams.default(rb_att=5)
    assert dg0.mc_units(wp) == 0 and dg0.det_units() == 0
    ps = dg0.walk_mc(capi.METRIC_ES, wp)
    assert ps.n_paths == 0 and ps.n_units == 0
    assert len(ps.to_host()) == 0


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_graph_build(ctx, seed):
    """Random records, degenerate ones included (start beyond end, zero lengths, matches above the overlap, the same
    pair many times, a name only ever seen as query, contigs as queries): filter, scores and adjacency bit for bit."""
    import dataclasses
    rng = np.random.default_rng(500 + seed)
    base = synth.make_config("tiny")
    n_names = base.n_reads + base.n_contigs

    def records(n, ctg_targets):
        ln = rng.integers(1, 60_000, size=n_names)
        q = rng.integers(0, n_names, size=n)
        t = rng.integers(base.n_reads, n_names, size=n) if ctg_targets else rng.integers(0, n_names, size=n)
        if rng.random() < 0.5:                      # a few pairs many times over (multi-edges)
            k = max(1, n // 5)
            q[:k], t[:k] = q[0], t[0]
        qs = (rng.random(n) * ln[q]).astype(np.int64)
        qe = np.minimum(qs + rng.integers(0, 30_000, size=n), ln[q])
        ts = (rng.random(n) * ln[t]).astype(np.int64)
        te = np.minimum(ts + rng.integers(0, 30_000, size=n), ln[t])
        bad = rng.random(n) < 0.05                  # start beyond end
        qs[bad], qe[bad] = qe[bad] + 3, qs[bad]
        m = (rng.random(n) * (np.abs(qe - qs) + 5)).astype(np.int64)
        i32 = lambda v: torch.tensor(np.asarray(v), dtype=torch.int32)
        return synth.Records(i32(q), i32(t), i32(ln[q]), i32(qs), i32(qe), i32(ln[t]), i32(ts), i32(te), i32(m),
                             i32(rng.integers(0, 2, size=n)))

    ds = dataclasses.replace(base, rr=records(int(rng.integers(0, 400)), False), cr=records(int(rng.integers(1, 60)), True))
    mode = str(rng.choice(["avg", "min", "max", "sum"]))
    extra = {} if rng.random() < 0.5 else dict(min_oll=int(rng.integers(0, 8000)), min_olp=float(rng.random() * 0.3),
                                               max_ohl=int(rng.integers(100, 40_000)), max_ohp=float(rng.random()))
    fp = capi.FilterParams.default(mode, **extra)
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    H.assert_csr_equal(dg.download(), og.csr())
    assert dg.det_units() == og.det_units()
    ids = dg.download().node_name_id
    assert [og.node_ext_id(i) for i in range(len(ids))] == ids.tolist()


# ---------------------------------------------------------------------------------------------
# a7 / a8

@pytest.mark.parametrize("name,rb,bt", [("tiny", 60, 60), ("toy", 40, 60), ("sparse", 80, 3), ("sparse", 40, 0),
                                        ("dense", 30, 60)])
@pytest.mark.parametrize("metric", [capi.METRIC_ES, capi.METRIC_OS])
def test_walk_mc_per_walk_seed(ctx, name, rb, bt, metric):
    ds = synth.make_config(name)
    fp = capi.FilterParams.default()
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    wp = capi.WalkParams.default(rb_att=rb, bt_att=bt)
    assert dg.mc_units(wp) == og.mc_units(H.to_oracle_wp(wp))
    ps, ops = _walk_both(ctx, dg, og, metric, wp, mc=True)
    if name != "sparse":
        assert ps.n_paths > 0          # the comparison is not vacuous


@pytest.mark.parametrize("name", ["tiny", "toy", "sparse", "dense"])
@pytest.mark.parametrize("metric", [capi.METRIC_ES, capi.METRIC_OS])
def test_walk_det(ctx, name, metric):
    ds = synth.make_config(name)
    fp = capi.FilterParams.default()
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    assert dg.det_units() == og.det_units()
    _walk_both(ctx, dg, og, metric, None, mc=False)


@pytest.mark.parametrize("metric", [capi.METRIC_ES, capi.METRIC_OS, capi.METRIC_ES_SQRT, capi.METRIC_OS_SQRT])
def test_walks_quirks_all_metrics(ctx, metric):
    """multi-edges (DET re-find of the first entry with the same neighbour), contig<->contig edges
    (forced anchor rule, DET first-edge break), filter-min graph with high-degree contigs."""
    ds = H.quirks_dataset()
    fp = capi.FilterParams.default("min")
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    wp = capi.WalkParams.default(rb_att=50, bt_att=5)
    ps, _ = _walk_both(ctx, dg, og, metric, wp, mc=True)
    pd, _ = _walk_both(ctx, dg, og, metric, None, mc=False)
    assert ps.n_paths + pd.n_paths > 0


@pytest.mark.parametrize("name,rb", [("toy", 30), ("quirks", 24), ("hard_tiny", 40)])
def test_unit_lists_equal_unit_ranges(ctx, name, rb):
    """(ABI 4) tlm_walk_list / tlm_walk_builders_list: a LIST of units returns, unit for unit, what the range calls return
    for those units -- random sublists, and the sliced shares of 3 ranks put together again (tlm_dist_unit_share)."""
    ds = H.dataset_by_name(name)
    dg = H.device_graph(ctx, ds, capi.FilterParams.default("min" if name == "quirks" else "avg"))
    wp = capi.WalkParams.default(rb_att=rb, bt_att=6)
    whole = dg.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp)
    whole_h = [ps.to_host() for ps in whole]
    whole_st = [ps.unit_stats() for ps in whole]
    rng = np.random.default_rng(7)

    def check(sets, mc_units, det_units):
        for k, ps in enumerate(sets):
            units = (mc_units if k < 2 else det_units).astype(np.int64)
            np.testing.assert_array_equal(ps.unit_stats(), whole_st[k][units])
            h, w = ps.to_host(), whole_h[k]
            sel = np.isin(w.unit.astype(np.int64), units)
            np.testing.assert_array_equal(h.unit, w.unit[sel])
            np.testing.assert_array_equal(h.length, w.length[sel])
            cnt = np.diff(w.off.astype(np.int64))
            np.testing.assert_array_equal(h.nodes, w.nodes[np.repeat(sel, cnt)])
            np.testing.assert_array_equal(h.slots, w.slots[np.repeat(sel, cnt)])
            assert ps.counters.evals == int(whole_st[k]["evals"][units].sum())

    n_mc, n_det = dg.mc_units(wp), dg.det_units()
    mc_units = np.sort(rng.choice(n_mc, size=n_mc // 3, replace=False)).astype(np.uint64)
    det_units = np.sort(rng.choice(n_det, size=max(1, n_det // 2), replace=False)).astype(np.uint64)
    check(dg.walk_builders_list(capi.METRIC_ES, capi.METRIC_OS, wp, mc_units, det_units), mc_units, det_units)
    one = dg.walk_list(False, capi.METRIC_OS, -1, wp, mc_units)          # single builder, single metric
    np.testing.assert_array_equal(one.unit_stats(), whole_st[1][mc_units.astype(np.int64)])
    two = dg.walk_list(True, capi.METRIC_ES, capi.METRIC_OS, wp, det_units)
    np.testing.assert_array_equal(two[1].unit_stats(), whole_st[3][det_units.astype(np.int64)])
    # the sliced partition: three ranks' shares are disjoint, cover everything, and each walks exactly its units
    world = 3
    shares = [(dg.unit_share(r, world, wp, False), dg.unit_share(r, world, wp, True)) for r in range(world)]
    for det, total in ((0, n_mc), (1, n_det)):
        allu = np.concatenate([sh[det] for sh in shares])
        assert allu.shape[0] == total and np.unique(allu).shape[0] == total
    for mcu, detu in shares:
        assert np.all(np.diff(mcu.astype(np.int64)) > 0) and np.all(np.diff(detu.astype(np.int64)) > 0)
        assert abs(mcu.shape[0] - n_mc / world) <= dg.info().n_anchors          # a slice of EVERY anchor's trials
        check(dg.walk_builders_list(capi.METRIC_ES, capi.METRIC_OS, wp, mcu, detu), mcu, detu)
    # refused: a list that is not ascending, a unit that does not exist
    with pytest.raises(capi.TlmError):
        dg.walk_list(False, capi.METRIC_ES, -1, wp, np.array([5, 5], np.uint64))
    with pytest.raises(capi.TlmError):
        dg.walk_list(True, capi.METRIC_ES, -1, wp, np.array([n_det], np.uint64))
    empty = dg.walk_builders_list(capi.METRIC_ES, capi.METRIC_OS, wp, np.zeros(0, np.uint64), np.zeros(0, np.uint64))
    assert [ps.n_paths for ps in empty] == [0, 0, 0, 0]


def test_walk_unit_ranges_concatenate(ctx):
    """Sharding contract: any partition of the units yields the same paths."""
    ds = synth.make_config("tiny")
    fp = capi.FilterParams.default()
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    wp = capi.WalkParams.default(rb_att=30)
    total = dg.mc_units(wp)
    cut = total // 3 + 1
    whole = dg.walk_mc(capi.METRIC_ES, wp).to_host()
    a, _ = _walk_both(ctx, dg, og, capi.METRIC_ES, wp, mc=True, unit_begin=0, unit_end=cut)
    b, _ = _walk_both(ctx, dg, og, capi.METRIC_ES, wp, mc=True, unit_begin=cut, unit_end=total)
    ha, hb = a.to_host(), b.to_host()
    np.testing.assert_array_equal(np.concatenate([ha.unit, hb.unit]), whole.unit)
    np.testing.assert_array_equal(np.concatenate([ha.nodes, hb.nodes]), whole.nodes)
    np.testing.assert_array_equal(np.concatenate([ha.slots, hb.slots]), whole.slots)
    dtot = dg.det_units()
    d1, _ = _walk_both(ctx, dg, og, capi.METRIC_OS, None, mc=False, unit_begin=0, unit_end=dtot // 2)
    d2, _ = _walk_both(ctx, dg, og, capi.METRIC_OS, None, mc=False, unit_begin=dtot // 2, unit_end=dtot)
    dw = dg.walk_det(capi.METRIC_OS).to_host()
    np.testing.assert_array_equal(np.concatenate([d1.to_host().nodes, d2.to_host().nodes]), dw.nodes)


def test_two_metric_launch_equals_two_launches(ctx):
    """tlm_walk_mc2 / tlm_walk_det2: both metrics in one persistent launch, same results as separate launches."""
    ds = synth.make_config("toy")
    fp = capi.FilterParams.default()
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    wp = capi.WalkParams.default(rb_att=30)
    for (ma, mb) in ((capi.METRIC_ES, capi.METRIC_OS), (capi.METRIC_OS_SQRT, capi.METRIC_ES_SQRT)):
        a, b = dg.walk_mc2(ma, mb, wp, 7, 83)
        for ps, m in ((a, ma), (b, mb)):
            ops, octr, ostats = og.walk_mc(m, H.to_oracle_wp(wp), po.SEED_PER_WALK, 7, 83)
            H.assert_stats_equal(ps.unit_stats(), ostats)
            H.assert_paths_equal(ps.to_host(), ops)
            assert (ps.counters.evals, ps.counters.accepted) == (octr.evals, octr.accepted)
        assert a.counters.kernel_launches >= 1 and b.counters.kernel_launches == 0
        da, db = dg.walk_det2(ma, mb)
        for ps, m in ((da, ma), (db, mb)):
            ops, octr, ostats = og.walk_det(m)
            H.assert_stats_equal(ps.unit_stats(), ostats)
            H.assert_paths_equal(ps.to_host(), ops)
    with pytest.raises(capi.TlmError):
        dg.walk_mc2(capi.METRIC_ES, capi.METRIC_OS_SQRT, wp)


@pytest.mark.parametrize("config,rb", [("toy", 30), ("sparse", 60)])
def test_builders_call_equals_separate_calls(ctx, config, rb):
    """tlm_walk_builders (DET queued on a second stream behind MC): the four Path sets of the four separate calls, and
    of the oracle."""
    ds = synth.make_config(config)
    fp = capi.FilterParams.default()
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    wp = capi.WalkParams.default(rb_att=rb)
    for rep in range(2):        # the second round reuses both workspaces
        sets = dg.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp)
        for ps, m in zip(sets[:2], (capi.METRIC_ES, capi.METRIC_OS)):
            ops, octr, ostats = og.walk_mc(m, H.to_oracle_wp(wp), po.SEED_PER_WALK)
            H.assert_stats_equal(ps.unit_stats(), ostats)
            H.assert_paths_equal(ps.to_host(), ops)
        for ps, m in zip(sets[2:], (capi.METRIC_ES, capi.METRIC_OS)):
            ops, octr, ostats = og.walk_det(m)
            H.assert_stats_equal(ps.unit_stats(), ostats)
            H.assert_paths_equal(ps.to_host(), ops)
    # unit ranges (the sharding contract) pass through
    n_mc, n_det = dg.mc_units(wp), dg.det_units()
    part = dg.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp, (n_mc // 3, n_mc // 2), (1, max(1, n_det // 2)))
    whole = dg.walk_mc(capi.METRIC_OS, wp).unit_stats()
    np.testing.assert_array_equal(part[1].unit_stats(), whole[n_mc // 3:n_mc // 2])
    np.testing.assert_array_equal(part[3].unit_stats(), dg.walk_det(capi.METRIC_OS).unit_stats()[1:max(1, n_det // 2)])


@pytest.mark.parametrize("name,mode", [("quirks", "min"), ("tiny", "avg"), ("toy", "avg")])
def test_filter_unique_and_anchor_pairs_on_device(ctx, name, mode):
    """SURVEY.md §8f row 1 on the packed Path sets: concat (paths_ of one PathManager), filterUnique,
    getPathsBetweenAnchors -- against the oracle's restatement (pinned to the reference in test_oracle_golden.py)."""
    ds = H.dataset_by_name(name)
    fp = capi.FilterParams.default(mode)
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    wp = capi.WalkParams.default(rb_att=40, bt_att=5)
    owp = H.to_oracle_wp(wp)
    dsets = list(dg.walk_mc2(capi.METRIC_ES, capi.METRIC_OS, wp)) + list(dg.walk_det2(capi.METRIC_ES, capi.METRIC_OS))
    osets = [og.walk_mc(m, owp, po.SEED_PER_WALK)[0] for m in (0, 1)] + [og.walk_det(m)[0] for m in (0, 1)]
    dall = capi.PathSet.concat(dsets)
    oall = po.concat_paths(osets)
    H.assert_paths_equal(dall.to_host(), oall)
    duniq = dall.filter_unique()
    ouniq = po.filter_unique(oall)
    H.assert_paths_equal(duniq.to_host(), ouniq)
    if name == "quirks":
        assert len(ouniq) < len(oall)          # the contig<->contig edge makes every trial of one anchor identical
    assert duniq.anchor_pairs() == po.paths_between_anchors(ouniq)
    assert dall.anchor_pairs() == po.paths_between_anchors(oall)
    # idempotent, and an empty set stays empty
    H.assert_paths_equal(duniq.filter_unique().to_host(), ouniq)
    empty = dg.walk_det(capi.METRIC_ES, 0, 0)
    assert empty.filter_unique().n_paths == 0 and empty.anchor_pairs() == []
    # the consuming variant: the filtered set when something is dropped, the set itself when nothing is
    H.assert_paths_equal(capi.PathSet.concat(dsets).filter_unique_take().to_host(), ouniq)
    again = duniq.filter_unique_take()
    H.assert_paths_equal(again.to_host(), ouniq)
    assert empty.filter_unique_take().n_paths == 0


def test_walk_overflow_reruns(ctx):
    """A tiny path buffer forces the scratch-overflow re-run; results must not change."""
    ds = synth.make_config("tiny")
    fp = capi.FilterParams.default()
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    c2 = capi.Context(0)
    c2.set_walk_tuning(waves_per_cu=2, path_capacity=8)
    dg = H.device_graph(c2, ds, fp)
    wp = capi.WalkParams.default(rb_att=25)
    ps, _ = _walk_both(c2, dg, og, capi.METRIC_ES, wp, mc=True)
    assert ps.counters.kernel_launches >= 2
    pd, _ = _walk_both(c2, dg, og, capi.METRIC_ES, None, mc=False)


@pytest.mark.parametrize("arena_bytes,path_capacity", [(4096, 0), (4096, 8), (64 * 8, 0)])
def test_walk_arena_overflow_reruns(ctx, arena_bytes, path_capacity):
    """A tiny arena forces ST_ARENA_OVF: the paths that did fit are kept, the arena grows, the units that did not fit
    are re-run -- alone and mixed with the path-buffer overflow.  Results must not change (MC and DET, two metrics in
    one launch included)."""
    ds = synth.make_config("toy")
    fp = capi.FilterParams.default()
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    c2 = capi.Context(0)
    c2.set_walk_tuning(waves_per_cu=4, path_capacity=path_capacity)
    c2.set_arena_limit(arena_bytes)
    dg = H.device_graph(c2, ds, fp)
    wp = capi.WalkParams.default(rb_att=40)
    ps, _ = _walk_both(c2, dg, og, capi.METRIC_ES, wp, mc=True)
    assert ps.n_paths > 0 and ps.n_nodes_total * 8 > arena_bytes      # the arena really was too small
    assert ps.counters.kernel_launches >= 2
    _walk_both(c2, dg, og, capi.METRIC_OS, None, mc=False)
    a, b = dg.walk_mc2(capi.METRIC_ES, capi.METRIC_OS, wp)
    for got, metric in ((a, capi.METRIC_ES), (b, capi.METRIC_OS)):
        ops, _, ostats = og.walk_mc(metric, H.to_oracle_wp(wp), po.SEED_PER_WALK)
        H.assert_stats_equal(got.unit_stats(), ostats)
        H.assert_paths_equal(got.to_host(), ops)
    # the workspace has grown: the same call now fits in one launch, same result
    c2.set_arena_limit(0)
    again = dg.walk_mc(capi.METRIC_ES, wp)
    H.assert_paths_equal(again.to_host(), ps.to_host())


def test_layout_limits_are_refused(ctx):
    """TLM_ELIMIT: 2^31 records, 2^32 adjacency entries, degree 2^24 -- refused, never truncated."""
    import ctypes as C
    L = capi.lib()
    # records: checked from the sizes alone, before any column is touched
    dummy = torch.zeros(16, dtype=torch.int32, device="cuda")
    rs = capi._Records(1 << 31, *([dummy.data_ptr()] * 9), dummy.data_ptr(), 0)
    arr = (capi._Records * 1)(rs)
    h = C.c_void_p()
    fp = capi.FilterParams.default()
    rc = L.tlm_graph_build(ctx._h, C.byref(fp), arr, 1, 4, dummy.data_ptr(), C.byref(h))
    assert rc == -3 and b"2^31" in L.tlm_last_error(), L.tlm_last_error()
    # adjacency entries of an adopted CSR
    one = np.zeros(4, np.uint64)
    rc = L.tlm_graph_from_csr(ctx._h, 2, (1 << 32) - 64, *([one.ctypes.data] * 12), C.byref(h))
    assert rc == -3, L.tlm_last_error()
    # degree: a star whose hub holds 2^24 entries
    d = 1 << 24
    row_ptr = np.array([0, d, d + 1], np.uint64)
    nbr = np.ones(d + 1, np.uint32)
    nbr[d] = 0
    z = np.zeros(d + 1, np.uint32)
    f = np.ones(d + 1, np.float32)
    with pytest.raises(capi.TlmError, match="2\\^24"):
        ctx.graph_from_csr(row_ptr, nbr, z, z, f, f, np.array([1000, 1000], np.uint32), np.array([1, 0], np.uint8))
    # one entry less is accepted
    row_ptr = np.array([0, d - 1, d], np.uint64)
    g = ctx.graph_from_csr(row_ptr, nbr[:d].copy(), z[:d], z[:d], f[:d], f[:d], np.array([1000, 1000], np.uint32),
                           np.array([1, 0], np.uint8))
    assert g.info().max_degree == d - 1


def test_memory_is_reused_and_trimmed(ctx):
    """The library's device cache: a second build of the same input allocates nothing new; trim gives the cache back."""
    ds = synth.make_config("toy")
    fp = capi.FilterParams.default()
    c2 = capi.Context(0)
    g = H.device_graph(c2, ds, fp)
    first = g.info()
    del g
    m0 = c2.memory_info()
    g = H.device_graph(c2, ds, fp)
    m1 = c2.memory_info()
    assert m1.driver_allocs == m0.driver_allocs, "the rebuild went to hipMalloc"
    assert m1.reuses > m0.reuses
    gi = g.info()
    assert gi.build_alloc_ms == 0.0 and gi.host_syncs <= first.host_syncs
    assert abs(sum(gi.phases().values()) - gi.build_ms) < 0.25 * gi.build_ms + 1.0     # the phases cover the build
    del g
    before = c2.memory_info().cached_bytes
    assert before > 0
    c2.trim()
    m2 = c2.memory_info()
    # every WHOLLY released segment went back to the driver; what may stay is the released part of a segment another
    # owner in this process (a module-level context, a graph of another test) still holds a block of
    assert m2.cached_bytes < min(before, 64 << 20)


def test_pathset_summary_and_fetch(ctx):
    """What the packed host keeps of a path (front, back, node count, hash) and the on-demand fetch of node lists."""
    ds = synth.make_config("toy")
    dg = H.device_graph(ctx, ds, capi.FilterParams.default())
    wp = capi.WalkParams.default(rb_att=30)
    a, b = dg.walk_mc(capi.METRIC_ES, wp), dg.walk_det(capi.METRIC_OS)
    both = capi.PathSet.concat([a, b, a])           # every path of `a` occurs twice: equal sequences, equal hashes
    h = both.to_host()
    front, back, count, hsh = both.summary()
    off = h.off.astype(np.int64)
    np.testing.assert_array_equal(front, h.nodes[off[:-1]])
    np.testing.assert_array_equal(back, h.nodes[off[1:] - 1])
    np.testing.assert_array_equal(count, np.diff(off))
    seqs = [h.nodes[off[i]:off[i + 1]].tobytes() for i in range(len(h))]
    by_hash = {}
    for i, x in enumerate(hsh):
        by_hash.setdefault(int(x), set()).add(seqs[i])
    assert all(len(v) == 1 for v in by_hash.values()), "two different node sequences share a hash"
    assert len(by_hash) == len(set(seqs))
    np.testing.assert_array_equal(hsh[:a.n_paths], hsh[a.n_paths + b.n_paths:])
    pick = np.array([len(h) - 1, 0, 3, 3, a.n_paths], np.uint64)
    nodes, slots = both.fetch(pick, count)
    want_n = np.concatenate([h.nodes[off[i]:off[i + 1]] for i in pick.astype(np.int64)])
    want_s = np.concatenate([h.slots[off[i]:off[i + 1]] for i in pick.astype(np.int64)])
    np.testing.assert_array_equal(nodes, want_n)
    np.testing.assert_array_equal(slots, want_s)
    with pytest.raises(capi.TlmError):
        both.fetch(np.array([len(h)], np.uint64), np.append(count, 0))


@pytest.mark.parametrize("name", ["toy", "quirks"])
def test_graph_edges_and_stats(ctx, name):
    """tlm_graph_edges / tlm_graph_get_stats (lean hosts) against the downloaded CSR."""
    ds = H.dataset_by_name(name)
    dg = H.device_graph(ctx, ds, capi.FilterParams.default("min" if name == "quirks" else "avg"))
    c = dg.download(cold=True)
    deg = np.diff(c.row_ptr.astype(np.int64))
    rng = np.random.default_rng(3)
    nodes = rng.integers(0, c.n_nodes, size=500).astype(np.uint32)
    slots = (rng.random(500) * deg[nodes]).astype(np.uint32)
    e = dg.edges(nodes, slots)
    k = c.row_ptr[nodes].astype(np.int64) + slots
    np.testing.assert_array_equal(e["q_index"], c.nbr[k])
    np.testing.assert_array_equal(e["t_index"], nodes)
    for f, col in (("q_start", c.q_start), ("q_end", c.q_end), ("t_start", c.t_start), ("t_end", c.t_end)):
        np.testing.assert_array_equal(e[f], col[k])
    for f, col in (("overlap_score", c.os), ("sequence_identity", c.si), ("extension_score", c.es)):
        np.testing.assert_array_equal(e[f].view(np.uint32), col[k].view(np.uint32))
    np.testing.assert_array_equal(e["relative_strand"], c.strand[k])
    with pytest.raises(capi.TlmError):
        dg.edges(np.array([0], np.uint32), np.array([deg[0]], np.uint32))
    st = dg.stats()
    assert (st.anchors, st.reads) == (int(c.node_anchor.sum()), c.n_nodes - int(c.node_anchor.sum()))
    assert (st.min_len, st.max_len, st.min_con, st.max_con) == (int(c.node_len.min()), int(c.node_len.max()), int(deg.min()), int(deg.max()))
    ol = np.maximum(c.q_end - c.q_start, c.t_end - c.t_start).astype(np.int32).astype(np.int64).astype(np.uint64)
    assert (st.min_ol, st.max_ol, st.edges) == (int(ol.min()), int(ol.max()), c.n_adj // 2)
    for got, col, fn in ((st.min_os, c.os, np.min), (st.max_os, c.os, np.max), (st.min_es, c.es, np.min), (st.max_es, c.es, np.max),
                         (st.min_si, c.si, np.min), (st.max_si, c.si, np.max)):
        want = np.float32(fn(col))
        want = np.float32(max(want, 0)) if fn is np.max else want      # the reference's maxima start at 0
        assert np.float32(got) == want


def test_exchange_single_rank(ctx, monkeypatch):
    """The ABI's exchange on one GPU: world 1 is tlm_pathset_concat; with a one-rank RCCL communicator and
    TLM_EXCHANGE_SELFTEST the whole exchange path runs (RCCL loaded, sizes collective, placement, fix-up kernels)."""
    ds = synth.make_config("toy")
    c2 = capi.Context(0)
    dg = H.device_graph(c2, ds, capi.FilterParams.default())
    wp = capi.WalkParams.default(rb_att=30)
    sets = list(dg.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp))
    want = capi.PathSet.concat(sets).to_host()
    got, info = capi.PathSet.allgather(sets)
    H.assert_paths_equal(got.to_host(), want)
    assert info.bytes_sent == 0
    assert dg.anchor_range(0, 1) == (0, 3) and [dg.anchor_range(r, 2) for r in range(2)] == [(0, 2), (2, 3)]
    assert [dg.anchor_range(r, 8) for r in (0, 2, 3, 7)] == [(0, 1), (2, 3), (3, 3), (3, 3)]
    c2.dist_init(0, 1, capi.dist_unique_id())
    monkeypatch.setenv("TLM_EXCHANGE_SELFTEST", "1")
    got, info = capi.PathSet.allgather(sets)
    H.assert_paths_equal(got.to_host(), want)
    assert info.slot_bits == 16 and info.bytes_sent == 0 and info.device_ms > 0
    with pytest.raises(capi.TlmError):
        c2.dist_init(0, 1, capi.dist_unique_id())       # one communicator per context
    c2.dist_finalize()
    c2.dist_finalize()


@pytest.mark.parametrize("selftest", [False, True])
def test_lazy_exchange_single_rank(ctx, monkeypatch, selftest):
    """tlm_pathset_gather_summaries / tlm_gathered_fetch on one GPU: the summaries are those of tlm_pathset_concat's set,
    the duplicate flags are exact inside a builder block, fetched node lists equal the set's.  With a one-rank communicator
    and TLM_EXCHANGE_SELFTEST the RCCL all-gathers and broadcasts really run."""
    ds = H.quirks_dataset()                  # its contig<->contig edge makes whole runs of identical paths
    c2 = capi.Context(0)
    dg = H.device_graph(c2, ds, capi.FilterParams.default("min"))
    wp = capi.WalkParams.default(rb_att=30, bt_att=5)
    sets = list(dg.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp))
    if selftest:
        c2.dist_init(0, 1, capi.dist_unique_id())
        monkeypatch.setenv("TLM_EXCHANGE_SELFTEST", "1")
    want = capi.PathSet.concat(sets)
    h = want.to_host()
    front, back, count, hsh = want.summary()
    g = capi.Gathered(sets)
    assert g.n_paths == g.n_local_paths == len(h)
    e = g.export()
    np.testing.assert_array_equal(e["unit"], h.unit)
    np.testing.assert_array_equal(e["length"], h.length)
    for k, w in (("front", front), ("back", back), ("n_nodes", count), ("hash", hsh)):
        np.testing.assert_array_equal(e[k], w, err_msg=k)
    assert int(e["owner"].max()) == 0
    off = h.off.astype(np.int64)
    seqs = [h.nodes[off[i]:off[i + 1]].tobytes() for i in range(len(h))]
    builder = (h.unit >> np.uint64(56)).astype(np.int64)
    first = np.r_[True, builder[1:] != builder[:-1]]
    nonempty = np.array([s.n_paths for s in sets]) > 0
    np.testing.assert_array_equal(e["block_first"].astype(bool), first)
    assert int(first.sum()) == int(nonempty.sum())
    want_dup = np.array([(not first[i]) and seqs[i] == seqs[i - 1] for i in range(len(h))])
    np.testing.assert_array_equal(e["dup_of_prev"].astype(bool), want_dup)
    assert want_dup.any()
    pick = np.array([len(h) - 1, 0, 5, 5, len(h) // 2], np.uint64)
    nodes, slots = g.fetch(pick, e["n_nodes"])
    np.testing.assert_array_equal(nodes, np.concatenate([h.nodes[off[i]:off[i + 1]] for i in pick.astype(np.int64)]))
    np.testing.assert_array_equal(slots, np.concatenate([h.slots[off[i]:off[i + 1]] for i in pick.astype(np.int64)]))
    assert (g.info.bytes_sent == 0) and (g.info.device_ms > 0)


def _synthetic_csr(n_reads=400, deg=90, seed=0, wild=False):
    """A random graph handed over as CSR (tlm_graph_from_csr): degrees above one 64-lane chunk,
    optionally metrics whose double sums are NOT exact (forces the in-order serial summation)."""
    rng = np.random.default_rng(seed)
    n = n_reads + 3
    anchors = np.zeros(n, np.uint8)
    anchors[[5, n // 2, n - 2]] = 1
    rows = []
    for i in range(n):
        d = int(rng.integers(2, deg))
        nb = rng.integers(0, n, size=d)
        nb = nb[nb != i]
        rows.append(nb)
    row_ptr = np.zeros(n + 1, np.uint64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    nbr = np.concatenate(rows).astype(np.uint32)
    a = nbr.shape[0]
    if wild:
        ex = rng.integers(-40, 60, size=a)
        es = (rng.random(a).astype(np.float32) + np.float32(0.5)) * np.exp2(ex).astype(np.float32)
        os_ = (rng.random(a).astype(np.float32) + np.float32(0.5)) * np.exp2(-ex).astype(np.float32)
    else:
        es = (rng.random(a) * 20000).astype(np.float32)
        os_ = (rng.random(a) * 9000).astype(np.float32)
    q_start = rng.integers(0, 9000, size=a).astype(np.uint32)
    t_start = rng.integers(0, 6000, size=a).astype(np.uint32)
    node_len = rng.integers(20000, 90000, size=n).astype(np.uint32)
    return row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors


@pytest.mark.parametrize("wild", [False, True])
def test_walks_on_adopted_csr(ctx, wild):
    row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors = _synthetic_csr(wild=wild)
    dg = ctx.graph_from_csr(row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    og = po.Graph.from_csr(row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    gi = dg.info()
    if wild:
        assert gi.unsafe_nodes[0] > 0 and gi.unsafe_nodes[1] > 0      # the serial path is exercised
    else:
        assert gi.unsafe_nodes[0] == 0 and gi.unsafe_nodes[1] == 0
    wp = capi.WalkParams.default(rb_att=40, bt_att=4, len_thr=400_000)
    for metric in (capi.METRIC_ES, capi.METRIC_OS):
        _walk_both(ctx, dg, og, metric, wp, mc=True)
        _walk_both(ctx, dg, og, metric, None, mc=False)


def _csr_with_degrees(degrees, seed=0, zero_rows=()):
    """Node i has exactly degrees[i] list entries (multi-edges allowed); rows in `zero_rows` carry metric 0 only."""
    rng = np.random.default_rng(seed)
    n = len(degrees)
    anchors = np.zeros(n, np.uint8)
    anchors[[1, n // 3, (2 * n) // 3]] = 1
    row_ptr = np.zeros(n + 1, np.uint64)
    row_ptr[1:] = np.cumsum(degrees)
    a = int(row_ptr[-1])
    # neighbours are reads; only a few lists hold an anchor, so that walks roam instead of being forced home at once
    reads = np.nonzero(anchors == 0)[0].astype(np.uint32)
    nbr = reads[rng.integers(0, reads.shape[0], size=a)]
    owner = np.repeat(np.arange(n, dtype=np.uint32), degrees)
    clash = nbr == owner
    nbr[clash] = reads[(np.searchsorted(reads, nbr[clash]) + 1) % reads.shape[0]]
    for r in rng.choice(n, size=6, replace=False):
        if degrees[r] > 1:
            nbr[int(row_ptr[r]) + int(rng.integers(0, degrees[r]))] = np.nonzero(anchors)[0][int(rng.integers(0, 3))]
    # multiples of 1/8: every sum of a list is exact in a double, whatever its length (no serial fallback)
    es = (rng.integers(1, 160000, size=a) / 8.0).astype(np.float32)
    os_ = (rng.integers(1, 72000, size=a) / 8.0).astype(np.float32)
    for r in zero_rows:
        es[int(row_ptr[r]):int(row_ptr[r + 1])] = 0.0
        os_[int(row_ptr[r]):int(row_ptr[r + 1])] = -0.0
    q_start = rng.integers(0, 9000, size=a).astype(np.uint32)
    t_start = rng.integers(0, 8990, size=a).astype(np.uint32)
    node_len = rng.integers(20000, 90000, size=n).astype(np.uint32)
    return row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors


def test_walks_across_lane_widths(ctx):
    """List lengths on both sides of every switch of the Monte-Carlo kernel: 4 / 6 / 8 entries per lane (256, 384, 512),
    the chunked two-pass path above 512, more chunks than parked closing sums (> 4096), rows whose padded length ends
    exactly at the list, and rows whose candidates all have metric 0 (u * total == 0: first candidate wins)."""
    degrees = list(range(1, 40)) + list(range(120, 136)) + list(range(246, 300)) + list(range(370, 396)) \
        + list(range(500, 530)) + [640, 1023, 1024, 1025, 1536, 4095, 4097, 5000, 2, 3, 5, 8]
    degrees = np.array(degrees * 2, np.int64)
    zero_rows = [0, 7, 60, 80, 100, 130, 150]
    args = _csr_with_degrees(degrees, seed=11, zero_rows=zero_rows)
    dg = ctx.graph_from_csr(*args)
    og = po.Graph.from_csr(*args)
    gi = dg.info()
    assert gi.unsafe_nodes[0] == 0 and gi.unsafe_nodes[1] == 0 and gi.max_degree == 5000
    wp = capi.WalkParams.default(rb_att=150, bt_att=6, len_thr=300_000)
    for metric in (capi.METRIC_ES, capi.METRIC_OS, capi.METRIC_ES_SQRT):
        ps, _ = _walk_both(ctx, dg, og, metric, wp, mc=True)
        assert ps.unit_stats()["steps"].sum() > 2000
    _walk_both(ctx, dg, og, capi.METRIC_OS, None, mc=False)


@pytest.mark.parametrize("seed", list(range(64)))
def test_fuzz_small_graphs(ctx, ctx_full, seed):
    """Random small graphs with everything the generators above keep apart mixed together: nodes without a list,
    multi-edges, lists of one entry, several anchors next to each other or none reachable, metrics that are zero,
    negative zero, negative, huge, tiny or equal (ties in the deterministic ranks), random walk parameters."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(4, 70))
    degrees = rng.integers(0, int(rng.choice([3, 8, 70, 300])), size=n).astype(np.int64)
    anchors = (rng.random(n) < rng.choice([0.05, 0.2, 0.6])).astype(np.uint8)
    if anchors.sum() == 0:
        anchors[int(rng.integers(0, n))] = 1
    row_ptr = np.zeros(n + 1, np.uint64)
    row_ptr[1:] = np.cumsum(degrees)
    a = int(row_ptr[-1])
    nbr = rng.integers(0, n, size=a).astype(np.uint32)
    owner = np.repeat(np.arange(n, dtype=np.uint32), degrees)
    nbr = np.where(nbr == owner, (nbr + 1) % n, nbr).astype(np.uint32)
    kind = rng.choice(["ints", "ties", "wild", "zeros"])
    if kind == "ints":
        es = rng.integers(0, 5000, size=a).astype(np.float32)
        os_ = (rng.integers(0, 40000, size=a) / 16.0).astype(np.float32)
    elif kind == "ties":
        es = rng.integers(1, 4, size=a).astype(np.float32)
        os_ = np.full(a, 2.5, np.float32)
    elif kind == "wild":
        es = ((rng.random(a) - 0.2) * np.exp2(rng.integers(-30, 40, size=a))).astype(np.float32)
        os_ = (rng.random(a) * np.exp2(rng.integers(-60, 60, size=a))).astype(np.float32)
    else:
        es = np.where(rng.random(a) < 0.7, 0.0, 3.0).astype(np.float32)
        os_ = np.where(rng.random(a) < 0.5, -0.0, 0.0).astype(np.float32)
    q_start = rng.integers(0, 9000, size=a).astype(np.uint32)
    t_start = rng.integers(0, 9000, size=a).astype(np.uint32)
    node_len = rng.integers(1, 90000, size=n).astype(np.uint32)
    args = (row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    dg = ctx.graph_from_csr(*args)
    og = po.Graph.from_csr(*args)
    wp = capi.WalkParams.default(rb_att=int(rng.integers(1, 40)), bt_att=int(rng.integers(0, 8)),
                                 len_thr=int(rng.choice([20_000, 200_000, 4_700_000])))
    # Graphs without a list of more than 512 entries (and, Monte-Carlo, without an order-unsafe row) are walked by the LITE
    # instantiations of the kernels (csrc/api.hip); every fourth seed also goes through the full kernels, which these small
    # graphs would otherwise never reach on their short-list paths
    for full in ((False, True) if seed % 4 == 0 else (False,)):
        g = ctx_full.graph_from_csr(*args) if full else dg
        for metric in (capi.METRIC_ES, capi.METRIC_OS, capi.METRIC_ES_SQRT, capi.METRIC_OS_SQRT):
            _walk_both(ctx, g, og, metric, wp, mc=True)
            _walk_both(ctx, g, og, metric, None, mc=False)


@pytest.mark.parametrize("seed", list(range(8)))
def test_fuzz_deterministic_dense(ctx, seed):
    """The deterministic builder on lists long enough for its in-register selection (4 / 6 / 8 entries per lane, compact and
    8-byte rows): few distinct metric values (ties between ranks, the lowest rank shared by many entries, zeros), multi-edges
    (the pushed entry is not the ranked one), few anchors so that walks run until they have to go back (rank 0 skipped
    afterwards), negative metrics in some graphs (the general selection), lists that end inside a lane."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(150, 260))
    degrees = rng.integers(1, int(rng.choice([40, 300, 420, 600])), size=n).astype(np.int64)
    anchors = np.zeros(n, np.uint8)
    anchors[rng.choice(n, size=2, replace=False)] = 1
    degrees[anchors != 0] = rng.integers(20, 120, size=2)
    row_ptr = np.zeros(n + 1, np.uint64)
    row_ptr[1:] = np.cumsum(degrees)
    a = int(row_ptr[-1])
    nbr = rng.integers(0, n, size=a).astype(np.uint32)
    owner = np.repeat(np.arange(n, dtype=np.uint32), degrees)
    nbr = np.where(nbr == owner, (nbr + 1) % n, nbr).astype(np.uint32)
    if seed % 4 != 3:                      # an anchor among the neighbours is rare, as on generated data
        hit = anchors[nbr] != 0
        keep_hit = rng.random(a) < 0.02
        repl = np.nonzero(anchors == 0)[0][rng.integers(0, n - 2, size=a)]
        repl = np.where(repl == owner, np.nonzero(anchors == 0)[0][0 if owner[0] != np.nonzero(anchors == 0)[0][0] else 1], repl)
        nbr = np.where(hit & ~keep_hit, repl, nbr).astype(np.uint32)
        nbr = np.where(nbr == owner, (nbr + 1) % n, nbr).astype(np.uint32)
    levels = int(rng.choice([3, 12, 400, 100000]))
    es = rng.integers(0, levels, size=a).astype(np.float32)
    os_ = (rng.integers(0, levels, size=a) / 4.0).astype(np.float32)
    if seed % 4 == 2:
        os_ = np.where(rng.random(a) < 0.01, -os_, os_).astype(np.float32)
    q_start = rng.integers(0, 9000, size=a).astype(np.uint32)
    t_start = rng.integers(0, 9000, size=a).astype(np.uint32)
    node_len = rng.integers(1, 90000, size=n).astype(np.uint32)
    args = (row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    dg = ctx.graph_from_csr(*args)
    og = po.Graph.from_csr(*args)
    steps = 0
    for metric in (capi.METRIC_ES, capi.METRIC_OS, capi.METRIC_OS_SQRT):
        ps, _ = _walk_both(ctx, dg, og, metric, None, mc=False)
        steps += int(ps.unit_stats()["steps"].sum())
    assert steps > 3000


def test_rng_stream_through_long_walks(ctx):
    """More than 312 draws in one walk crosses an mt19937 twist on the device."""
    row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors = _synthetic_csr(n_reads=3000, deg=40, seed=3)
    anchors[:] = 0
    anchors[7] = 1                       # a single anchor: walks only end by length or dead end
    q_start[:] = 10                      # +5 bp per step: the length cap is never reached
    t_start[:] = 5
    dg = ctx.graph_from_csr(row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    og = po.Graph.from_csr(row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    wp = capi.WalkParams.default(rb_att=6, bt_att=2)
    ps, _ = _walk_both(ctx, dg, og, capi.METRIC_ES, wp, mc=True)
    assert ps.unit_stats()["draws"].max() > 700


def test_no_cpu_fallback_symbols():
    """The product library must not link or load the oracle."""
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    nm = subprocess.run(["nm", "-D", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "tlo_" not in nm


def test_walk_accounting_counts_layout_bytes_and_changes_nothing(ctx):
    """tlm_ctx_set_walk_accounting: the accounting builds of the walk kernels return the same Path sets and per-unit records
    and report the bytes their steps asked the layout for: per step a list of ceil(deg / K) * K entries of 6 or 8 bytes, a
    row record and (Monte-Carlo) a {q_start, t_start} pair."""
    c2 = capi.Context(0)
    dg = H.device_graph(c2, synth.make_config("dense"), capi.FilterParams.default())
    wp = capi.WalkParams.default(rb_att=20)
    plain = list(dg.walk_mc2(capi.METRIC_ES, capi.METRIC_OS, wp)) + list(dg.walk_det2(capi.METRIC_ES, capi.METRIC_OS))
    assert all(p.counters.layout_bytes == 0 for p in plain)
    c2.set_walk_accounting(True)
    counted = list(dg.walk_mc2(capi.METRIC_ES, capi.METRIC_OS, wp)) + list(dg.walk_det2(capi.METRIC_ES, capi.METRIC_OS))
    c2.set_walk_accounting(False)
    for a, b in zip(plain, counted):
        H.assert_paths_equal(a.to_host(), b.to_host())
        H.assert_stats_equal(a.unit_stats(), b.unit_stats())
    for k, extra in ((0, 16), (2, 8)):            # the launch's bytes are booked on the first set of a two-metric call
        evals = counted[k].counters.evals + counted[k + 1].counters.evals
        steps = counted[k].counters.steps + counted[k + 1].counters.steps
        lb = counted[k].counters.layout_bytes
        assert counted[k + 1].counters.layout_bytes == 0
        # MC evals stop at a forcing anchor edge, the list is read whole: evals <= entries read <= evals' lists padded by < 12
        assert lb >= 6 * evals + extra * steps
        assert lb <= 8 * (evals + 12 * steps) + extra * steps + 8 * 600 * steps


def test_mark_log_of_a_walk_that_jumps_and_backtracks(ctx, monkeypatch):
    """The mark log's worst case: ids that jump further than the LDS band at every step (no locality order:
    TLM_NO_LOCALIZE) log a far mark AND a band move per push, and every dead-end stub adds a go-back with another band move.
    A chain of 60 000 reads with a dead-end stub at each -- heavy enough that the roulette all but always tries it first --
    walked end to end with unlimited backtracking needs about 2.4 N log entries (0.86 N far marks: the band holds a seventh of
    the ids; 1.5 N band moves): more than the 2 N + 64 the re-run at full capacity used to have, within the 3 N + 64 it has now."""
    monkeypatch.setenv("TLM_NO_LOCALIZE", "1")
    rng = np.random.default_rng(3)
    chain = 60_000
    n = 2 + 2 * chain                           # two anchors, the chain, one stub per chain node
    ids = rng.permutation(n)                    # random ids: consecutive chain nodes lie far apart
    a0, a1 = int(ids[0]), int(ids[1])
    cid = ids[2:2 + chain]
    sid = ids[2 + chain:]
    rows = [[] for _ in range(n)]
    rows[a0] = [int(cid[0])]
    for k in range(chain):
        nxt = int(cid[k + 1]) if k + 1 < chain else a1
        rows[int(cid[k])] = [int(sid[k]), nxt]  # the stub first: the roulette meets it with a heavy weight
        rows[int(sid[k])] = [int(cid[k])]       # its only neighbour is visited: a dead end
    rows[a1] = [int(cid[-1])]
    row_ptr = np.zeros(n + 1, np.uint64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    nbr = np.concatenate([np.array(r, np.uint32) for r in rows])
    a = nbr.shape[0]
    es = np.ones(a, np.float32)
    stub_entry = row_ptr[cid].astype(np.int64)              # the stub is the first entry of every chain node's list
    es[stub_entry] = 1.0e6
    os_ = es.copy()
    q_start = np.full(a, 10, np.uint32)
    t_start = np.full(a, 3, np.uint32)
    node_len = np.full(n, 1000, np.uint32)
    anchors = np.zeros(n, np.uint8)
    anchors[[a0, a1]] = 1
    c2 = capi.Context(0)
    dg = c2.graph_from_csr(row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    wp = capi.WalkParams.default(rb_att=6, bt_att=1_000_000, len_thr=1 << 40)
    og = po.Graph.from_csr(row_ptr, nbr, q_start, t_start, os_, es, node_len, anchors)
    ps = dg.walk_mc(capi.METRIC_ES, wp)
    ops, _, ostats = og.walk_mc(po.METRIC_ES, H.to_oracle_wp(wp), po.SEED_PER_WALK)
    H.assert_stats_equal(ps.unit_stats(), ostats)
    H.assert_paths_equal(ps.to_host(), ops)
    st = ps.unit_stats()
    assert ps.counters.kernel_launches >= 2, "the walks were meant to outgrow the default path buffer"
    assert (st["status"] == 1).any() and int(st["steps"].max()) > 2 * chain      # chain + stubs + the scans after each go-back


def test_pool_splits_and_merges(ctx):
    """The library's device cache hands out PARTS of released blocks (round 3) and merges released neighbours again: graphs
    of different sizes built and dropped in changing order settle on the memory the first round obtained (no further
    driver allocation), their walks stay the oracle's, and when everything has been released a trim returns every byte."""
    c2 = capi.Context(0)
    fp = capi.FilterParams.default()
    sets = {name: synth.make_config(name) for name in ("tiny", "toy", "dense")}
    wp = capi.WalkParams.default(rb_att=12)
    want = {}
    allocs = []
    for rnd, order in enumerate((("toy", "tiny", "dense"), ("dense", "toy", "tiny"), ("tiny", "dense", "toy"), ("toy", "tiny", "dense"))):
        graphs = {}
        for name in order:
            graphs[name] = H.device_graph(c2, sets[name], fp)
            ps = graphs[name].walk_mc(capi.METRIC_ES, wp)
            got = (ps.to_host(), ps.unit_stats())
            if name not in want:
                og = H.oracle_graph(sets[name], H.to_oracle_fp(fp))
                ops, _, ostats = og.walk_mc(po.METRIC_ES, H.to_oracle_wp(wp), po.SEED_PER_WALK)
                want[name] = (ops, ostats)
            H.assert_paths_equal(got[0], want[name][0])
            H.assert_stats_equal(got[1], want[name][1])
            del ps
            if name == order[0]:
                del graphs[name]            # one graph goes while the others are still being built
        graphs.clear()
        allocs.append(c2.memory_info().driver_allocs)
    # a new ORDER of sizes may still need a block the earlier orders did not; an order seen before needs none
    assert allocs[3] == allocs[2] and allocs[3] - allocs[0] <= 16, allocs
    mi = c2.memory_info()
    assert mi.live_bytes < (64 << 20)        # nothing but the context's own small buffers
    c2.trim()
    assert c2.memory_info().cached_bytes < (64 << 20)      # (0 in a process of its own: see test_memory_is_reused_and_trimmed)
