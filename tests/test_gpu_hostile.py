"""The hostile generator (telomeri_amd.synth.make_hard_dataset) on the GPU: what minimap2 on real reads
brings and the friendly generator does not -- repeat families (hub lists of 10^3..10^4 entries, neighbours megabases away:
8-byte rows, probes of the HBM visited map, the chunked selections), log-normal reads, contigs over two decades of length,
chimeric reads, contig <-> contig records, multi-edges.

* hard_tiny: every unit of all four builders, all four metrics, bit for bit against the oracle (which
  tests/test_oracle_vs_reference.py::test_hostile_datasets pins to the unmodified reference on the same generator);
* hard_small (50 k reads, lists up to ~8 700 entries): the graph bit for bit; Monte-Carlo trials of every anchor and a
  spread of deterministic units against the oracle; every accepted path checked against the CSR;
* athaliana_hard (C4's size): graph invariants, a bundle through the one-call builders with every path checked, a spread
  of units of all four builders against the oracle.
"""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import pyoracle as po
from telomeri_amd import capi, synth
from test_gpu_full_size import _check_paths, _graph_checksums, _oracle_sample

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return capi.Context(0)


@pytest.mark.parametrize("seed,mode,rb,bt", [(21, "avg", 60, 60), (33, "avg", 40, 4), (35, "min", 30, 60)])
def test_hard_tiny_every_unit(ctx, seed, mode, rb, bt):
    ds = synth.make_config("hard_tiny", seed=seed)
    fp = capi.FilterParams.default(mode)
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    H.assert_csr_equal(dg.download(), og.csr())
    wp = capi.WalkParams.default(rb_att=rb, bt_att=bt)
    for metric in (capi.METRIC_ES, capi.METRIC_OS, capi.METRIC_ES_SQRT, capi.METRIC_OS_SQRT):
        ps = dg.walk_mc(metric, wp)
        ops, octr, ostats = og.walk_mc(metric, H.to_oracle_wp(wp), po.SEED_PER_WALK)
        H.assert_stats_equal(ps.unit_stats(), ostats)
        H.assert_paths_equal(ps.to_host(), ops)
        pd = dg.walk_det(metric)
        opd, _, dstats = og.walk_det(metric)
        H.assert_stats_equal(pd.unit_stats(), dstats)
        H.assert_paths_equal(pd.to_host(), opd)


def test_hard_small_hubs(ctx):
    ds = synth.make_config("hard_small")
    fp = capi.FilterParams.default()
    dg = H.device_graph(ctx, ds, fp)
    og = H.oracle_graph(ds, H.to_oracle_fp(fp))
    csr = dg.download()
    H.assert_csr_equal(csr, og.csr())
    gi = dg.info()
    deg = np.diff(csr.row_ptr.astype(np.int64))
    assert gi.max_degree == deg.max() > 4096 and (deg > 768).sum() > 1000, "this config is here for its hub lists"
    assert gi.unsafe_nodes[0] == 0 and gi.unsafe_nodes[1] == 0
    rb = 100
    wp = capi.WalkParams.default(rb_att=rb)
    anchors, det_off = dg.anchor_table()
    mc_anchor = lambda u: anchors[(u // rb).astype(np.int64)]
    det_anchor = lambda u: anchors[np.searchsorted(det_off, u, side="right") - 1]
    hub_steps = 0
    for metric in (capi.METRIC_ES, capi.METRIC_OS):
        ps = dg.walk_mc(metric, wp)                                 # every anchor, 100 trials each
        ops, octr, ostats = og.walk_mc(metric, H.to_oracle_wp(wp), po.SEED_PER_WALK, threads=8)
        H.assert_stats_equal(ps.unit_stats(), ostats)
        h = ps.to_host()
        H.assert_paths_equal(h, ops)
        assert ps.counters.evals == octr.evals
        _check_paths(h, csr, mc_anchor, mc=True)
        hub_steps += int((deg[h.nodes] > 768).sum())
        # deterministic walks: every path checked, three stretches of units against the oracle (the whole set is half a
        # minute of one oracle thread per metric)
        pd = dg.walk_det(metric)
        hd = pd.to_host()
        if len(hd):
            _check_paths(hd, csr, det_anchor, mc=False)
        n = pd.n_units
        st = pd.unit_stats()
        for u0 in (0, int(det_off[1]), n // 2, n - 96):             # (the second anchor owns a hub list of its own)
            ops, _, ostats = og.walk_det(metric, u0, u0 + 96, threads=8)
            sub = dg.walk_det(metric, u0, u0 + 96)
            H.assert_stats_equal(sub.unit_stats(), ostats)
            H.assert_paths_equal(sub.to_host(), ops)
            np.testing.assert_array_equal(sub.unit_stats(), st[u0:u0 + 96])
    assert hub_steps > 0, "no accepted Monte-Carlo path crossed a hub"


def test_hard_small_accounting_and_lite_choice(ctx):
    """The hostile shape takes the kernels' other branches: 8-byte rows and chunked long lists carry a visible share of the
    steps (the friendly configs: 8 % and 0 %)."""
    ds = synth.make_config("hard_small")
    dg = H.device_graph(ctx, ds, capi.FilterParams.default())
    wp = capi.WalkParams.default(rb_att=60)
    plain = dg.walk_mc(capi.METRIC_ES, wp)
    ctx.set_walk_accounting(True)
    try:
        acct = dg.walk_mc(capi.METRIC_ES, wp)
    finally:
        ctx.set_walk_accounting(False)
    np.testing.assert_array_equal(plain.unit_stats(), acct.unit_stats())
    H.assert_paths_equal(plain.to_host(), acct.to_host())
    cls = np.array([int(x) for x in acct.counters.step_class], dtype=np.float64)
    assert cls.sum() == acct.counters.steps
    share = cls / cls.sum()
    assert share[4:].sum() > 0.10, share        # classes 0-3: compact rows; 4-7: 8-byte rows; 8: chunked long lists; 9: serial


def test_athaliana_hard_bundle_walks(ctx):
    """C4's size, hostile: ~1 M reads, 200 contigs of 20 kb .. 2 Mb, 3 % of the genome in 10..50-copy repeat families."""
    ds = synth.make_config("athaliana_hard", device="cuda")
    fp = capi.FilterParams.default()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    n_contigs, n_records = ds.n_contigs, len(ds.rr) + len(ds.cr)
    del ds
    torch.cuda.empty_cache()
    gi = g.info()
    assert gi.n_records == n_records and gi.n_adj == 2 * gi.n_kept and gi.n_anchors == n_contigs == 200
    assert gi.max_degree > 4096, "this config is here for its hub lists"
    assert gi.unsafe_nodes[0] == 0 and gi.unsafe_nodes[1] == 0
    csr = g.download(cold=False)
    assert int(np.diff(csr.row_ptr.astype(np.int64)).max()) == gi.max_degree
    _graph_checksums(csr)
    rb = 2000
    wp = capi.WalkParams.default(rb_att=rb)
    anchors, det_off = g.anchor_table()
    a0, a1 = 96, 104                      # eight anchors from the middle of the genome
    d0, d1 = int(det_off[a0]), int(det_off[a1])
    sets = g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp, (a0 * rb, a1 * rb), (d0, d1))
    mc_anchor = lambda u: anchors[(u // rb).astype(np.int64)]
    det_anchor = lambda u: anchors[np.searchsorted(det_off, u, side="right") - 1]
    deg = np.diff(csr.row_ptr.astype(np.int64))
    hub_steps = 0
    for k, ps in enumerate(sets):
        st = ps.unit_stats()
        c = ps.counters
        assert (c.steps, c.evals, c.draws) == (int(st["steps"].sum()), int(st["evals"].sum()), int(st["draws"].sum()))
        assert c.accepted == int((st["status"] == 1).sum()) == ps.n_paths
        assert ps.n_units == ((a1 - a0) * rb if k < 2 else d1 - d0)
        h = ps.to_host()
        if len(h):
            _check_paths(h, csr, mc_anchor if k < 2 else det_anchor, mc=k < 2)
            hub_steps += int((deg[h.nodes] > 768).sum())
        del h
    assert sets[0].n_paths > 100 and hub_steps > 0
    og = po.Graph.from_csr(csr.row_ptr, csr.nbr, csr.q_start, csr.t_start, csr.os, csr.es, csr.node_len, csr.node_anchor)
    mc_units = [(a * rb + r, a * rb + r + 40) for a, r in ((a0, 0), (a0 + 3, 1230), (a1 - 1, rb - 40), (17, 500), (180, 40))]
    nd = int(det_off[-1])
    det_units = [(d0, d0 + 24), (d1 - 24, d1), (0, 24), (nd // 3, nd // 3 + 24), (nd - 24, nd)]
    _oracle_sample(g, og, csr, wp, mc_units, det_units)
    sub = g.walk_mc(capi.METRIC_OS, wp, (a0 + 3) * rb, (a0 + 3) * rb + 64)
    np.testing.assert_array_equal(sub.unit_stats(), sets[1].unit_stats()[3 * rb:3 * rb + 64])
