"""Parity at BASELINE.json's sizes, through properties that do not need the oracle to walk everything
(the oracle would take hours on the A. thaliana-shaped graph):

* every accepted path is a real walk of the graph: consecutive nodes are joined by the reported edge slot,
  starts at the unit's anchor, ends at another anchor, visits no node twice (Monte-Carlo), and its reported
  length equals Path::updateLength recomputed on the host from the downloaded CSR;
* counters are sums of the per-unit records; unit ranges concatenate (the sharding contract);
* a sample of units is walked by the oracle on the SAME graph (copied to the host) and must agree per unit
  in status, nodes, steps, evals, draws, length -- and path for path.
The graph itself (150 M records at the largest size) is checked against the oracle's filter/score on a
random sample of records and through CSR invariants (symmetry of the two entries of a record).
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


def _check_paths(h: capi.HostPaths, csr: capi.HostCsr, anchors_of_unit, mc: bool) -> None:
    n = len(h)
    assert np.all(np.diff(h.unit.astype(np.int64)) > 0)
    lens = np.diff(h.off.astype(np.int64))
    assert lens.min() >= 2
    first = h.nodes[h.off[:-1].astype(np.int64)]
    last = h.nodes[h.off[1:].astype(np.int64) - 1]
    np.testing.assert_array_equal(first, anchors_of_unit(h.unit))
    assert np.all(csr.node_anchor[first] == 1) and np.all(csr.node_anchor[last] == 1)
    assert np.all(first != last)
    # continuity: nodes[k+1] == nbr[row_ptr[nodes[k]] + slots[k]]
    is_last = np.zeros(h.nodes.shape[0], bool)
    is_last[h.off[1:].astype(np.int64) - 1] = True
    k = np.nonzero(~is_last)[0]
    e = csr.row_ptr[h.nodes[k]].astype(np.int64) + h.slots[k].astype(np.int64)
    assert np.all(h.slots[k] < (csr.row_ptr[h.nodes[k] + 1] - csr.row_ptr[h.nodes[k]]).astype(np.int64))
    np.testing.assert_array_equal(csr.nbr[e], h.nodes[k + 1])
    assert np.all(h.slots[is_last] == 0xFFFFFFFF)
    # Path::updateLength (Path.cpp:6-16): first node's length + increments of all edges but the last
    delta = csr.q_start[e].astype(np.int64) - csr.t_start[e].astype(np.int64)
    path_of = np.repeat(np.arange(n), lens)[k]
    pos_in = k - h.off[:-1].astype(np.int64)[path_of]
    counted = pos_in < (lens[path_of] - 2)              # edge j counts iff j < m-1, m = lens-1 edges
    want = csr.node_len[first].astype(np.int64) + np.bincount(path_of[counted], weights=delta[counted], minlength=n).astype(np.int64)
    np.testing.assert_array_equal(h.length, want)
    assert np.all(h.length >= 0)
    if mc:      # the visited set: no node twice on a Monte-Carlo path
        for i in np.random.default_rng(0).choice(n, size=min(n, 200), replace=False):
            seg = h.nodes[int(h.off[i]):int(h.off[i + 1])]
            assert np.unique(seg).shape[0] == seg.shape[0]


@pytest.mark.parametrize("config,rb,sample_units", [("ecoli", 24, 12), ("yeast", 6, 12)])
def test_walks_at_size(ctx, config, rb, sample_units):
    ds = synth.make_config(config, device="cuda")
    fp = capi.FilterParams.default()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    gi = g.info()
    assert gi.n_kept * 2 == gi.n_adj and gi.unsafe_nodes[0] == 0 and gi.unsafe_nodes[1] == 0
    csr = g.download(cold=False)
    anchors, det_off = g.anchor_table()
    assert anchors.shape[0] == ds.n_contigs
    wp = capi.WalkParams.default(rb_att=rb)
    og = po.Graph.from_csr(csr.row_ptr, csr.nbr, csr.q_start, csr.t_start, csr.os, csr.es, csr.node_len, csr.node_anchor)
    for metric in (capi.METRIC_ES, capi.METRIC_OS):
        ps = g.walk_mc(metric, wp)
        st = ps.unit_stats()
        c = ps.counters
        assert (c.steps, c.evals, c.draws) == (int(st["steps"].sum()), int(st["evals"].sum()), int(st["draws"].sum()))
        assert c.accepted == int((st["status"] == 1).sum()) == ps.n_paths
        _check_paths(ps.to_host(), csr, lambda u: anchors[(u // rb).astype(np.int64)], mc=True)
        # the oracle on a sample of units of the same graph
        total = g.mc_units(wp)
        u0 = (total // 2 // sample_units) * sample_units
        sub = g.walk_mc(metric, wp, u0, u0 + sample_units)
        ops, octr, ostats = og.walk_mc(metric, po.WalkParams(rb, wp.backtrack_attempts, wp.len_threshold), po.SEED_PER_WALK,
                                       u0, u0 + sample_units, threads=4)
        H.assert_stats_equal(sub.unit_stats(), ostats)
        H.assert_paths_equal(sub.to_host(), ops)
        np.testing.assert_array_equal(sub.unit_stats(), st[u0:u0 + sample_units])       # range == slice of the whole
    pd = g.walk_det(capi.METRIC_ES)
    hd = pd.to_host()
    if len(hd):
        unit_anchor = lambda u: anchors[np.searchsorted(det_off, u, side="right") - 1]
        _check_paths(hd, csr, unit_anchor, mc=False)
    dsub = g.walk_det(capi.METRIC_ES, 0, 16)
    ops, _, ostats = og.walk_det(capi.METRIC_ES, 0, 16, threads=4)
    H.assert_stats_equal(dsub.unit_stats(), ostats)
    H.assert_paths_equal(dsub.to_host(), ops)


def test_graph_at_full_size(ctx):
    """A. thaliana-shaped input (150.8 M records): filter/score bits on a record sample, CSR invariants."""
    ds = synth.make_config("athaliana", device="cuda")
    fp = capi.FilterParams.default()
    keep, si, os_, es, nk = ctx.filter_score(fp, ds.rr)
    idx = torch.randint(0, len(ds.rr), (200_000,), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    sample = ds.rr.take(idx)
    cols = [getattr(sample, c).cpu().numpy() for c in H.REC_COLS[:7]]
    okeep, osi, oos, oes = po.filter_score(H.to_oracle_fp(fp), *cols, (sample.q_id == sample.t_id).cpu().numpy().astype(np.uint8))
    np.testing.assert_array_equal(keep[idx].cpu().numpy(), okeep)
    for dev, ora in ((si, osi), (os_, oos), (es, oes)):
        np.testing.assert_array_equal(dev[idx].cpu().numpy().view(np.uint32), ora.view(np.uint32))
    assert nk == int(keep.sum().item())
    del keep, si, os_, es
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    gi = g.info()
    assert gi.n_anchors == ds.n_contigs and gi.n_kept_per_file[0] == nk
    assert gi.n_adj == 2 * gi.n_kept
    csr = g.download(cold=True)
    deg = np.diff(csr.row_ptr.astype(np.int64))
    assert deg.min() >= 1 and deg.max() == gi.max_degree
    # every record contributes one entry to each endpoint with the same scores: multiset of (min, max, es bits)
    # over owner<nbr entries equals the one over owner>nbr entries
    owner = np.repeat(np.arange(csr.n_nodes, dtype=np.uint32), deg)
    lo = np.minimum(owner, csr.nbr).astype(np.uint64)
    hi = np.maximum(owner, csr.nbr).astype(np.uint64)
    key = (lo * np.uint64(csr.n_nodes) + hi) * np.uint64(1 << 20) ^ csr.es.view(np.uint32).astype(np.uint64)
    a = np.sort(key[owner < csr.nbr])
    b = np.sort(key[owner > csr.nbr])
    np.testing.assert_array_equal(a, b)
    # swapped coordinates between the two entries of a record: sum of q_start over all == sum of t_start over all
    assert int(csr.q_start.astype(np.int64).sum()) == int(csr.t_start.astype(np.int64).sum())
    assert int(csr.q_end.astype(np.int64).sum()) == int(csr.t_end.astype(np.int64).sum())
    # anchors are the contigs: targets of the cr file that survive the filter
    assert int(csr.node_anchor.sum()) == ds.n_contigs


def test_pack_roundtrip_on_device(ctx):
    from telomeri_amd import pathgather
    ds = synth.make_config("tiny")
    g = H.device_graph(ctx, ds, capi.FilterParams.default())
    wp = capi.WalkParams.default(rb_att=20)
    sets = [pathgather.PackedPaths(*g.walk_mc(m, wp).to_device_tensors("cuda")) for m in (0, 1)]
    sets.append(pathgather.PackedPaths(*g.walk_det(0).to_device_tensors("cuda")))
    sizes, payload = pathgather.pack(sets)
    back = pathgather.unpack(sizes, payload)
    for x, y in zip(back, sets):
        for f in ("unit", "length", "off", "nodes", "slots"):
            assert torch.equal(getattr(x, f), getattr(y, f)), f
    assert pathgather.allgather_paths(sets)[0].n_paths == sets[0].n_paths      # world size 1: identity
