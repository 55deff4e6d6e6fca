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


def _graph_checksums(csr) -> None:
    """Order-independent invariants of a CSR too large to sort on the host: every kept record owns one entry in each
    endpoint's list with the same scores and swapped coordinates (OverlapGraph.cpp:241-248), so a wrapping sum of a
    mixed key over the entries with owner < neighbour equals the one over owner > neighbour."""
    deg = np.diff(csr.row_ptr.astype(np.int64))
    assert deg.min() >= 1
    owner = np.repeat(np.arange(csr.n_nodes, dtype=np.uint32), deg)
    lo = np.minimum(owner, csr.nbr).astype(np.uint64)
    hi = np.maximum(owner, csr.nbr).astype(np.uint64)
    with np.errstate(over="ignore"):
        key = (lo * np.uint64(0x9E3779B97F4A7C15) + hi) * np.uint64(0xBF58476D1CE4E5B9)
        key ^= csr.es.view(np.uint32).astype(np.uint64) * np.uint64(0x94D049BB133111EB)
        key ^= key >> np.uint64(29)
        key *= np.uint64(0xD6E8FEB86659FD93)
        up = owner < csr.nbr
        assert int(key[up].sum(dtype=np.uint64)) == int(key[~up & (owner != csr.nbr)].sum(dtype=np.uint64))
    assert int(up.sum()) * 2 + int((owner == csr.nbr).sum()) == csr.n_adj
    assert int(csr.q_start.astype(np.int64).sum()) == int(csr.t_start.astype(np.int64).sum())


def _oracle_sample(g, og, csr, wp, mc_units, det_units):
    """Oracle parity on a spread of units: MC under ES and OS (per-walk seeds), DET under ES and OS."""
    owp = po.WalkParams(wp.rebuild_attempts, wp.backtrack_attempts, wp.len_threshold)
    for metric in (capi.METRIC_ES, capi.METRIC_OS):
        for u0, u1 in mc_units:
            sub = g.walk_mc(metric, wp, u0, u1)
            ops, _, ostats = og.walk_mc(metric, owp, po.SEED_PER_WALK, u0, u1, threads=8)
            H.assert_stats_equal(sub.unit_stats(), ostats)
            H.assert_paths_equal(sub.to_host(), ops)
        for u0, u1 in det_units:
            sub = g.walk_det(metric, u0, u1)
            ops, _, ostats = og.walk_det(metric, u0, u1, threads=8)
            H.assert_stats_equal(sub.unit_stats(), ostats)
            H.assert_paths_equal(sub.to_host(), ops)


def test_athaliana_bundle_walks(ctx):
    """BASELINE config 4 (A. thaliana-shaped, 150.8 M records): one 25-anchor bundle (= one GPU's share of the 8-GPU
    split) through the four builders of Main.cpp:237-243 in ONE call, exactly as bench.py runs it; every path checked
    against the downloaded CSR, a spread of units of all four builders against the oracle."""
    ds = synth.make_config("athaliana", device="cuda")
    fp = capi.FilterParams.default()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    n_contigs = ds.n_contigs
    del ds
    torch.cuda.empty_cache()
    gi = g.info()
    assert gi.n_anchors == n_contigs == 200 and gi.unsafe_nodes[0] == 0 and gi.unsafe_nodes[1] == 0
    rb = 2000
    wp = capi.WalkParams.default(rb_att=rb)
    anchors, det_off = g.anchor_table()
    a0, a1 = 75, 100                      # a bundle from the middle of the genome
    d0, d1 = int(det_off[a0]), int(det_off[a1])
    sets = g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp, (a0 * rb, a1 * rb), (d0, d1))
    csr = g.download(cold=False)
    mc_anchor = lambda u: anchors[(u // rb).astype(np.int64)]
    det_anchor = lambda u: anchors[np.searchsorted(det_off, u, side="right") - 1]
    evals = 0
    for k, ps in enumerate(sets):
        st = ps.unit_stats()
        c = ps.counters
        assert (c.steps, c.evals, c.draws) == (int(st["steps"].sum()), int(st["evals"].sum()), int(st["draws"].sum()))
        assert c.accepted == int((st["status"] == 1).sum()) == ps.n_paths
        assert ps.n_units == ((a1 - a0) * rb if k < 2 else d1 - d0)
        evals += c.evals
        h = ps.to_host()
        if len(h):
            _check_paths(h, csr, mc_anchor if k < 2 else det_anchor, mc=k < 2)
        del h
    assert sets[0].n_paths > 1000 and sets[1].n_paths > 1000 and evals > 5e10
    og = po.Graph.from_csr(csr.row_ptr, csr.nbr, csr.q_start, csr.t_start, csr.os, csr.es, csr.node_len, csr.node_anchor)
    mc_units = [(a * rb + r, a * rb + r + 40) for a, r in ((a0, 0), (a0 + 7, 1230), (a1 - 1, rb - 40))]
    det_units = [(d0, d0 + 24), ((d0 + d1) // 2, (d0 + d1) // 2 + 24), (d1 - 24, d1)]
    _oracle_sample(g, og, csr, wp, mc_units, det_units)
    # the bundle equals the same units of separate calls (the sharding contract at full size)
    sub = g.walk_mc(capi.METRIC_OS, wp, (a0 + 3) * rb, (a0 + 3) * rb + 64)
    np.testing.assert_array_equal(sub.unit_stats(), sets[1].unit_stats()[3 * rb:3 * rb + 64])


def test_chr1_graph_and_walks(ctx):
    """BASELINE config 5 (human chr1-shaped: 614 M records, 936 M adjacency entries = 22 % of the 2^32 layout limit,
    degrees above 512 -> the chunked selection path): graph invariants, then the four builders on a few anchors with
    every path checked and a spread of units compared with the oracle."""
    ds = synth.make_config("chr1", device="cuda")
    fp = capi.FilterParams.default()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    n_contigs, n_records = ds.n_contigs, len(ds.rr) + len(ds.cr)
    del ds
    torch.cuda.empty_cache()
    gi = g.info()
    assert gi.n_records == n_records and gi.n_adj == 2 * gi.n_kept and gi.n_anchors == n_contigs == 500
    assert gi.max_degree > 512, "this config is here for the lists longer than 512 entries"
    assert gi.unsafe_nodes[0] == 0 and gi.unsafe_nodes[1] == 0
    csr = g.download(cold=False)
    assert int(np.diff(csr.row_ptr.astype(np.int64)).max()) == gi.max_degree
    _graph_checksums(csr)
    assert int(csr.node_anchor.sum()) == n_contigs
    rb = 2000
    wp = capi.WalkParams.default(rb_att=rb)
    anchors, det_off = g.anchor_table()
    a0, a1 = 242, 258                     # a 16-anchor bundle (3 anchors in round 3; 40 and 24 ran in round 4: profiles/probes/chain_r04x.txt, _r04ar)
    d0, d1 = int(det_off[a0]), int(det_off[a1])
    sets = g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp, (a0 * rb, a1 * rb), (d0, d1))
    mc_anchor = lambda u: anchors[(u // rb).astype(np.int64)]
    det_anchor = lambda u: anchors[np.searchsorted(det_off, u, side="right") - 1]
    long_steps = 0
    for k, ps in enumerate(sets):
        st = ps.unit_stats()
        assert ps.counters.evals == int(st["evals"].sum()) and ps.counters.accepted == ps.n_paths
        h = ps.to_host()
        if len(h):
            _check_paths(h, csr, mc_anchor if k < 2 else det_anchor, mc=k < 2)
            deg = np.diff(csr.row_ptr.astype(np.int64))
            long_steps += int((deg[h.nodes] > 512).sum())
        del h
    assert sets[0].n_paths > 100 and long_steps > 0, "no accepted path crossed a node with more than 512 neighbours"
    og = po.Graph.from_csr(csr.row_ptr, csr.nbr, csr.q_start, csr.t_start, csr.os, csr.es, csr.node_len, csr.node_anchor)
    # EVERY unit of the bundle, all four builders, against the oracle: 64 000 Monte-Carlo + ~7 500 deterministic walks,
    # ~4.6e10 evals -- a quarter of a minute of the oracle on the box's CPU share (the whole GPU tier stays near nine minutes)
    from telomeri_amd import cpu_share
    threads = max(1, min(16, cpu_share.cpu_share()))
    owp = po.WalkParams(wp.rebuild_attempts, wp.backtrack_attempts, wp.len_threshold)
    evals = 0
    for k, (kind, metric) in enumerate((("mc", po.METRIC_ES), ("mc", po.METRIC_OS), ("det", po.METRIC_ES), ("det", po.METRIC_OS))):
        ops, octr, ostats = (og.walk_mc(metric, owp, po.SEED_PER_WALK, a0 * rb, a1 * rb, threads=threads) if kind == "mc"
                             else og.walk_det(metric, d0, d1, threads=threads))
        H.assert_stats_equal(sets[k].unit_stats(), ostats)
        assert sets[k].counters.evals == octr.evals
        H.assert_paths_equal(sets[k].to_host(), ops)
        evals += int(octr.evals)
    assert evals > 2e10


def test_yeast_full_trials(ctx):
    """BASELINE config 3 (S. cerevisiae-shaped) with the config's 1000 Monte-Carlo trials per anchor: ALL trials of two
    anchors compared with the oracle unit for unit and path for path, under both metrics."""
    ds = synth.make_config("yeast", device="cuda")
    fp = capi.FilterParams.default()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    del ds
    csr = g.download(cold=False)
    rb = 1000
    wp = capi.WalkParams.default(rb_att=rb)
    og = po.Graph.from_csr(csr.row_ptr, csr.nbr, csr.q_start, csr.t_start, csr.os, csr.es, csr.node_len, csr.node_anchor)
    owp = po.WalkParams(rb, wp.backtrack_attempts, wp.len_threshold)
    a, b = g.walk_mc2(capi.METRIC_ES, capi.METRIC_OS, wp, 11 * rb, 13 * rb)
    for got, metric in ((a, capi.METRIC_ES), (b, capi.METRIC_OS)):
        ops, octr, ostats = og.walk_mc(metric, owp, po.SEED_PER_WALK, 11 * rb, 13 * rb, threads=8)
        H.assert_stats_equal(got.unit_stats(), ostats)
        H.assert_paths_equal(got.to_host(), ops)
        assert got.counters.evals == octr.evals and got.n_paths == octr.accepted
    # and the whole config (30 anchors x 1000 trials x 2 metrics + DET) in one builders call, every path checked
    anchors, det_off = g.anchor_table()
    sets = g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp)
    mc_anchor = lambda u: anchors[(u // rb).astype(np.int64)]
    det_anchor = lambda u: anchors[np.searchsorted(det_off, u, side="right") - 1]
    for k, ps in enumerate(sets):
        h = ps.to_host()
        if len(h):
            _check_paths(h, csr, mc_anchor if k < 2 else det_anchor, mc=k < 2)
    np.testing.assert_array_equal(sets[0].unit_stats()[11 * rb:13 * rb], a.unit_stats())


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


def test_full_config_scaffold_parity(ctx):
    """The WHOLE A. thaliana-shaped config, once, end to end (reference surface Main.cpp:212-388):
      * every unit of all four builders (800 000 Monte-Carlo + ~97 000 deterministic walks, 6.6e11 evals) is walked by the
        oracle on the same graph (all host threads of the box's CPU share, about two minutes) and must agree with the device unit for
        unit in status, nodes, steps, evals, draws, length, and path for path (nodes and edge slots);
      * the product program `telomeri_hip` reads the same config as TEXT (26 GB: PAF + FASTA) and writes a scaffold; the
        expected scaffold is what the host stages (`telomeri_post --binary`, validated against the unmodified reference at small
        sizes in test_host_post_stages.py) make of the ORACLE's paths: same bytes."""
    import hashlib
    import os
    import shutil
    import subprocess
    import time
    from telomeri_amd import cpu_share, scaffold_bench as sb
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary, post = [os.path.join(root, "telomeri_amd", "bin", b) for b in ("telomeri_hip", "telomeri_post")]
    threads = max(1, min(16, cpu_share.cpu_share()))
    t_start = time.time()
    ds = synth.make_config("athaliana", device="cuda")
    g = ctx.build_graph(capi.FilterParams.default(), [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    wp = capi.WalkParams.default()                  # 2000 trials, 60 backtracks, 4.7 Mb: the reference's defaults
    tmp = sb.scratch_dir(40 << 30)
    try:
        # ---- the text the product reads (written while the dataset is on the device), then the product itself
        rr, cr = os.path.join(tmp, "rr_overlaps.paf"), os.path.join(tmp, "cr_overlaps.paf")
        reads, contigs = os.path.join(tmp, "reads.fasta"), os.path.join(tmp, "contigs.fasta")
        sb.write_paf(ds, ds.rr, rr, threads=threads)
        sb.write_paf(ds, ds.cr, cr, threads=threads)
        sb.write_fasta(ds, reads, contigs)
        n_reads = ds.n_reads
        names_of = lambda i: "read%07d" % (i + 1) if i < n_reads else "ctg%d" % (i - n_reads + 1)       # synth.Dataset.name_of
        assert names_of(0) == ds.name_of(0) and names_of(n_reads) == ds.name_of(n_reads)
        # the records as host columns, for the oracle's own graph build below (6 GB)
        host_ctg = ds.name_is_ctg().numpy().astype(np.uint8)
        cols = lambda rec: [rec.q_id.cpu().numpy(), rec.t_id.cpu().numpy()] + [getattr(rec, c).cpu().numpy() for c in H.REC_COLS]
        host_rr, host_cr = cols(ds.rr), cols(ds.cr)
        del ds
        torch.cuda.empty_cache()
        # ---- device: all four builders over all units
        sets = list(g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp))
        csr = g.download(cold=True)
        # ---- a4 at full size: the ORACLE's own graph, built from the 150.8 M records by its own filter,
        # first-seen numbering and adjacency insertion (tlo_graph_add_records), equals the device's CSR bit for bit -- the
        # oracle below then walks ITS graph, not a copy of the device's
        t_graph = time.time()
        og = po.Graph()
        og.add_records(False, H.to_oracle_fp(capi.FilterParams.default()), host_ctg, *host_rr)
        og.add_records(True, H.to_oracle_fp(capi.FilterParams.default()), host_ctg, *host_cr)
        del host_rr, host_cr
        ocsr = og.csr()
        H.assert_csr_equal(csr, ocsr)
        del ocsr
        t_graph = time.time() - t_graph
        owp = po.WalkParams(wp.rebuild_attempts, wp.backtrack_attempts, wp.len_threshold)
        osets = []
        evals = units = 0
        t_oracle = time.time()
        for k, (kind, metric) in enumerate((("mc", po.METRIC_ES), ("mc", po.METRIC_OS), ("det", po.METRIC_ES), ("det", po.METRIC_OS))):
            ops, octr, ostats = (og.walk_mc(metric, owp, po.SEED_PER_WALK, threads=threads) if kind == "mc"
                                 else og.walk_det(metric, threads=threads))
            H.assert_stats_equal(sets[k].unit_stats(), ostats)
            assert sets[k].counters.evals == octr.evals
            h = sets[k].to_host()
            H.assert_paths_equal(h, ops)
            evals += int(octr.evals)
            units += int(ostats.shape[0])
            osets.append(ops)
            del h
        t_oracle = time.time() - t_oracle
        del sets
        # ---- expected scaffold: the host stages over the oracle's paths_
        allp = po.concat_paths(osets)
        del osets
        uniq, dropped = H.drop_adjacent_duplicates(allp)
        names = [names_of(int(i)) for i in csr.node_name_id]
        H.write_binary_dump(os.path.join(tmp, "bin"), names, csr, uniq)
        n_paths = len(uniq.length)
        del uniq, allp, csr, og
        del g
        ctx.trim()
        want_fa, got_fa = os.path.join(tmp, "want.fasta"), os.path.join(tmp, "got.fasta")
        t0 = time.time()
        exp = subprocess.run([post, "--binary", os.path.join(tmp, "bin"), reads, contigs, want_fa], capture_output=True, text=True, timeout=900)
        t_post = time.time() - t0
        assert exp.returncode == 0, exp.stderr[-2000:]
        shutil.rmtree(os.path.join(tmp, "bin"))
        t0 = time.time()
        res = subprocess.run([binary, rr, cr, reads, contigs, got_fa], capture_output=True, text=True, timeout=600)
        t_hip = time.time() - t0
        assert res.returncode == 0, res.stderr[-2000:]
        pick = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(("Consensus length", "Final consensus"))]
        assert pick(res.stdout) == pick(exp.stdout)
        assert "- total_num: %d" % n_paths in res.stdout
        a, b = open(got_fa, "rb").read(), open(want_fa, "rb").read()
        assert len(a) == len(b) and a == b
        md5 = hashlib.md5(a).hexdigest()
        print("\nfull-config scaffold parity: the oracle's own graph from the records in %.0f s; %d units' worth of %.3g evals by the oracle on %d threads in %.0f s; %d paths after filterUnique "
              "(%d adjacent duplicates); scaffold %d bytes md5 %s; telomeri_post %.0f s, telomeri_hip %.1f s; whole test %.0f s" % (
                  t_graph, units, evals, threads, t_oracle, n_paths, dropped, len(a), md5, t_post, t_hip, time.time() - t_start))
        rec = os.path.join(root, "gpurun_out", "full_config_scaffold_parity.json")
        try:
            os.makedirs(os.path.dirname(rec), exist_ok=True)
            import json
            json.dump({"config": "athaliana", "oracle_own_graph_from_records_s": t_graph, "units": units, "evals": evals, "oracle_threads": threads, "oracle_s": t_oracle, "paths": n_paths,
                       "adjacent_duplicates": dropped, "scaffold_bytes": len(a), "scaffold_md5": md5, "telomeri_post_s": t_post,
                       "telomeri_hip_s": t_hip, "test_s": time.time() - t_start}, open(rec, "w"))
        except OSError:
            pass
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
