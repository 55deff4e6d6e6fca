"""Shared test plumbing: one dataset -> the oracle's graph (CPU) and the device graph (HIP)."""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from oracle import pyoracle as po
from telomeri_amd import synth

REC_COLS = ("q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches", "strand")


def oracle_graph(ds: synth.Dataset, fp: po.FilterParams, files=None) -> po.Graph:
    """files: list of (Records, anchors) in load order; default rr then cr (reference Main.cpp:223)."""
    g = po.Graph()
    ctg = ds.name_is_ctg().numpy().astype(np.uint8)
    for rec, anchors in files or [(ds.rr, False), (ds.cr, True)]:
        cols = [getattr(rec, c).cpu().numpy() for c in REC_COLS]
        g.add_records(anchors, fp, ctg, rec.q_id.cpu().numpy(), rec.t_id.cpu().numpy(), *cols)
    return g


def to_oracle_fp(fp) -> po.FilterParams:
    return po.FilterParams(fp.mode, fp.min_overlap_length, fp.min_overlap_percentage, fp.max_overhang_length,
                           fp.max_overhang_percentage)


def to_oracle_wp(wp) -> po.WalkParams:
    return po.WalkParams(wp.rebuild_attempts, wp.backtrack_attempts, wp.len_threshold)


def device_graph(ctx, ds: synth.Dataset, fp, files=None):
    files = files or [(ds.rr, False), (ds.cr, True)]
    recs = [r.to("cuda") for r, _ in files]
    return ctx.build_graph(fp, recs, [a for _, a in files], ds.n_reads + ds.n_contigs, ds.name_is_ctg())


def assert_csr_equal(dev_csr, ora_csr) -> None:
    assert dev_csr.n_nodes == ora_csr.n_nodes
    assert dev_csr.n_adj == ora_csr.n_adj
    np.testing.assert_array_equal(dev_csr.row_ptr, ora_csr.row_ptr)
    np.testing.assert_array_equal(dev_csr.node_len, ora_csr.node_len)
    np.testing.assert_array_equal(dev_csr.node_anchor, ora_csr.node_anchor)
    for name in ("nbr", "q_start", "q_end", "t_start", "t_end", "strand"):
        np.testing.assert_array_equal(getattr(dev_csr, name), getattr(ora_csr, name), err_msg=name)
    for name in ("os", "si", "es"):   # bit patterns, not values
        np.testing.assert_array_equal(getattr(dev_csr, name).view(np.uint32), getattr(ora_csr, name).view(np.uint32),
                                      err_msg=name)


def assert_paths_equal(dev_paths, ora_paths) -> None:
    """Both are packed Path sets (unit, length, off, nodes, slots)."""
    np.testing.assert_array_equal(dev_paths.unit, ora_paths.unit)
    np.testing.assert_array_equal(dev_paths.length, ora_paths.length)
    np.testing.assert_array_equal(dev_paths.off, ora_paths.off)
    np.testing.assert_array_equal(dev_paths.nodes, ora_paths.nodes)
    np.testing.assert_array_equal(dev_paths.slots, ora_paths.slots)


def assert_stats_equal(dev_stats: np.ndarray, ora_stats: np.ndarray) -> None:
    for f in ("status", "n_nodes", "steps", "evals", "draws", "length"):
        bad = np.nonzero(dev_stats[f] != ora_stats[f])[0]
        assert bad.size == 0, "per-unit %s differs at units %s: device %s oracle %s" % (
            f, bad[:8], dev_stats[f][bad[:8]], ora_stats[f][bad[:8]])


def quirks_dataset(seed: int = 5) -> synth.Dataset:
    """A small dataset plus the record shapes the reference treats specially:
    multi-edges, self overlaps, a contig<->contig record, a `ctg` name as QUERY of a cr record,
    a read first seen in the cr file, records that fail the filter, shuffled record order."""
    ds = synth.make_dataset("quirks", seed=seed, genome_len=150_000, n_contigs=3, contig_len=25_000, n_reads=160,
                            read_len_lo=4_000, read_len_hi=11_000)
    g = torch.Generator().manual_seed(seed)
    rr, cr = ds.rr, ds.cr
    # 15 % multi-edges: repeat some rr records with shifted coordinates
    n = len(rr)
    dup = torch.nonzero(torch.rand(n, generator=g) < 0.15).flatten()
    extra = rr.take(dup)
    extra.q_start = extra.q_start + 7
    extra.q_end = torch.maximum(extra.q_end - 5, extra.q_start + 1)
    extra.matches = torch.clamp(extra.matches - 11, min=0)
    # self overlaps (same name): must be rejected by OverlapGraph.cpp:157
    selfo = rr.take(torch.arange(0, min(n, 40), 4))
    selfo.t_id = selfo.q_id.clone()
    rr = rr.cat(extra).cat(selfo)
    perm = torch.randperm(len(rr), generator=g)
    rr = rr.take(perm)
    # cr: a contig<->contig record (forced-anchor rule, DET break), a ctg-as-query record, shuffled
    nreads = ds.n_reads
    def rec(q, t, ql, qs, qe, tl, ts, te, m, s):
        i32 = lambda v: torch.tensor([v], dtype=torch.int32)
        return synth.Records(i32(q), i32(t), i32(ql), i32(qs), i32(qe), i32(tl), i32(ts), i32(te), i32(m), i32(s))
    c2c = rec(nreads + 1, nreads + 0, 25_000, 20_000, 24_990, 25_000, 10, 5_000, 4_500, 0)     # ctg2 (query) -> ctg1
    r = int(cr.q_id[0])
    cq = rec(nreads + 2, r, 25_000, 100, 6_000, int(ds.read_len[r]), 50, 5_950, 5_000, 1)     # ctg3 as QUERY vs a read
    cr = cr.cat(c2c).cat(cq)
    perm = torch.randperm(len(cr), generator=g)
    cr = cr.take(perm)
    return dataclasses.replace(ds, rr=rr, cr=cr)


def overflow_dataset(seed: int = 51) -> synth.Dataset:
    """Contigs that CLAIM 40 Mb (the cr file's target-length column; the geometry stays small): every path is ~40 Mb long, and
    the `int` accumulator of PathGroup::calculateConsensusPath (reference PathGroup.cpp:66-68, std::accumulate(..., 0, ...))
    overflows after 54 paths.  Run with --len-thr 100000000."""
    ds = synth.make_dataset("overflow", seed=seed, genome_len=220_000, n_contigs=3, contig_len=32_000, n_reads=300,
                            read_len_lo=4_000, read_len_hi=12_000)
    return dataclasses.replace(ds, cr=dataclasses.replace(ds.cr, t_len=ds.cr.t_len + 40_000_000))


def dataset_by_name(name: str) -> synth.Dataset:
    """The dataset behind a committed golden set (tests/golden/<name>.json) or a named config."""
    if name == "quirks":
        return quirks_dataset()
    if name == "overflow":
        return overflow_dataset()
    return synth.make_config(name)


def write_binary_dump(dirname: str, names, csr, paths) -> None:
    """The input of `telomeri_post --binary <dir>` (host/post_tool.cpp): the graph as raw CSR arrays and a packed Path set
    (paths_ AFTER filterUnique), for comparisons at sizes where the text dump would be tens of GB."""
    import os
    os.makedirs(dirname, exist_ok=True)
    p = lambda n: os.path.join(dirname, n)
    with open(p("names.txt"), "w") as fh:
        fh.write("\n".join(names) + ("\n" if len(names) else ""))
    np.ascontiguousarray(csr.node_len, np.uint32).tofile(p("node_len.u32"))
    np.ascontiguousarray(csr.node_anchor, np.uint8).tofile(p("node_anchor.u8"))
    np.ascontiguousarray(csr.row_ptr, np.uint64).tofile(p("row_ptr.u64"))
    for name, dt, ext in (("nbr", np.uint32, "u32"), ("q_start", np.uint32, "u32"), ("q_end", np.uint32, "u32"),
                          ("t_start", np.uint32, "u32"), ("t_end", np.uint32, "u32"), ("os", np.float32, "f32"),
                          ("si", np.float32, "f32"), ("es", np.float32, "f32"), ("strand", np.uint8, "u8")):
        np.ascontiguousarray(getattr(csr, name), dt).tofile(p("%s.%s" % (name, ext)))
    np.ascontiguousarray(paths.off, np.uint64).tofile(p("path_off.u64"))
    np.ascontiguousarray(paths.length, np.int64).tofile(p("path_length.i64"))
    np.ascontiguousarray(paths.nodes, np.uint32).tofile(p("path_nodes.u32"))
    np.ascontiguousarray(paths.slots, np.uint32).tofile(p("path_slots.u32"))
    with open(p("meta.txt"), "w") as fh:
        fh.write("%d %d %d %d\n" % (csr.n_nodes, csr.n_adj, len(paths.length), paths.nodes.shape[0]))


def drop_adjacent_duplicates(ps):
    """PathManager::filterUnique (reference PathManager.cpp:432-438) on a packed set WITHOUT the per-path Python loop of
    oracle.pyoracle.filter_unique: only neighbours with equal node counts are compared.  -> (set, number dropped)"""
    from oracle import pyoracle as po
    off = ps.off.astype(np.int64)
    cnt = np.diff(off)
    n = cnt.shape[0]
    keep = np.ones(n, bool)
    same_cnt = np.nonzero(cnt[1:] == cnt[:-1])[0] + 1
    # std::unique compares each element with the last one KEPT: walk the candidates in order
    prev = np.arange(n) - 1
    for i in same_cnt:
        j = int(prev[i])
        while j >= 0 and not keep[j]:
            j -= 1
        if j >= 0 and cnt[j] == cnt[i] and np.array_equal(ps.nodes[off[i]:off[i + 1]], ps.nodes[off[j]:off[j + 1]]):
            keep[i] = False
    dropped = int(n - keep.sum())
    if dropped == 0:
        return ps, 0
    idx = np.nonzero(keep)[0]
    sel = np.repeat(keep, cnt)
    new_off = np.zeros(idx.shape[0] + 1, np.uint64)
    new_off[1:] = np.cumsum(cnt[idx])
    return po.PathSet(ps.unit[idx], ps.length[idx], new_off, ps.nodes[sel], ps.slots[sel]), dropped
