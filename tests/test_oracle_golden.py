"""Pins the oracle (oracle/telomeri_oracle.c) to the committed outputs of the UNMODIFIED reference
(tests/golden/, produced by tests/golden/make_golden.py through oracle/_ref/ref_dump)."""
import hashlib
import tempfile

import numpy as np
import pytest

import golden_util as G
from oracle import pyoracle as po


def test_mt19937_known_answers():
    v = G.load_json("mt19937.json")
    raw = po.mt19937_raw(5489, 10000)
    assert raw[:16].tolist() == v["raw"]
    assert int(raw[9999]) == v["raw_10000th"] == 4123659995      # [rand.predef] requirement on mt19937
    can = po.mt19937_canonical(5489, 700)
    assert can.view(np.uint64).tolist() == v["canonical_bits"]
    can2 = po.mt19937_canonical(v["seed"], 8)
    assert can2.view(np.uint64).tolist() == v["seed_canonical_bits"]
    assert 0.0 <= can.min() and can.max() < 1.0


def _digest(g: po.Graph) -> str:
    c = g.csr()
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(c.node_anchor).tobytes())
    h.update(np.ascontiguousarray(c.node_len).tobytes())
    h.update("\n".join(g.node_name(i) for i in range(c.n_nodes)).encode())
    h.update(np.ascontiguousarray(G.csr_as_dump_rows(c)).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("key", ["avg", "sum", "min", "max", "strict"])
def test_bundled_fixture_known_answers(key):
    """The reference's own data file (test/res/EColi_CR_overlaps.paf), SURVEY.md §4 table."""
    known = G.load_json("ecoli_cr_known.json")[key]
    expected_kept = {"avg": 398, "sum": 398, "min": 5507, "max": 248, "strict": 32}[key]
    assert known["kept"] == expected_kept
    with tempfile.TemporaryDirectory() as tmp:
        paf = G.gunzip_to(tmp, "ecoli_cr.paf.gz")
        empty = tmp + "/empty.paf"
        open(empty, "w").close()
        fp, wp = G.args_to_params(known["args"] + ["--rb-att", "20"])
        g = po.Graph()
        assert g.load_paf(empty, False, fp) and g.load_paf(paf, True, fp)
        assert g.records == [0, 24127]
        c = g.csr()
        assert c.n_edges == known["kept"] and c.n_nodes == known["nodes"]
        assert int(c.node_anchor.sum()) == known["anchors"] == 3
        assert int(np.diff(c.row_ptr).max()) == known["max_con"]
        assert float(c.os.min()) == known["min_os"] and float(c.os.max()) == known["max_os"]
        assert float(c.es.min()) == known["min_es"] and float(c.es.max()) == known["max_es"]
        assert _digest(g) == known["graph_sha256"]
        for b, (mc, metric) in {0: (True, 0), 1: (True, 1), 2: (False, 0), 3: (False, 1)}.items():
            ps = g.walk_mc(metric, wp, po.SEED_SHARED)[0] if mc else g.walk_det(metric)[0]
            assert len(ps) == known["paths"][str(b)]


def test_load_missing_file_fails():
    g = po.Graph()
    assert not g.load_paf("/nonexistent/x.paf", False, po.FilterParams.default())


@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks", "hard_tiny", "overflow"])
def test_synthetic_sets_graph_and_paths(name):
    meta = G.load_json(name + ".json")
    fp, wp = G.args_to_params(meta["args"])
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")
        ref = po.parse_ref_dump(G.gunzip_to(tmp, name + ".dump.gz"))
        g = po.Graph()
        assert g.load_paf(rr, False, fp) and g.load_paf(cr, True, fp)
    c = g.csr()
    assert [g.node_name(i) for i in range(c.n_nodes)] == ref.node_name
    np.testing.assert_array_equal(c.node_anchor, ref.node_anchor)
    np.testing.assert_array_equal(c.node_len, ref.node_len)
    np.testing.assert_array_equal(G.csr_as_dump_rows(c), ref.adj)
    builders = {0: (True, po.METRIC_ES), 1: (True, po.METRIC_OS), 2: (False, po.METRIC_ES), 3: (False, po.METRIC_OS),
                4: (True, po.METRIC_ES_SQRT), 5: (True, po.METRIC_OS_SQRT), 6: (False, po.METRIC_ES_SQRT),
                7: (False, po.METRIC_OS_SQRT)}
    assert set(ref.builders) <= set(builders)
    n_checked = 0
    for b, paths in ref.builders.items():
        mc, metric = builders[b]
        # SEED_SHARED = the reference's one default-seeded mt19937 per builder call (PathManager.cpp:13)
        ps = g.walk_mc(metric, wp, po.SEED_SHARED)[0] if mc else g.walk_det(metric)[0]
        assert len(ps) == len(paths) == meta["paths"][str(b)], b
        for i, (length, nodes, slots) in enumerate(paths):
            l2, n2, s2 = ps.path(i)
            assert l2 == length, (b, i)
            np.testing.assert_array_equal(n2, nodes)
            np.testing.assert_array_equal(s2, slots)
            n_checked += 1
    assert n_checked > 50


@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks"])
def test_filter_unique_and_anchor_pairs(name):
    """paths_ of one PathManager after the four builders, filterUnique, getPathsBetweenAnchors."""
    meta = G.load_json(name + ".json")
    fp, wp = G.args_to_params(meta["args"])
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")
        ref = po.parse_ref_dump(G.gunzip_to(tmp, name + ".dump.gz"))
        g = po.Graph()
        assert g.load_paf(rr, False, fp) and g.load_paf(cr, True, fp)
    sets = [g.walk_mc(po.METRIC_ES, wp, po.SEED_SHARED)[0], g.walk_mc(po.METRIC_OS, wp, po.SEED_SHARED)[0],
            g.walk_det(po.METRIC_ES)[0], g.walk_det(po.METRIC_OS)[0]]
    allp = po.concat_paths(sets)
    uniq = po.filter_unique(allp)
    assert len(uniq) == len(ref.unique)
    assert len(uniq) <= len(allp)
    for i, (length, nodes, slots) in enumerate(ref.unique):
        l2, n2, s2 = uniq.path(i)
        assert l2 == length
        np.testing.assert_array_equal(n2, nodes)
        np.testing.assert_array_equal(s2, slots)
    pairs = po.paths_between_anchors(uniq)
    assert [(a, b, idx) for a, b, idx in pairs] == [(a, b, idx) for a, b, idx in ref.pairs]


def test_per_walk_seed_first_unit_equals_shared_stream_when_seed_matches():
    """Per-walk seeding only moves the seeding point: a walk whose per-walk seed is S equals the FIRST walk of a
    shared stream seeded S.  Checked through the oracle's own RNG helper against libstdc++ vectors above; here:
    threads and unit ranges do not change per-walk results."""
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, "tiny_rr.paf.gz"), G.gunzip_to(tmp, "tiny_cr.paf.gz")
        g = po.Graph()
        fp = po.FilterParams.default()
        assert g.load_paf(rr, False, fp) and g.load_paf(cr, True, fp)
    wp = po.WalkParams.default(30)
    whole, cw, sw = g.walk_mc(po.METRIC_ES, wp, po.SEED_PER_WALK, threads=1)
    par, cp, sp = g.walk_mc(po.METRIC_ES, wp, po.SEED_PER_WALK, threads=4)
    np.testing.assert_array_equal(whole.nodes, par.nodes)
    np.testing.assert_array_equal(sw, sp)
    total = g.mc_units(wp)
    a = g.walk_mc(po.METRIC_ES, wp, po.SEED_PER_WALK, 0, total // 2)[0]
    b = g.walk_mc(po.METRIC_ES, wp, po.SEED_PER_WALK, total // 2, total)[0]
    np.testing.assert_array_equal(np.concatenate([a.nodes, b.nodes]), whole.nodes)
    # shared-stream ranges: walking [k, n) after skipping [0, k) equals the tail of the whole run
    sh = g.walk_mc(po.METRIC_ES, wp, po.SEED_SHARED)[0]
    tail = g.walk_mc(po.METRIC_ES, wp, po.SEED_SHARED, total // 2, total)[0]
    keep = sh.unit >= total // 2
    np.testing.assert_array_equal(tail.unit, sh.unit[keep])
    np.testing.assert_array_equal(tail.length, sh.length[keep])
