"""The C++ host (telomeri_amd/host: the reference's class surface over the C ABI) end to end on the GPU:
PAF text in -> graph + paths out, compared with the UNMODIFIED reference's committed output (tests/golden) for
everything that does not involve the random stream (graph bits, node names and order, both deterministic
builders), and with the oracle under per-walk seeding for the Monte-Carlo builders."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as G
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_hip")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binary():
    if not os.path.exists(BIN):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "telomeri_amd", "host")], check=True)
    return BIN


def _flags(args):
    return [a for a in args if a != "--sqrt"]


@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks"])
def test_cli_against_reference_dump(binary, name):
    meta = G.load_json(name + ".json")
    fp, wp = G.args_to_params(meta["args"])
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")
        ref = po.parse_ref_dump(G.gunzip_to(tmp, name + ".dump.gz"))
        out = os.path.join(tmp, "host.dump")
        res = subprocess.run([binary] + _flags(meta["args"]) + ["--dump", out, rr, cr, "reads.fa", "contigs.fa", "out.fa"],
                             capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        got = po.parse_ref_dump(out)
        g = po.Graph()
        assert g.load_paf(rr, False, fp) and g.load_paf(cr, True, fp)
    # the graph, against the reference itself
    assert got.node_name == ref.node_name
    np.testing.assert_array_equal(got.node_anchor, ref.node_anchor)
    np.testing.assert_array_equal(got.node_len, ref.node_len)
    np.testing.assert_array_equal(got.adj, ref.adj)
    # log lines the reference prints
    assert "Loaded %d/" % (ref.adj.shape[0] // 2) in res.stdout
    assert "> Monte Carlo heuristic: EXTENSION_SCORE" in res.stdout
    assert "> Deterministic heuristic: OVERLAP_SCORE" in res.stdout
    # deterministic builders, against the reference itself
    for b in (2, 3):
        assert len(got.builders[b]) == len(ref.builders[b])
        for (l1, n1, s1), (l2, n2, s2) in zip(got.builders[b], ref.builders[b]):
            assert l1 == l2
            np.testing.assert_array_equal(n1, n2)
            np.testing.assert_array_equal(s1, s2)
    # filterUnique + getPathsBetweenAnchors (device post-passes behind the host's members), against the oracle's
    # restatement over the same per-walk-seeded paths
    osets = [g.walk_mc(m, wp, po.SEED_PER_WALK)[0] for m in (po.METRIC_ES, po.METRIC_OS)] + [g.walk_det(m)[0] for m in (0, 1)]
    ouniq = po.filter_unique(po.concat_paths(osets))
    assert len(got.unique) == len(ouniq)
    for i, (l1, n1, s1) in enumerate(got.unique):
        l2, n2, s2 = ouniq.path(i)
        assert l1 == l2
        np.testing.assert_array_equal(n1, n2)
        np.testing.assert_array_equal(s1, s2)
    assert got.pairs == po.paths_between_anchors(ouniq)
    assert "- total_num: %d" % len(ouniq) in res.stdout
    # Monte-Carlo builders, against the oracle under per-walk seeding
    for b, metric in ((0, po.METRIC_ES), (1, po.METRIC_OS)):
        ps = g.walk_mc(metric, wp, po.SEED_PER_WALK)[0]
        assert len(got.builders[b]) == len(ps)
        for i, (l1, n1, s1) in enumerate(got.builders[b]):
            l2, n2, s2 = ps.path(i)
            assert l1 == l2
            np.testing.assert_array_equal(n1, n2)
            np.testing.assert_array_equal(s1, s2)


def test_cli_error_conventions(binary):
    res = subprocess.run([binary, "/nonexistent/rr.paf", "/nonexistent/cr.paf", "a", "b", "c"], capture_output=True, text=True)
    assert res.returncode == 1 and "Cannot open file: /nonexistent/rr.paf" in res.stderr
    res = subprocess.run([binary, "only", "three", "args"], capture_output=True, text=True)
    assert res.returncode == 1 and "Please provide a path" in res.stderr


def test_bundled_fixture_through_cli(binary):
    """The reference's own data file: 398 of 24127 records kept under the default filter."""
    known = G.load_json("ecoli_cr_known.json")
    with tempfile.TemporaryDirectory() as tmp:
        paf = G.gunzip_to(tmp, "ecoli_cr.paf.gz")
        empty = os.path.join(tmp, "empty.paf")
        open(empty, "w").close()
        for key, flags in (("avg", []), ("min", ["--filter-min"]), ("max", ["--filter-max"])):
            res = subprocess.run([binary] + flags + ["--rb-att", "20", empty, paf, "r", "c", "o"], capture_output=True,
                                 text=True, timeout=300)
            assert res.returncode == 0, res.stderr
            assert "Loaded %d/24127 edges" % known[key]["kept"] in res.stdout
            assert "-  anchor: 3" in res.stdout
            assert "- max_con: %d" % known[key]["max_con"] in res.stdout
