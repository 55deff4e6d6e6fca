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
        res = subprocess.run([binary] + _flags(meta["args"]) + ["--no-scaffold", "--dump", out, rr, cr, "reads.fa", "contigs.fa",
                                                                "out.fa"], capture_output=True, text=True, timeout=300)
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


def _write_dump(path, g, csr, paths):
    """A ref_dump-format file (graph + `U` paths) from the oracle's objects: the input of telomeri_post."""
    with open(path, "w") as fh:
        fh.write("TLMDUMP 1\nN %d\n" % csr.n_nodes)
        deg = np.diff(csr.row_ptr.astype(np.int64))
        for i in range(csr.n_nodes):
            fh.write("n %d %d %d %d %s\n" % (i, csr.node_anchor[i], csr.node_len[i], deg[i], g.node_name(i)))
        fh.write("A %d\n" % csr.n_adj)
        for row in G.csr_as_dump_rows(csr):
            fh.write("a " + " ".join(str(int(x)) for x in row) + "\n")
        fh.write("U %d\n" % len(paths))
        for i in range(len(paths)):
            length, nodes, slots = paths.path(i)
            fh.write("p %d %d %s %s\n" % (length, len(nodes), " ".join(map(str, nodes)), " ".join(map(str, slots))))


@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks"])
def test_cli_scaffold_end_to_end(binary, name):
    """PAF in -> scaffold FASTA out on the GPU host program; expected = the (reference-validated, see
    test_host_post_stages.py) host stages applied to the oracle's per-walk-seeded paths."""
    from telomeri_amd import synth
    import helpers as H
    meta = G.load_json(name + ".json")
    fp, wp = G.args_to_params(meta["args"])
    ds = H.quirks_dataset() if name == "quirks" else synth.make_config(name)
    post = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_post")
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")
        reads, contigs = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa")
        synth.write_fasta(ds, reads, contigs)
        out = os.path.join(tmp, "gpu.fa")
        res = subprocess.run([binary] + _flags(meta["args"]) + [rr, cr, reads, contigs, out], capture_output=True, text=True,
                             timeout=300)
        g = po.Graph()
        assert g.load_paf(rr, False, fp) and g.load_paf(cr, True, fp)
        osets = [g.walk_mc(m, wp, po.SEED_PER_WALK)[0] for m in (po.METRIC_ES, po.METRIC_OS)] + [g.walk_det(m)[0] for m in (0, 1)]
        ouniq = po.filter_unique(po.concat_paths(osets))
        dump = os.path.join(tmp, "oracle.dump")
        _write_dump(dump, g, g.csr(), ouniq)
        want = os.path.join(tmp, "want.fa")
        exp = subprocess.run([post, dump, reads, contigs, want], capture_output=True, text=True, timeout=120)
        assert res.returncode == exp.returncode, res.stderr + exp.stderr
        if exp.returncode == 0:
            assert open(out, "rb").read() == open(want, "rb").read()
            pick = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(("Consensus length", "Final consensus", "ctg", "read"))]
            assert pick(res.stdout) == pick(exp.stdout)


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
            res = subprocess.run([binary] + flags + ["--no-scaffold", "--rb-att", "20", empty, paf, "r", "c", "o"],
                                 capture_output=True, text=True, timeout=300)
            assert res.returncode == 0, res.stderr
            assert "Loaded %d/24127 edges" % known[key]["kept"] in res.stdout
            assert "-  anchor: 3" in res.stdout
            assert "- max_con: %d" % known[key]["max_con"] in res.stdout


@pytest.mark.parametrize("name", ["tiny", "quirks"])
def test_lean_host_equals_materialised_host(binary, name):
    """Default mode (adjacency and path node lists stay on the device: stats() reduced there, Edge records fetched for the
    scaffold's paths only) writes the same report and the same scaffold as --materialise (every nodes_[..].edges, edges_
    and Path list filled as the reference does)."""
    from telomeri_amd import synth
    import helpers as H
    meta = G.load_json(name + ".json")
    ds = H.quirks_dataset() if name == "quirks" else synth.make_config(name)
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")
        reads, contigs = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa")
        synth.write_fasta(ds, reads, contigs)
        outs = []
        for extra in ([], ["--materialise"]):
            out = os.path.join(tmp, "o%d.fa" % len(outs))
            res = subprocess.run([binary] + _flags(meta["args"]) + extra + [rr, cr, reads, contigs, out], capture_output=True,
                                 text=True, timeout=300)
            body = open(out, "rb").read() if os.path.exists(out) else b""
            keep = [ln for ln in res.stdout.splitlines() if ln.startswith(("-", "Nodes", "Edges", "Paths", "Found", "Consensus",
                                                                           "Final", "ctg", "read", "Anchor pairs"))]
            outs.append((res.returncode, body, keep))
        assert outs[0] == outs[1]
        assert any(ln.startswith("-  min_OS") for ln in outs[0][2])
