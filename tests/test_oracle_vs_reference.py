"""Live check of the oracle against the UNMODIFIED reference (oracle/_ref/ref_dump, built from the
sources under /root/reference by `make -C oracle ref`).  Runs only where that binary exists (the build
container); on the GPU box the committed fixtures of tests/golden/ stand in (test_oracle_golden.py)."""
import os
import tempfile

import numpy as np
import pytest

import golden_util as G
import helpers as H
from oracle import pyoracle as po
from telomeri_amd import synth

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref/ref_dump not built (no /root/reference here)")


def _compare(ds, args):
    fp, wp = G.args_to_params(args)
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), os.path.join(tmp, "cr.paf")
        synth.write_paf(ds, ds.rr, rr)
        synth.write_paf(ds, ds.cr, cr)
        ref = po.run_ref_dump(rr, cr, os.path.join(tmp, "out.dump"), args)
        g = po.Graph()
        assert g.load_paf(rr, False, fp) and g.load_paf(cr, True, fp)
    c = g.csr()
    np.testing.assert_array_equal(G.csr_as_dump_rows(c), ref.adj)
    np.testing.assert_array_equal(c.node_anchor, ref.node_anchor)
    # the integer-column ingestion builds the same graph as the text loader
    g2 = H.oracle_graph(ds, fp)
    c2 = g2.csr()
    np.testing.assert_array_equal(G.csr_as_dump_rows(c2), ref.adj)
    for b, (mc, metric) in {0: (True, 0), 1: (True, 1), 2: (False, 0), 3: (False, 1)}.items():
        ps = g.walk_mc(metric, wp, po.SEED_SHARED)[0] if mc else g.walk_det(metric)[0]
        paths = ref.builders[b]
        assert len(ps) == len(paths)
        for i, (length, nodes, slots) in enumerate(paths):
            l2, n2, s2 = ps.path(i)
            assert l2 == length
            np.testing.assert_array_equal(n2, nodes)
            np.testing.assert_array_equal(s2, slots)


@pytest.mark.parametrize("seed,args", [(21, ["--rb-att", "40"]), (22, ["--filter-min", "--rb-att", "30", "--bt-att", "2"]),
                                       (23, ["--filter-sum", "--rb-att", "25", "--len-thr", "120000"]),
                                       (24, ["--filter-max", "--rb-att", "25", "--min-oll", "2000", "--max-ohp", "0.8"])])
def test_fresh_datasets(seed, args):
    ds = synth.make_dataset("fresh", seed=seed, genome_len=180_000, n_contigs=3, contig_len=28_000, n_reads=230,
                            read_len_lo=4_000, read_len_hi=12_000)
    _compare(ds, args)


def test_mid_size_dataset():
    """The oracle against the unmodified reference where the reference's O(N) nodeIndex scan still allows it: 16 000 reads,
    30 contigs, 456 000 records -> ~0.9 M adjacency entries, and every path of all four builders (shared-stream Monte-Carlo,
    200 trials per anchor) -- sixty times the size of the other live sets (the reference needs ~25 s for it)."""
    ds = synth.make_dataset("mid", seed=61, genome_len=9_000_000, n_contigs=30, contig_len=60_000, n_reads=16_000,
                            read_len_lo=4_000, read_len_hi=14_000)
    _compare(ds, ["--rb-att", "200"])


def test_quirks_dataset_other_seed():
    _compare(H.quirks_dataset(seed=9), ["--rb-att", "30", "--bt-att", "4"])


def test_low_coverage_backtracking():
    ds = synth.make_dataset("lowcov", seed=31, genome_len=260_000, n_contigs=3, contig_len=35_000, n_reads=150,
                            read_len_lo=4_000, read_len_hi=12_000)
    _compare(ds, ["--rb-att", "60", "--bt-att", "3"])


@pytest.mark.parametrize("seed,args", [(21, ["--rb-att", "40"]), (33, ["--rb-att", "30", "--bt-att", "4"]),
                                       (35, ["--filter-min", "--rb-att", "25", "--len-thr", "150000"])])
def test_hostile_datasets(seed, args):
    """The hostile generator (telomeri_amd.synth.make_hard_dataset: repeat families whose copies make hub lists and
    far-apart neighbours, chimeric reads, log-normal reads, skewed contigs, contig <-> contig records, multi-edges):
    graph and all four builders of the oracle against the unmodified reference."""
    _compare(synth.make_config("hard_tiny", seed=seed), args)


def test_token_stream_quirks_against_reference():
    """Malformed PAF text: where operator>> stops (OverlapGraph.cpp:109-127) -- record counts and graphs of the oracle's
    loader against the reference binary."""
    import re
    import subprocess
    from test_host_ingest import LINE
    texts = {
        "ragged_17": LINE + LINE.replace("dv:f:0.01", "dv:f:0.01\textra") + LINE * 3,
        "ragged_15": LINE * 2 + LINE.replace("\tdv:f:0.01", "") + LINE * 2,
        "suffix": LINE + LINE.replace("\t6500\t", "\t65x0\t") + LINE,
        "overflow": LINE + LINE.replace("\t9000\t", "\t99999999999\t") + LINE,
        "strand2": LINE + LINE.replace("\t+\t", "\t+-\t") + LINE,
        "truncated": LINE * 3 + "readC\t100\t1",
        "blank_crlf": LINE + "\n\n" + LINE.replace("\n", "\r\n") + "   \n" + LINE,
        "split": LINE.replace("\t+\t", "\n+\n") * 2,
    }
    fp = po.FilterParams.default("min")
    with tempfile.TemporaryDirectory() as tmp:
        empty = os.path.join(tmp, "empty.paf")
        open(empty, "w").close()
        for name, text in texts.items():
            p = os.path.join(tmp, name + ".paf")
            open(p, "w", newline="").write(text)
            out = os.path.join(tmp, name + ".dump")
            res = subprocess.run([po.REF_DUMP, "--filter-min", "--no-paths", p, empty, out], capture_output=True, text=True)
            assert res.returncode == 0
            m = re.search(r"Loaded (\d+)/(\d+) edges from " + re.escape(p), res.stdout)
            ref = po.parse_ref_dump(out)
            g = po.Graph()
            assert g.load_paf(p, False, fp)
            assert g.records == [int(m.group(2))], name
            c = g.csr()
            assert c.n_edges == int(m.group(1)), name
            if ref.adj is not None:
                np.testing.assert_array_equal(G.csr_as_dump_rows(c), ref.adj)
