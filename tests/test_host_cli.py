"""The C++ host (telomeri_amd/host: the reference's class surface over the C ABI) end to end on the GPU:
PAF text in -> graph + paths out, compared with the UNMODIFIED reference's committed output (tests/golden) for
everything that does not involve the random stream (graph bits, node names and order, both deterministic
builders), and with the oracle under per-walk seeding for the Monte-Carlo builders."""
import os
import re
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


@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks", "hard_tiny", "overflow"])
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


@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks", "hard_tiny", "overflow"])
def test_cli_scaffold_end_to_end(binary, name):
    """PAF in -> scaffold FASTA out on the GPU host program; expected = the (reference-validated, see
    test_host_post_stages.py) host stages applied to the oracle's per-walk-seeded paths."""
    from telomeri_amd import synth
    import helpers as H
    meta = G.load_json(name + ".json")
    fp, wp = G.args_to_params(meta["args"])
    ds = H.dataset_by_name(name)
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
            # the same program with its shortcuts off: no helper process (overlap text uploaded and sequence files scanned
            # by this process, the latter where the reference loads them), destructors instead of _Exit -- same bytes,
            # same report
            out2 = os.path.join(tmp, "gpu2.fa")
            res2 = subprocess.run([binary] + _flags(meta["args"]) + [rr, cr, reads, contigs, out2], capture_output=True, text=True,
                                  timeout=300, env=dict(os.environ, TLM_NO_FASTA_AHEAD="1", TLM_CLEAN_EXIT="1", TLM_NO_UPLOAD_HELPER="1"))
            assert res2.returncode == 0, res2.stderr
            assert open(out2, "rb").read() == open(want, "rb").read()
            assert pick(res2.stdout) == pick(exp.stdout)


def test_cli_error_conventions(binary):
    res = subprocess.run([binary, "/nonexistent/rr.paf", "/nonexistent/cr.paf", "a", "b", "c"], capture_output=True, text=True)
    assert res.returncode == 1 and "Cannot open file: /nonexistent/rr.paf" in res.stderr
    res = subprocess.run([binary, "only", "three", "args"], capture_output=True, text=True)
    assert res.returncode == 1 and "Please provide a path" in res.stderr
    # a sequence file that cannot be opened (Scaffolder.cpp:15-18, Main.cpp:357-360): message and exit code 1, whether the
    # files are scanned ahead by the helper process or loaded where the reference loads them
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, "tiny_rr.paf.gz"), G.gunzip_to(tmp, "tiny_cr.paf.gz")
        for env in ({}, {"TLM_NO_FASTA_AHEAD": "1"}, {"TLM_NO_UPLOAD_HELPER": "1"}):
            res = subprocess.run([binary, "--rb-att", "20", rr, cr, "/nonexistent/reads.fa", "/nonexistent/contigs.fa",
                                  os.path.join(tmp, "o.fa")], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
            assert res.returncode == 1, res.stdout[-500:]
            assert "Cannot open file: /nonexistent/reads.fa" in res.stderr or "No consensus found!" in res.stderr


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
    ds = H.dataset_by_name(name)
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


# ---- the device tokeniser (csrc/paf_parse.hip) against the host tokeniser and the token-stream model ----------------

def _host_parse(binary, paths):
    import test_host_ingest as TI
    return TI.run(binary, paths, threads=3)


def test_device_tokeniser_equals_host_tokeniser(binary):
    """Clean PAF text: columns, name ids (first-seen order over the files) and names from the device are those of the
    host tokeniser -- compared through the same checksum `telomeri_hip --parse-only` prints."""
    import test_host_ingest as TI
    from telomeri_amd import capi
    ctx = capi.Context(0)
    with tempfile.TemporaryDirectory() as tmp:
        paths = [G.gunzip_to(tmp, "tiny_rr.paf.gz"), G.gunzip_to(tmp, "tiny_cr.paf.gz"), G.gunzip_to(tmp, "ecoli_cr.paf.gz")]
        recs, modes, n_names, cs = _host_parse(binary, paths)
        paf = capi.PafTokens(ctx)
        got = [paf.add_file(p, anchors=i > 0) for i, p in enumerate(paths)]
        assert [n for n, _ in got] == recs and all(c for _, c in got)
        names, is_ctg = paf.finish()
        files = [paf.columns(i) for i in range(3)]
        assert len(names) == n_names and TI.checksum(files, names) == cs
        assert [bool(x) for x in is_ctg] == [nm.lower().startswith("ctg") for nm in names]
        # and the graph built from the device's records is the graph of the files
        fp = capi.FilterParams.default()
        g = paf.build_graph(fp)
        og = po.Graph()
        pfp = po.FilterParams.default()
        for i, p in enumerate(paths):
            assert og.load_paf(p, i > 0, pfp)
        import helpers as H
        H.assert_csr_equal(g.download(), og.csr())


@pytest.mark.parametrize("name,text,expect_clean", [
    ("clean", None, True),
    ("no_trailing_newline", "strip", True),
    ("crlf", "crlf", True),
    ("ragged_17_tokens", "extra", False),
    ("ragged_15_tokens", "short", False),
    ("number_with_suffix", "suffix", False),
    ("overflow", "overflow", False),
    ("two_char_strand", "strand", False),
    ("blank_line", "blank", False),
    ("empty", "empty", True),
])
def test_device_tokeniser_verdict_on_malformed_text(name, text, expect_clean):
    """What is not 16 clean tokens per line is refused (clean = 0: the host's exact token-stream reader takes over);
    what is accepted equals the token-stream model."""
    import test_host_ingest as TI
    from telomeri_amd import capi
    L = TI.LINE
    body = {None: L * 40, "strip": (L * 5).rstrip("\n"), "crlf": (L * 4).replace("\n", "\r\n"),
            "extra": L * 3 + L.replace("dv:f:0.01", "dv:f:0.01\textra") + L, "short": L * 2 + L.replace("\tdv:f:0.01", "") + L,
            "suffix": L + L.replace("\t6500\t", "\t65x0\t") + L, "overflow": L + L.replace("\t9000\t", "\t99999999999\t"),
            "strand": L + L.replace("\t+\t", "\t+-\t"), "blank": L + "\n" + L, "empty": ""}[text]
    ctx = capi.Context(0)
    paf = capi.PafTokens(ctx)
    n, clean = paf.add_text(body.encode(), anchors=False)
    assert clean == expect_clean
    if clean:
        cols, names = TI.stream_parse(body)
        assert n == len(cols["q_len"])
        got_names, _ = paf.finish()
        assert got_names == names
        if n:
            got = paf.columns(0)
            for k in cols:
                np.testing.assert_array_equal(got[k], np.array(cols[k]), err_msg=k)
    else:
        # a handle that has refused a file is good for nothing but destruction: the refused file's names sit in its
        # dictionary (include/telomeri_hip.h) -- adding more or finishing is an error, not a wrong numbering
        with pytest.raises(capi.TlmError, match="refused"):
            paf.add_text((L * 2).encode(), anchors=False)
        with pytest.raises(capi.TlmError, match="refused"):
            paf.finish()


def test_cli_falls_back_to_the_host_tokeniser(binary):
    """A file that is not clean PAF sends the WHOLE graph back to the host tokeniser: same output as with the device
    tokeniser switched off."""
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, "tiny_rr.paf.gz"), G.gunzip_to(tmp, "tiny_cr.paf.gz")
        # a blank line in the SECOND file: the first one has already been tokenised on the device by then
        txt = open(cr).read().split("\n")
        open(cr, "w").write("\n".join(txt[:3] + [""] + txt[3:]))
        outs = []
        for env in ({}, {"TLM_DEVICE_PARSE": "0"}):
            res = subprocess.run([binary, "--no-scaffold", "--rb-att", "20", rr, cr, "r", "c", "o"], capture_output=True, text=True,
                                 timeout=300, env=dict(os.environ, **env))
            assert res.returncode == 0, res.stderr
            outs.append([ln for ln in res.stdout.splitlines() if ln.startswith(("Loaded", "-", "Found", "Anchor pairs"))])
        assert outs[0] == outs[1] and any(ln.startswith("Loaded") for ln in outs[0])
        res = subprocess.run([binary, "--no-scaffold", "--rb-att", "20", rr, G.gunzip_to(tmp, "tiny_cr.paf.gz"), "r", "c", "o"],
                             capture_output=True, text=True, timeout=300)
        assert "(device)" in res.stdout


@pytest.mark.parametrize("name", ["tiny", "quirks"])
@pytest.mark.parametrize("mode", ["lazy", "full"])
def test_cli_through_the_gathered_representation(binary, name, mode):
    """The host's multi-GPU code path on one GPU (TLM_FORCE_GATHER): paths_ rebuilt from the exchange's result -- the
    gathered summaries with node lists fetched on demand (lazy), or the gathered set (full) -- gives the same report and
    the same scaffold as the one-process path.  quirks has runs of identical paths: filterUnique on the flags."""
    from telomeri_amd import synth
    import helpers as H
    meta = G.load_json(name + ".json")
    ds = H.dataset_by_name(name)
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")
        reads, contigs = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa")
        synth.write_fasta(ds, reads, contigs)
        outs = []
        for env in ({}, {"TLM_FORCE_GATHER": "1", "TLM_EXCHANGE": mode}):
            out = os.path.join(tmp, "o%d.fa" % len(outs))
            res = subprocess.run([binary] + _flags(meta["args"]) + [rr, cr, reads, contigs, out], capture_output=True, text=True,
                                 timeout=300, env=dict(os.environ, **env))
            body = open(out, "rb").read() if os.path.exists(out) else b""
            keep = [ln for ln in res.stdout.splitlines() if ln.startswith(("-", "Paths", "Found", "Consensus", "Final", "ctg", "read",
                                                                           "Anchor pairs"))]
            outs.append((res.returncode, body, keep))
            if env:
                assert "Gathered" in res.stdout
        assert outs[0] == outs[1]


def test_device_tokeniser_long_lines_and_tile_borders():
    """Lines that straddle the 16 KB tiles of the tokeniser kernel and run past the overhang staged in LDS (names of
    several thousand characters) are parsed through the same accessor from HBM; names of 65 535 bytes are refused."""
    import test_host_ingest as TI
    from telomeri_amd import capi
    rng = np.random.default_rng(5)
    lines = []
    for i in range(3000):
        qn = "read%d" % rng.integers(0, 40) if i % 97 else "q" * int(rng.integers(3000, 9000)) + str(i % 3)
        tn = "read%d" % rng.integers(0, 40) if i % 89 else "t" * int(rng.integers(4100, 20000))
        lines.append("%s\t9000\t%d\t8000\t%s\t%s\t7000\t5\t6990\t6500\t7990\t60\ttp:A:S\tcm:i:9\ts1:i:8\tdv:f:0.01\n" % (
            qn, i, "+-"[i % 2], tn))
    body = "".join(lines)
    ctx = capi.Context(0)
    paf = capi.PafTokens(ctx)
    n, clean = paf.add_text(body.encode(), anchors=False)
    assert clean and n == len(lines)
    cols, names = TI.stream_parse(body)
    got_names, _ = paf.finish()
    assert got_names == names
    got = paf.columns(0)
    for k in cols:
        np.testing.assert_array_equal(got[k], np.array(cols[k]), err_msg=k)
    paf = capi.PafTokens(ctx)
    n, clean = paf.add_text((TI.LINE + TI.LINE.replace("readA", "r" * 70000)).encode(), anchors=False)
    assert not clean


def test_device_tokeniser_lines_around_the_overhang():
    """Lines of 900..1150 bytes: the last line of a 16 KB tile ends just inside or just outside the 1 KB overhang that is
    staged with the tile -- both readers of the kernel (all lanes out of LDS; one thread out of HBM) meet the same names,
    in every phase of the line start against the tile end."""
    import test_host_ingest as TI
    from telomeri_amd import capi
    rng = np.random.default_rng(11)
    lines = []
    for i in range(1500):
        qn = "read%d" % (i // 7)
        tn = "t%d_" % rng.integers(0, 30) + "x" * int(rng.integers(830, 1080))
        lines.append("%s\t9000\t%d\t8000\t%s\t%s\t7000\t5\t6990\t6500\t7990\t60\ttp:A:S\tcm:i:9\ts1:i:8\tdv:f:0.01\n" % (
            qn, i, "+-"[i % 2], tn))
    body = "".join(lines)
    ctx = capi.Context(0)
    paf = capi.PafTokens(ctx)
    n, clean = paf.add_text(body.encode(), anchors=False)
    assert clean and n == len(lines)
    cols, names = TI.stream_parse(body)
    got_names, _ = paf.finish()
    assert got_names == names
    got = paf.columns(0)
    for k in cols:
        np.testing.assert_array_equal(got[k], np.array(cols[k]), err_msg=k)


def test_cli_text_upload_through_slabs(binary):
    """The forms of the text upload -- by the helper process through the shared window of device memory (the default), by
    this process out of its own registered mapping, by pread into pinned slabs (files that cannot be mapped) -- feed the
    tokeniser the same bytes."""
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = G.gunzip_to(tmp, "tiny_rr.paf.gz"), G.gunzip_to(tmp, "tiny_cr.paf.gz")
        outs = []
        for env in ({}, {"TLM_NO_UPLOAD_HELPER": "1"}, {"TLM_NO_UPLOAD_HELPER": "1", "TLM_UPLOAD_SLABS": "1"},
                    {"TLM_NO_UPLOAD_HELPER": "1", "TLM_TIDY": "after"}, {"TLM_HELPER_FAIL": "1"}):       # (a helper that refuses)
            res = subprocess.run([binary, "--no-scaffold", "--rb-att", "20", rr, cr, "r", "c", "o"], capture_output=True, text=True,
                                 timeout=300, env=dict(os.environ, **env))
            assert res.returncode == 0 and "(device)" in res.stdout, res.stderr
            outs.append([ln for ln in res.stdout.splitlines() if ln.startswith(("Loaded", "-", "Found", "Anchor pairs"))])
        assert outs[0] == outs[1] == outs[2] == outs[3] == outs[4]


def test_cli_upload_through_the_window_reuses_its_slots(binary):
    """A text of four 1 GiB pieces goes through the three slots of the window the program shares with its helper process:
    piece 3 waits for the word that piece 0 has been copied to its place.  Every line arrives (record count as written)
    and the run equals the one without a helper."""
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), G.gunzip_to(tmp, "tiny_cr.paf.gz")
        lines = []
        for k in range(4096):
            lines.append("r%d\t9000\t10\t8000\t+\tr%d\t9500\t20\t8100\t7000\t8100\t60\ta\tb\tc\td\n" % (k, k + 1))
        block = "".join(lines).encode()
        reps = (3300 << 20) // len(block) + 1
        with open(rr, "wb") as fh:
            for _ in range(reps):
                fh.write(block)
        assert os.path.getsize(rr) > 3 * (1 << 30)
        outs = []
        for env in ({}, {"TLM_NO_UPLOAD_HELPER": "1"}):
            res = subprocess.run([binary, "--no-scaffold", "--rb-att", "2", rr, cr, "r", "c", "o"], capture_output=True, text=True,
                                 timeout=900, env=dict(os.environ, TLM_HOST_TIMING="1", **env))
            assert res.returncode == 0 and "(device)" in res.stdout, res.stderr[-2000:]
            assert ("by the helper process" in res.stderr) == (not env), res.stderr[-2000:]
            m = re.search(r"Loaded (\d+)/(\d+) edges from \S*rr\.paf", res.stdout)
            assert m and int(m.group(2)) == reps * 4096, res.stdout[:2000]
            outs.append([ln for ln in res.stdout.splitlines() if ln.startswith(("Loaded", "-", "Found", "Anchor pairs"))])
        assert outs[0] == outs[1]


@pytest.mark.parametrize("seed", list(range(24)))
def test_device_tokeniser_fuzz(seed):
    """Random 16-column text -- runs of blanks and tabs between tokens, leading blanks, CRLF, signs and leading zeros on
    the numbers, names of 1..40 arbitrary printable characters, a few hundred to a few thousand lines so that several
    16 KB tiles and a last line without a newline occur -- against the token-stream model; every third seed plants one
    defect (a 15- or 17-token line, a non-number, an int overflow, a two-character strand, a blank line): refused."""
    import test_host_ingest as TI
    from telomeri_amd import capi
    rng = np.random.default_rng(4000 + seed)
    alphabet = np.array(list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_.:/|#@!-+=*"))
    pool = ["".join(rng.choice(alphabet, size=int(rng.integers(1, 41)))) for _ in range(int(rng.integers(3, 60)))]
    def sep():
        return "".join(rng.choice(np.array([" ", "\t"]), size=int(rng.integers(1, 4))))
    def num(lo=0, hi=2_000_000_000):
        v = int(rng.integers(lo, hi))
        s = str(v)
        if rng.random() < 0.1:
            s = "0" * int(rng.integers(1, 3)) + s
        if rng.random() < 0.1:
            s = "+" + s
        return s if len(s) <= 10 or s[0] in "+-" and len(s) <= 11 else str(v)
    lines = []
    for i in range(int(rng.integers(300, 2500))):
        toks = [pool[int(rng.integers(0, len(pool)))], num(), num(), num(), "+-"[int(rng.integers(0, 2))],
                pool[int(rng.integers(0, len(pool)))], num(), num(), num(),
                ("-" if rng.random() < 0.05 else "") + str(int(rng.integers(0, 100000))), num(), num(), "tp:A:S", "cm:i:9", "s1:i:8", "dv:f:0.01"]
        lead = sep() if rng.random() < 0.1 else ""
        eol = "\r\n" if rng.random() < 0.2 else "\n"
        lines.append(lead + sep().join(toks) + (sep() if rng.random() < 0.1 else "") + eol)
    defect = seed % 3 == 2
    if defect:
        k = int(rng.integers(0, len(lines)))
        kind = seed // 3 % 5
        parts = lines[k].split()
        if kind == 0:
            lines[k] = "\t".join(parts[:15]) + "\n"
        elif kind == 1:
            lines[k] = "\t".join(parts + ["extra"]) + "\n"
        elif kind == 2:
            lines[k] = "\t".join(parts[:2] + ["12x"] + parts[3:]) + "\n"
        elif kind == 3:
            lines[k] = "\t".join(parts[:1] + ["99999999999"] + parts[2:]) + "\n"
        else:
            lines.insert(k, "  \n")
    body = "".join(lines)
    if seed % 2:
        body = body.rstrip("\r\n")
    ctx = capi.Context(0)
    paf = capi.PafTokens(ctx)
    n, clean = paf.add_text(body.encode(), anchors=False)
    assert clean == (not defect)
    if clean:
        cols, names = TI.stream_parse(body)
        assert n == len(cols["q_len"]) == len(lines)
        got_names, _ = paf.finish()
        assert got_names == names
        got = paf.columns(0)
        for k in cols:
            np.testing.assert_array_equal(got[k], np.array(cols[k]), err_msg=k)
