"""SURVEY.md §8f rows 3-4 (host logic, no GPU): grouping, consensus, scaffold-path chaining and the Scaffolder of the
C++ host (telomeri_amd/host: consensus.cpp, scaffolder.cpp, pipeline.cpp), run by `telomeri_post` on the reference's
own filtered path list (the `U` section of a ref_dump file), must write the reference's scaffold FASTA byte for byte
and report the reference's groups.
  * golden: tests/golden/<set>.json holds what the unmodified reference program wrote for the committed inputs;
  * live (only where oracle/_ref exists): fresh datasets and grouping parameters."""
import hashlib
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as G
import helpers as H
from oracle import pyoracle as po
from telomeri_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POST = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_post")


@pytest.fixture(scope="module")
def post():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "telomeri_amd", "host")], check=True)
    return POST


def _dataset(name):
    return H.dataset_by_name(name)


def _grouping_flags(args):
    """The flags of a set that the host stages (not the walks) read: telomeri_post takes them as the reference's main() does."""
    out = []
    for k, a in enumerate(args):
        if a in ("--len-thr", "--w-size", "--r-thr"):
            out += [a, args[k + 1]]
    return out


@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks", "hard_tiny", "overflow"])
def test_scaffold_and_groups_against_golden(post, name):
    meta = G.load_json(name + ".json")
    ds = _dataset(name)
    post = [post] + _grouping_flags(meta["args"])
    with tempfile.TemporaryDirectory() as tmp:
        dump = G.gunzip_to(tmp, name + ".dump.gz")
        reads, contigs, out = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "out.fa")
        synth.write_fasta(ds, reads, contigs)
        res = subprocess.run(post + [dump, reads, contigs, out], capture_output=True, text=True, timeout=120)
        want = meta["scaffold"]
        assert res.returncode == want["exit"], res.stderr
        body = open(out, "rb").read()
        assert len(body) == want["bytes"] and hashlib.md5(body).hexdigest() == want["md5"]
        # the sequence files as telomeri_hip gets them from its helper process (scanned by one Scaffolder, read from its
        # index by another): the same scaffold
        out2 = os.path.join(tmp, "out2.fa")
        res2 = subprocess.run(post + [dump, reads, contigs, out2], capture_output=True, text=True, timeout=120,
                              env=dict(os.environ, TLM_POST_VIA_INDEX="1"))
        assert res2.returncode == want["exit"], res2.stderr
        assert open(out2, "rb").read() == body
        chain = [ln for ln in res.stdout.splitlines() if ln.startswith(("ctg", "read")) and "-" in ln]
        assert hashlib.md5(chain[-1].encode()).hexdigest() == want["chain_md5"]
        got_lines = [ln for ln in res.stdout.splitlines() if ln.startswith(("Consensus length", "Final consensus"))]
        assert got_lines == want["consensus_lines"]
        for params, report in meta["groups"].items():
            g_out = os.path.join(tmp, "groups.txt")
            subprocess.run([post[0], "--groups", g_out] + params.split() + [dump, "x", "y", "z"], check=True, capture_output=True)
            assert open(g_out).read() == report, params


live = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


@live
@pytest.mark.parametrize("seed,args", [(41, ["--rb-att", "50"]), (42, ["--filter-min", "--rb-att", "40", "--bt-att", "4"]),
                                       (43, ["--rb-att", "70", "--w-size", "300"])])
def test_scaffold_against_live_reference(post, seed, args):
    ds = synth.make_dataset("fresh", seed=seed, genome_len=220_000, n_contigs=3, contig_len=32_000, n_reads=300,
                            read_len_lo=4_000, read_len_hi=12_000)
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), os.path.join(tmp, "cr.paf")
        reads, contigs = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa")
        synth.write_paf(ds, ds.rr, rr)
        synth.write_paf(ds, ds.cr, cr)
        synth.write_fasta(ds, reads, contigs, seed=seed)
        ref_out, my_out, dump = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "mine.fa"), os.path.join(tmp, "d.dump")
        ref = subprocess.run([po.REF_MAIN] + args + [rr, cr, reads, contigs, ref_out], capture_output=True, text=True)
        walk_args = list(args)
        if "--w-size" in walk_args:         # a grouping parameter only: the dump driver takes it with --groups
            k = walk_args.index("--w-size")
            del walk_args[k:k + 2]
        subprocess.run([po.REF_DUMP] + walk_args + [rr, cr, dump], check=True, stdout=subprocess.DEVNULL)
        post_args = []
        if "--w-size" in args:
            post_args = ["--w-size", args[args.index("--w-size") + 1]]
        mine = subprocess.run([post] + post_args + [dump, reads, contigs, my_out], capture_output=True, text=True)
        assert mine.returncode == ref.returncode
        if ref.returncode == 0:
            assert open(my_out, "rb").read() == open(ref_out, "rb").read()
            pick = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(("Consensus length", "Final consensus", "ctg", "read", "===="))]
            assert pick(mine.stdout) == pick(ref.stdout)
        for gp in (("1500", "400", "0.9"), ("300", "100", "0.6")):
            gd, mg = os.path.join(tmp, "g.dump"), os.path.join(tmp, "g.mine")
            subprocess.run([po.REF_DUMP] + walk_args + ["--no-graph", "--groups", *gp, rr, cr, gd], check=True, stdout=subprocess.DEVNULL)
            subprocess.run([post, "--groups", mg, *gp, dump, "x", "y", "z"], check=True, capture_output=True)
            txt = open(gd).read()
            assert open(mg).read() == txt[txt.index("\nG ") + 1:]


@live
def test_scaffold_against_live_reference_mid_size(post):
    """f3 / f4 against the unmodified reference where its O(N) nodeIndex scan still allows it: 16 000 reads, 30 contigs,
    456 000 overlap records, 29 anchor pairs, ~4 900 paths, a 6 MB scaffold (VERDICT r3 "weak" 1: the host stages were pinned
    at <= 300 reads; the reference needs 21 s for the load alone here).  telomeri_post, fed with the reference's own path
    list, must print the reference's consensus report and write its scaffold byte for byte."""
    ds = synth.make_dataset("mid", seed=61, genome_len=9_000_000, n_contigs=30, contig_len=60_000, n_reads=16_000,
                            read_len_lo=4_000, read_len_hi=14_000)
    args = ["--rb-att", "200"]
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), os.path.join(tmp, "cr.paf")
        reads, contigs = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa")
        synth.write_paf(ds, ds.rr, rr)
        synth.write_paf(ds, ds.cr, cr)
        synth.write_fasta(ds, reads, contigs, seed=61)
        ref_out, my_out, dump = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "mine.fa"), os.path.join(tmp, "d.dump")
        ref_run = subprocess.Popen([po.REF_MAIN] + args + [rr, cr, reads, contigs, ref_out], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        subprocess.run([po.REF_DUMP] + args + [rr, cr, dump], check=True, stdout=subprocess.DEVNULL)       # (beside the reference program)
        ref_stdout = ref_run.communicate()[0]
        assert ref_run.returncode == 0
        mine = subprocess.run([post, dump, reads, contigs, my_out], capture_output=True, text=True)
        assert mine.returncode == 0, mine.stderr[-2000:]
        n_paths = int([ln for ln in ref_stdout.splitlines() if ln.startswith("- total_num:")][0].split()[-1])
        assert n_paths > 3000
        pick = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(("Consensus length", "Final consensus", "ctg", "read", "===="))]
        assert pick(mine.stdout) == pick(ref_stdout) and len(pick(ref_stdout)) > 29
        a, b = open(my_out, "rb").read(), open(ref_out, "rb").read()
        assert len(b) > 5_000_000 and a == b


@live
def test_int_accumulator_overflow_against_live_reference(post):
    """PathGroup.cpp:66-68: std::accumulate(..., 0, ...) adds path lengths into an int.  With ~40 Mb paths it wraps after 54
    of them; what the reference then picks as "the first path at least as long as the average" is whatever the wrapped
    average says -- telomeri_post must pick the same (another seed than the golden set's)."""
    ds = H.overflow_dataset(seed=52)
    args = ["--len-thr", "100000000", "--rb-att", "120"]
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), os.path.join(tmp, "cr.paf")
        reads, contigs = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa")
        synth.write_paf(ds, ds.rr, rr)
        synth.write_paf(ds, ds.cr, cr)
        synth.write_fasta(ds, reads, contigs, seed=52)
        ref_out, my_out, dump = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "mine.fa"), os.path.join(tmp, "d.dump")
        ref = subprocess.run([po.REF_MAIN] + args + [rr, cr, reads, contigs, ref_out], capture_output=True, text=True)
        subprocess.run([po.REF_DUMP] + args + [rr, cr, dump], check=True, stdout=subprocess.DEVNULL)
        mine = subprocess.run([post, "--len-thr", "100000000", dump, reads, contigs, my_out], capture_output=True, text=True)
        assert ref.returncode == 0 == mine.returncode, ref.stderr + mine.stderr
        # the accumulator did overflow: more than 2^31 / 40 Mb paths between one pair of anchors
        n_paths = int([ln for ln in ref.stdout.splitlines() if ln.startswith("- total_num:")][0].split()[-1])
        assert n_paths * 40_000_000 > 2 ** 33
        pick = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(("Consensus length", "Final consensus", "ctg", "read", "===="))]
        assert pick(mine.stdout) == pick(ref.stdout) and len(pick(ref.stdout)) > 8
        assert open(my_out, "rb").read() == open(ref_out, "rb").read()


def test_scaffolder_fastq_and_single_node(post):
    """FASTQ input (two skipped lines per record, Scaffolder.cpp:13,33-36) gives the same scaffold as FASTA."""
    meta = G.load_json("tiny.json")
    ds = synth.make_config("tiny")
    with tempfile.TemporaryDirectory() as tmp:
        dump = G.gunzip_to(tmp, "tiny.dump.gz")
        reads, contigs, out = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "out.fa")
        synth.write_fasta(ds, reads, contigs)
        fq = os.path.join(tmp, "reads.fastq")
        with open(reads) as fi, open(fq, "w") as fo:
            lines = fi.read().splitlines()
            for i in range(0, len(lines), 2):
                fo.write("@%s some comment\n%s\n+\n%s\n" % (lines[i][1:], lines[i + 1], "I" * len(lines[i + 1])))
        res = subprocess.run([post, dump, fq, contigs, out], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["scaffold"]["md5"]
        res = subprocess.run([post, dump, os.path.join(tmp, "missing.fa"), contigs, out], capture_output=True, text=True)
        assert res.returncode == 1 and "Cannot open file" in res.stderr


def _model_load(text: str, fastq: bool):
    """Scaffolder::load (Scaffolder.cpp:12-41) as a model: lines take their role from their number alone."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                       # getline: no line after the final newline
    period = 4 if fastq else 2
    names, seqs = [], []
    for k, ln in enumerate(lines):
        if k % period == 0:
            tok = ln.split()              # first whitespace-separated token minus its first character
            names.append(tok[0][1:] if tok else ln[1:])
        elif k % period == 1:
            seqs.append(ln)
    return names, seqs


@pytest.mark.parametrize("name,fastq,text", [
    ("plain", False, ">r1 extra words\nACGT\n>r2\nGGA\n"),
    ("odd_line_count", False, ">r1\nACGT\n>r2\n"),
    ("no_final_newline", False, ">r1\nACGT\n>r2\nTT"),
    ("crlf_and_blank_sequence", False, ">r1\r\nAC\r\n>r2\n\n>r3\nT\n"),
    ("fastq", True, "@r1 c\nACGT\n+\nIIII\n@r2\nGG\n+\nII\n"),
    ("fastq_truncated", True, "@r1\nACGT\n+\nIIII\n@r2\nGG\n"),
    ("multi_line_fasta_is_misread_as_the_reference_does", False, ">r1\nAC\nGT\n>r2\nTT\n"),
    ("single_name", False, ">only\n"),
])
def test_scaffolder_load_roles_by_line_number(post, name, fastq, text):
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "x.fastq" if fastq else "x.fasta")
        open(p, "w", newline="").write(text)
        big = os.path.join(tmp, "big.fasta")            # large enough for several parser threads, roles still by number
        with open(big, "w") as fh:
            for i in range(120_000):
                fh.write(">read%d tail\n%s\n" % (i, "ACGT" * (5 + i % 37)))
        for paths, env in (([p], {}), ([big, p], {}), ([big, p], {"TLM_POST_VIA_INDEX": "1"})):
            res = subprocess.run([post, "--list-sequences"] + paths, capture_output=True, timeout=120, env=dict(os.environ, **env))
            assert res.returncode == 0, res.stderr
            res.stdout = res.stdout.decode()            # (no newline translation: a sequence may end in \r)
            names, seqs = [], []
            for q in paths:
                n, s = _model_load(open(q, newline="").read(), q.endswith("q"))
                names += n
                seqs += s
            out = res.stdout.split("\n")
            assert out[0] == "names %d sequences %d" % (len(names), len(seqs))
            body = res.stdout[len(out[0]) + 1:]
            want = "".join("%d\t[%s]%s\n" % (i, nm, "\t[%s]" % seqs[i] if i < len(seqs) else "") for i, nm in enumerate(names))
            assert body == want


@pytest.mark.parametrize("name", ["tiny", "quirks"])
def test_binary_dump_gives_the_text_dump_result(post, name):
    """`telomeri_post --binary <dir>` (raw CSR arrays + packed Path set: the input of the full-size comparison in
    tests/test_gpu_full_size.py) is the text dump in another encoding: same report, same scaffold bytes as the reference wrote."""
    import types
    meta = G.load_json(name + ".json")
    ds = _dataset(name)
    with tempfile.TemporaryDirectory() as tmp:
        ref = po.parse_ref_dump(G.gunzip_to(tmp, name + ".dump.gz"))
        adj = ref.adj                                   # rows: owner slot nbr q_start q_end t_start t_end os si es strand
        deg = ref.node_deg
        row_ptr = np.zeros(len(ref.node_name) + 1, np.uint64)
        row_ptr[1:] = np.cumsum(deg)
        order = np.lexsort((adj[:, 1], adj[:, 0]))
        a = adj[order]
        csr = types.SimpleNamespace(n_nodes=len(ref.node_name), n_adj=a.shape[0], node_len=ref.node_len, node_anchor=ref.node_anchor,
                                    row_ptr=row_ptr, nbr=a[:, 2], q_start=a[:, 3], q_end=a[:, 4], t_start=a[:, 5], t_end=a[:, 6],
                                    os=a[:, 7].astype(np.uint32).view(np.float32), si=a[:, 8].astype(np.uint32).view(np.float32),
                                    es=a[:, 9].astype(np.uint32).view(np.float32), strand=a[:, 10])
        counts = np.array([len(n) for _, n, _ in ref.unique], np.int64)
        off = np.zeros(len(counts) + 1, np.uint64)
        off[1:] = np.cumsum(counts)
        nodes = np.concatenate([n for _, n, _ in ref.unique]).astype(np.uint32)
        slots = np.concatenate([np.r_[s, np.uint32(0xFFFFFFFF)] for _, _, s in ref.unique]).astype(np.uint32)
        paths = types.SimpleNamespace(off=off, length=np.array([l for l, _, _ in ref.unique], np.int64), nodes=nodes, slots=slots)
        H.write_binary_dump(os.path.join(tmp, "bin"), ref.node_name, csr, paths)
        reads, contigs, out = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "out.fa")
        synth.write_fasta(ds, reads, contigs)
        res = subprocess.run([post, "--binary", os.path.join(tmp, "bin"), reads, contigs, out], capture_output=True, text=True, timeout=120)
        want = meta["scaffold"]
        assert res.returncode == want["exit"], res.stderr
        body = open(out, "rb").read()
        assert len(body) == want["bytes"] and hashlib.md5(body).hexdigest() == want["md5"]
        got_lines = [ln for ln in res.stdout.splitlines() if ln.startswith(("Consensus length", "Final consensus"))]
        assert got_lines == want["consensus_lines"]
        # a truncated array is refused, not read short
        with open(os.path.join(tmp, "bin", "path_nodes.u32"), "r+b") as fh:
            fh.truncate(max(0, nodes.nbytes - 4))
        bad = subprocess.run([post, "--binary", os.path.join(tmp, "bin"), reads, contigs, out], capture_output=True, text=True, timeout=120)
        assert bad.returncode == 1 and "expected" in bad.stderr


def test_scaffold_into_a_pipe(post):
    """The scaffold file is normally grown to its size and mapped (the threads copy their pieces into the page cache); a
    destination that cannot be mapped -- here a FIFO with a reader behind it -- gets the same bytes through the stream."""
    import threading
    meta = G.load_json("tiny.json")
    ds = _dataset("tiny")
    with tempfile.TemporaryDirectory() as tmp:
        dump = G.gunzip_to(tmp, "tiny.dump.gz")
        reads, contigs, out = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "out.fifo")
        synth.write_fasta(ds, reads, contigs)
        os.mkfifo(out)
        got = []
        reader = threading.Thread(target=lambda: got.append(open(out, "rb").read()))
        reader.start()
        res = subprocess.run([post, dump, reads, contigs, out], capture_output=True, text=True, timeout=120)
        reader.join(timeout=60)
        want = meta["scaffold"]
        assert res.returncode == want["exit"], res.stderr
        assert got and len(got[0]) == want["bytes"] and hashlib.md5(got[0]).hexdigest() == want["md5"]


def test_sequence_index_file_is_created_exclusively(post, tmp_path):
    """The index a Scaffolder leaves for another process (Scaffolder::save_index; the upload helper writes it into /dev/shm)
    is created O_EXCL | O_NOFOLLOW with mode 0600: a link or a file somebody put at its name is not written through."""
    fa = tmp_path / "reads.fa"
    fa.write_text(">r1\nACGT\n>r2\nGGCC\n")
    env = dict(os.environ, TLM_POST_VIA_INDEX="1")
    ok = subprocess.run([post, "--list-sequences", str(fa)], capture_output=True, text=True, env=env)
    assert ok.returncode == 0 and "names 2 sequences 2" in ok.stdout
    assert not os.path.exists(str(fa) + ".index")                  # removed after use
    victim = tmp_path / "victim"
    victim.write_text("precious")
    os.symlink(victim, str(fa) + ".index")
    bad = subprocess.run([post, "--list-sequences", str(fa)], capture_output=True, text=True, env=env)
    assert bad.returncode == 1
    assert victim.read_text() == "precious"
