"""SURVEY.md §8f row 2, host logic (no GPU): the C++ host's PAF tokeniser (`telomeri_hip --parse-only`) must read a
file exactly as the reference's operator>> loop does (OverlapGraph.cpp:109-127): one token stream, 12 typed tokens +
4 discarded ones per record, silently ending at the first token that does not parse.  Clean files take the parallel
line-wise path, anything else the sequential token-stream path; both must agree with the restatement below, which
is itself checked against the oracle's loader (pinned to the reference) through the record counts."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as G
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_hip")


@pytest.fixture(scope="module")
def binary():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "telomeri_amd", "host")], check=True)
    return BIN


def stream_parse(text: str):
    """operator>> semantics on the whole file as one token stream -> (columns, names)."""
    ws = " \t\n\r\v\f"
    pos, n = 0, len(text)

    def skip():
        nonlocal pos
        while pos < n and text[pos] in ws:
            pos += 1

    def token():
        nonlocal pos
        skip()
        if pos >= n:
            return None
        a = pos
        while pos < n and text[pos] not in ws:
            pos += 1
        return text[a:pos]

    def integer():
        nonlocal pos
        skip()
        if pos >= n:
            return None
        q = pos
        if text[q] in "+-":
            q += 1
        d0 = q
        while q < n and text[q].isdigit() and text[q].isascii():
            q += 1
        if q == d0:
            return None
        v = int(text[pos:q])
        pos = q
        if v > 2 ** 31 - 1 or v < -2 ** 31:
            return None
        return v

    def char():
        nonlocal pos
        skip()
        if pos >= n:
            return None
        c = text[pos]
        pos += 1
        return c

    names, ids = [], {}
    cols = {k: [] for k in ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches", "strand")}
    while True:
        qn = token()
        if qn is None:
            break
        f = [integer() for _ in range(0)]
        ql, qs, qe = integer(), None, None
        if ql is None:
            break
        qs = integer()
        if qs is None:
            break
        qe = integer()
        if qe is None:
            break
        st = char()
        if st is None:
            break
        tn = token()
        if tn is None:
            break
        rest = []
        bad = False
        for _ in range(6):
            v = integer()
            if v is None:
                bad = True
                break
            rest.append(v)
        if bad:
            break
        for _ in range(4):
            token()
        for key, nm in (("q_id", qn), ("t_id", tn)):
            if nm not in ids:
                ids[nm] = len(names)
                names.append(nm)
            cols[key].append(ids[nm])
        cols["q_len"].append(ql); cols["q_start"].append(qs); cols["q_end"].append(qe)
        cols["t_len"].append(rest[0]); cols["t_start"].append(rest[1]); cols["t_end"].append(rest[2])
        cols["matches"].append(rest[3])
        cols["strand"].append(1 if st == "-" else 0)
    return cols, names


def checksum(files, names) -> int:
    h = 1469598103934665603
    M = (1 << 64) - 1

    def mix(b: bytes):
        nonlocal h
        for x in b:
            h = ((h ^ x) * 1099511628211) & M

    for cols in files:
        for k in ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches"):
            mix(np.array(cols[k], dtype=np.int32).tobytes())
        mix(np.array(cols["strand"], dtype=np.uint8).tobytes())
    for nm in names:
        mix(nm.encode() + b"\0")
    return h


def run(binary, paths, threads=None):
    env = dict(os.environ)
    if threads:
        env["TLM_PARSE_THREADS"] = str(threads)
    res = subprocess.run([binary, "--parse-only"] + paths, capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0, res.stderr
    recs = [int(ln.split()[1]) for ln in res.stdout.splitlines() if ln.startswith("parsed ")]
    modes = [ln.rsplit("(", 1)[1].rstrip(")") for ln in res.stdout.splitlines() if ln.startswith("parsed ")]
    last = res.stdout.strip().splitlines()[-1].split()
    return recs, modes, int(last[1]), int(last[3])


LINE = "readA\t9000\t10\t8000\t+\treadB\t7000\t5\t6990\t6500\t7990\t60\ttp:A:S\tcm:i:9\ts1:i:8\tdv:f:0.01\n"


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_clean_files_parallel_and_sequential_agree(binary, threads):
    with tempfile.TemporaryDirectory() as tmp:
        paths = [G.gunzip_to(tmp, "tiny_rr.paf.gz"), G.gunzip_to(tmp, "tiny_cr.paf.gz"), G.gunzip_to(tmp, "ecoli_cr.paf.gz")]
        files, names, ids = [], [], {}
        total = []
        for p in paths:
            cols, nm = stream_parse(open(p).read())
            # one name table over all files, first-seen order
            remap = []
            for x in nm:
                if x not in ids:
                    ids[x] = len(names)
                    names.append(x)
                remap.append(ids[x])
            cols["q_id"] = [remap[i] for i in cols["q_id"]]
            cols["t_id"] = [remap[i] for i in cols["t_id"]]
            files.append(cols)
            total.append(len(cols["q_len"]))
        recs, modes, n_names, cs = run(binary, paths, threads)
        assert recs == total and recs[2] == 24127
        assert all(m == "parallel" for m in modes)
        assert n_names == len(names) and cs == checksum(files, names)
        # the oracle's loader (pinned to the reference) counts the same records
        g = po.Graph()
        fp = po.FilterParams.default()
        for p in paths:
            assert g.load_paf(p, False, fp)
        assert g.records == total


@pytest.mark.parametrize("name,text", [
    ("ragged_17_tokens", LINE + LINE.replace("dv:f:0.01", "dv:f:0.01\textra") + LINE * 3),
    ("ragged_15_tokens", LINE * 2 + LINE.replace("\tdv:f:0.01", "") + LINE * 2),
    ("number_with_suffix", LINE + LINE.replace("\t6500\t", "\t65x0\t") + LINE),
    ("overflow", LINE + LINE.replace("\t9000\t", "\t99999999999\t") + LINE),
    ("two_char_strand", LINE + LINE.replace("\t+\t", "\t+-\t") + LINE),
    ("no_trailing_newline", (LINE * 3).rstrip("\n")),
    ("truncated_last_record", LINE * 3 + "readC\t100\t1"),
    ("blank_lines_and_crlf", LINE + "\n\n" + LINE.replace("\n", "\r\n") + "   \n" + LINE),
    ("negative_and_plus_signs", LINE.replace("\t10\t", "\t-10\t").replace("\t5\t", "\t+5\t") * 2),
    ("empty", ""),
    ("records_split_over_lines", LINE.replace("\t+\t", "\n+\n") * 2),
])
def test_token_stream_quirks(binary, name, text):
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, name + ".paf")
        open(p, "w", newline="").write(text)
        cols, names = stream_parse(text)
        for threads in (1, 4):
            recs, modes, n_names, cs = run(binary, [p], threads)
            assert recs == [len(cols["q_len"])], (name, threads, recs)
            assert n_names == len(names) and cs == checksum([cols], names), (name, threads)
        g = po.Graph()
        assert g.load_paf(p, False, po.FilterParams.default())
        assert g.records == [len(cols["q_len"])]


def test_test_load_num(binary):
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "x.paf")
        open(p, "w").write(LINE * 10)
        res = subprocess.run([binary, "--parse-only", "--test-load-num", "3", p], capture_output=True, text=True)
        assert "parsed 4 records" in res.stdout and "(sequential)" in res.stdout     # OverlapGraph.cpp:145: breaks after 3 < ctr
        g = po.Graph()
        assert g.load_paf(p, False, po.FilterParams.default(), test_load_num=3) and g.records == [4]


def test_run_sh_contract(tmp_path):
    """reference run.sh:14-47: usage text without an argument; overlap files are reused when at least one exists; the
    program gets <dir>/rr_overlaps.paf, cr_overlaps.paf, reads.fast?, contigs.fast?, HERA_result.fasta."""
    import subprocess
    run = os.path.join(ROOT, "run.sh")
    res = subprocess.run(["bash", run], capture_output=True, text=True)
    assert res.returncode == 1 and "Please provide the dataset directory path!" in res.stdout
    assert "Usage: ./run.sh <dataset-dir>" in res.stdout
    d = tmp_path / "ds"
    d.mkdir()
    res = subprocess.run(["bash", run, str(d)], capture_output=True, text=True)
    assert res.returncode == 1 and "no ./minimap2/minimap2" in res.stderr          # both PAFs missing: would be built
    (d / "cr_overlaps.paf").write_text("")
    (d / "reads.fasta").write_text(">r\nA\n")
    (d / "contigs.fastq").write_text("@c\nA\n+\nI\n")
    res = subprocess.run(["bash", run, str(d)], capture_output=True, text=True)
    assert "Using existing overlap files!" in res.stdout                           # one of the two exists: reused as is
    assert res.returncode == 1 and "Cannot open file: %s" % (d / "rr_overlaps.paf") in res.stderr
