"""INTEGRATION.md option A as a program: the reference's UNMODIFIED src/Main.cpp compiled against the headers of
telomeri_amd/host/include (OverlapGraph.hpp, Path.hpp, PathGroup.hpp, PathWindow.hpp, PathManager.hpp, Scaffolder.hpp,
Utils.hpp, Stopwatch.hpp -- the reference's file names) and linked with the host classes + libtelomeri_hip.so
(`make -C oracle ref` -> oracle/_ref/Telomeri_on_hip; build container only, the binary travels to the GPU box).

CPU tier: the headers stand alone; the recipe is -I and a link line, nothing else; the program links.
GPU tier: the program's report and scaffold against
  * the UNMODIFIED reference program itself (tests/golden/<set>.json `program_det_only`, the bundled fixture's `program`):
    with --rb-att 0 everything main() prints and writes depends on the deterministic builders, which the device
    reproduces bit for bit -- same stdout, same exit code, same scaffold bytes as the reference binary;
  * telomeri_hip --materialise under the sets' own arguments (per-walk-seeded Monte-Carlo): same report, same bytes.
"""
import glob
import hashlib
import os
import re
import subprocess
import tempfile

import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "telomeri_amd", "host")
ON_HIP = os.path.join(ROOT, "oracle", "_ref", "Telomeri_on_hip")
HIP = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_hip")
REF_SRC = "/root/reference/src/Main.cpp"

needs_reference = pytest.mark.skipif(not os.path.exists(REF_SRC), reason="the reference's sources exist in the build container only")
needs_binary = pytest.mark.skipif(not os.path.exists(ON_HIP), reason="oracle/_ref/Telomeri_on_hip was not built (make -C oracle ref)")


# ---- CPU tier ------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("header", sorted(os.path.basename(h) for h in glob.glob(os.path.join(HOST, "include", "*.hpp"))))
def test_header_stands_alone(header):
    """Every header compiles as the first and only include of a translation unit (the reference's main() includes
    <PathManager.hpp>, <Stopwatch.hpp>, <Scaffolder.hpp> in that order, Main.cpp:3-5; any order must do)."""
    src = "#include <%s>\nint main() { return 0; }\n" % header
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(HOST, "include"), "-x", "c++", "-"],
                         input=src, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_reference_header_names_are_all_there():
    """One header per reference header, under the reference's name (include/*.hpp of the reference; names are a contract,
    listed here so that the test runs where the reference does not exist)."""
    want = {"OverlapGraph.hpp", "Path.hpp", "PathGroup.hpp", "PathManager.hpp", "PathWindow.hpp", "Scaffolder.hpp", "Stopwatch.hpp",
            "Utils.hpp"}
    have = {os.path.basename(h) for h in glob.glob(os.path.join(HOST, "include", "*.hpp"))}
    assert want <= have
    if os.path.isdir("/root/reference/include"):
        assert {os.path.basename(h) for h in glob.glob("/root/reference/include/*.hpp")} <= have


@needs_reference
def test_recipe_is_include_path_and_link_line_only():
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-n", "-B", os.path.join(ROOT, "oracle", "_ref", "Telomeri_on_hip")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    cmd = [ln for ln in res.stdout.splitlines() if "Main.cpp" in ln]
    assert len(cmd) == 1, res.stdout
    words = cmd[0].split()
    assert REF_SRC in words                                             # the file where it lies, not a copy
    assert [w for w in words if w.startswith("-I")] == ["-I" + os.path.join(HOST, "include")]
    assert not [w for w in words if w.startswith(("-D", "-include", "-imacros", "-U", "-fpermissive"))], cmd[0]
    assert not [w for w in words if "/root/reference" in w and w != REF_SRC], cmd[0]     # no other reference source
    srcs = sorted(os.path.basename(w) for w in words if w.endswith(".cpp"))
    assert srcs == ["Main.cpp", "consensus.cpp", "overlap_graph.cpp", "path_manager.cpp", "scaffolder.cpp"]


@needs_reference
def test_unmodified_main_compiles_and_links():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    assert os.path.exists(ON_HIP)
    ldd = subprocess.run(["ldd", ON_HIP], capture_output=True, text=True).stdout
    m = re.search(r"libtelomeri_hip\.so => (\S+)", ldd)
    assert m and os.path.realpath(m.group(1)) == os.path.realpath(os.path.join(ROOT, "telomeri_amd", "lib", "libtelomeri_hip.so")), ldd
    assert "oracle" not in ldd.replace(os.path.join(ROOT, "oracle", "_ref"), "")
    # the hot path's members come from the host classes (no reference object file is in the link: the recipe test above)
    syms = subprocess.run(["nm", "-C", "--defined-only", ON_HIP], capture_output=True, text=True).stdout
    for member in ("OverlapGraph::load(char*, bool)", "PathManager::buildMonteCarlo(OverlapGraph const&, Utils::Metrics const&)",
                   "PathManager::buildDeterministic(OverlapGraph const&, Utils::Metrics const&)", "Scaffolder::write(char const*)"):
        assert member in syms, member
    und = subprocess.run(["nm", "-C", "--undefined-only", ON_HIP], capture_output=True, text=True).stdout
    assert "tlm_walk_mc" in und and "tlm_walk_det" in und and "tlm_graph_build" in und


# ---- GPU tier ------------------------------------------------------------------------------------------------------

def _normalise(text):
    """As tests/golden/make_golden.py::normalise_report."""
    out = []
    for ln in text.splitlines():
        if ln.startswith("Done (") or ln.startswith("Total time:"):
            continue
        for key in ("Loaded ", "Writing result to: "):
            if ln.startswith(key) and "/" in ln:
                head, _, path = ln.rpartition(" ")
                ln = head + " " + os.path.basename(path)
        out.append(ln)
    return out


def _inputs(tmp, name):
    from telomeri_amd import synth
    import helpers as H
    ds = H.dataset_by_name(name)
    rr, cr = G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")
    reads, contigs = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa")
    synth.write_fasta(ds, reads, contigs)
    return [rr, cr, reads, contigs]


@pytest.mark.gpu
@needs_binary
@pytest.mark.parametrize("lean", [False, True])
@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks", "hard_tiny", "overflow"])
def test_report_and_scaffold_equal_the_reference_program(name, lean):
    want = G.load_json(name + ".json")["program_det_only"]
    with tempfile.TemporaryDirectory() as tmp:
        files = _inputs(tmp, name)
        out = os.path.join(tmp, "det.fa")       # (the name the golden run wrote to: it is part of the report)
        res = subprocess.run([ON_HIP] + want["args"] + files + [out], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, TLM_HOST_LEAN="1" if lean else "0"))
        assert _normalise(res.stdout) == want["stdout"]
        assert res.returncode == want["exit"], res.stderr
        if want["exit"] == 0:
            body = open(out, "rb").read()
            assert len(body) == want["bytes"] and hashlib.md5(body).hexdigest() == want["md5"]
        else:       # (quirks: the reference ends in an uncaught std::out_of_range of map::at -- so does its main() here)
            assert [ln for ln in _normalise(res.stderr) if not ln.startswith("[")] == want["stderr"]


def _mid_inputs(tmp):
    """The 16 000-read dataset of tests/golden/mid.json, regenerated from its seed (no fixture is stored for it); the text
    must be the text the golden run was made on."""
    from telomeri_amd import synth
    meta = G.load_json("mid.json")
    ds = synth.make_dataset("mid", **meta["dataset"])
    files = [os.path.join(tmp, n) for n in ("mid_rr.paf", "mid_cr.paf", "mid_reads.fa", "mid_contigs.fa")]
    synth.write_paf(ds, ds.rr, files[0])
    synth.write_paf(ds, ds.cr, files[1])
    synth.write_fasta(ds, files[2], files[3])
    for f in files:
        assert hashlib.md5(open(f, "rb").read()).hexdigest() == meta["inputs_md5"][os.path.basename(f)], f
    return meta, files


def test_mid_size_inputs_are_reproducible():
    """CPU tier: the generator gives the bytes the reference was run on (tests/golden/make_golden.py --mid)."""
    with tempfile.TemporaryDirectory() as tmp:
        meta, files = _mid_inputs(tmp)
        assert meta["records"] == [455862, 3642] and os.path.getsize(files[0]) > 40_000_000


@pytest.mark.gpu
@needs_binary
@pytest.mark.parametrize("lean", [False, True])
def test_mid_size_report_and_scaffold_equal_the_reference_program(lean):
    """The reference's own main() on the HIP path at sixty times the size of the other sets: 16 000 reads, 30 contigs,
    456 000 records through the device tokeniser, filter, graph build, the deterministic kernels (--rb-att 0), the post-passes
    and the host stages -- the whole report and the 352 KB scaffold equal what the UNMODIFIED reference binary printed and
    wrote (it needs 25 s for it; frozen in tests/golden/mid.json)."""
    with tempfile.TemporaryDirectory() as tmp:
        meta, files = _mid_inputs(tmp)
        want = meta["program_det_only"]
        out = os.path.join(tmp, "det.fa")
        res = subprocess.run([ON_HIP] + want["args"] + files + [out], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, TLM_HOST_LEAN="1" if lean else "0"))
        assert res.returncode == want["exit"] == 0, res.stderr[-2000:]
        assert _normalise(res.stdout) == want["stdout"]
        body = open(out, "rb").read()
        assert len(body) == want["bytes"] and hashlib.md5(body).hexdigest() == want["md5"]


@pytest.mark.gpu
@needs_binary
def test_bundled_fixture_report_equals_the_reference_program():
    want = G.load_json("ecoli_cr_known.json")["program"]
    with tempfile.TemporaryDirectory() as tmp:
        paf = G.gunzip_to(tmp, "ecoli_cr.paf.gz")
        os.rename(paf, os.path.join(tmp, "EColi_CR_overlaps.paf"))      # (the file's name is part of the report)
        paf = os.path.join(tmp, "EColi_CR_overlaps.paf")
        empty, none = os.path.join(tmp, "empty.paf"), os.path.join(tmp, "none.fa")
        open(empty, "w").close()
        open(none, "w").close()
        res = subprocess.run([ON_HIP, empty, paf, none, none, os.path.join(tmp, "fx.fa")], capture_output=True, text=True, timeout=300)
    assert res.returncode == want["exit"] == 1
    assert _normalise(res.stdout) == want["stdout"]
    assert "No consensus found!" in res.stderr
    assert "Loaded 398/24127 edges from EColi_CR_overlaps.paf" in _normalise(res.stdout)


HIP_ONLY = ("load breakdown:", "edge-extension evals:", "Anchor pairs with paths:", "Host stages after the path build:")


@pytest.mark.gpu
@needs_binary
@pytest.mark.parametrize("name", ["tiny", "sparse", "quirks", "hard_tiny", "overflow"])
def test_same_report_and_bytes_as_telomeri_hip(name):
    """Under the sets' own arguments (Monte-Carlo included, per-walk seeds): the reference's main() and telomeri_hip's are two
    callers of the same classes -- same report from "Loading overlaps..." on, same scaffold bytes."""
    meta = G.load_json(name + ".json")
    cli = [a for a in meta["args"] if a != "--sqrt"]
    with tempfile.TemporaryDirectory() as tmp:
        files = _inputs(tmp, name)
        runs = []
        for exe, extra in ((ON_HIP, []), (HIP, ["--materialise"])):
            out = os.path.join(tmp, "out.fa")
            if os.path.exists(out):
                os.unlink(out)
            res = subprocess.run([exe] + cli + extra + files + [out], capture_output=True, text=True, timeout=300)
            lines = _normalise(res.stdout)
            lines = lines[lines.index("Loading overlaps..."):]
            lines = [ln for ln in lines if not ln.startswith(HIP_ONLY)]
            runs.append((res.returncode, lines, open(out, "rb").read() if os.path.exists(out) else None))
        assert runs[0][0] == runs[1][0]
        assert runs[0][1] == runs[1][1]
        assert runs[0][2] == runs[1][2]
        assert runs[0][2] or runs[0][0] != 0
