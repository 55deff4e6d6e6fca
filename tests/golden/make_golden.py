#!/usr/bin/env python3
"""Regenerates tests/golden/ from the UNMODIFIED reference (oracle/_ref, built by `make -C oracle ref`
from the sources under /root/reference).  Run in the build container only; the fixtures it writes
are data (inputs + the reference's outputs), committed so that they travel to the GPU box where the
reference does not exist.

    python tests/golden/make_golden.py

Fixtures
  ecoli_cr.paf.gz            the reference's own bundled data file test/res/EColi_CR_overlaps.paf
  ecoli_cr_known.json        what the reference loads from it under 5 filter settings
  <set>_rr.paf.gz, <set>_cr.paf.gz, <set>.dump.gz, <set>.json
                             synthetic inputs (telomeri_amd.synth / tests.helpers), the reference's
                             graph + the four builders' path lists (ref_dump format), and the
                             arguments used
  mid.json                   the reference PROGRAM (--rb-att 0) on a 16 000-read dataset that synth regenerates from its
                             seed: report, exit code, scaffold md5 (no input fixture stored)
  mt19937.json               std::mt19937(5489) raw outputs and uniform_real_distribution<double>
                             values from libstdc++ (a 10-line C++ program compiled here)
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from oracle import pyoracle as po  # noqa: E402
from telomeri_amd import synth  # noqa: E402
import helpers as H  # noqa: E402

REF_FIXTURE = "/root/reference/test/res/EColi_CR_overlaps.paf"


def gz_write(src: str, dst: str) -> None:
    with open(src, "rb") as fi, gzip.GzipFile(dst, "wb", mtime=0) as fo:
        shutil.copyfileobj(fi, fo)


def graph_digest(d: po.RefDump) -> str:
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(d.node_anchor).tobytes())
    h.update(np.ascontiguousarray(d.node_len).tobytes())
    h.update("\n".join(d.node_name).encode())
    h.update(np.ascontiguousarray(d.adj).tobytes())
    return h.hexdigest()


def normalise_report(stdout: str) -> list:
    """The reference program's stdout as comparable lines: lap times and directory names are run-specific."""
    out = []
    for ln in stdout.splitlines():
        if ln.startswith("Done (") or ln.startswith("Total time:"):
            continue
        for key in ("Loaded ", "Writing result to: "):
            if ln.startswith(key) and "/" in ln:
                head, _, path = ln.rpartition(" ")
                ln = head + " " + os.path.basename(path)
        out.append(ln)
    return out


def run_program(cli, files, fasta_out) -> dict:
    """The unmodified reference program (oracle/_ref/Telomeri) as a whole: exit code, report, scaffold."""
    res = subprocess.run([po.REF_MAIN] + cli + files + [fasta_out], capture_output=True, text=True)
    rec = {"args": cli, "exit": res.returncode, "stdout": normalise_report(res.stdout),
           "stderr": normalise_report(res.stderr)}
    if res.returncode == 0:
        body = open(fasta_out, "rb").read()
        rec.update({"md5": hashlib.md5(body).hexdigest(), "bytes": len(body)})
    return rec


MID = dict(seed=61, genome_len=9_000_000, n_contigs=30, contig_len=60_000, n_reads=16_000, read_len_lo=4_000, read_len_hi=14_000)


def make_mid(tmp: str) -> None:
    """mid.json: the whole reference PROGRAM with the Monte-Carlo builders switched off (--rb-att 0) on a dataset sixty times
    the size of the other sets -- 16 000 reads, 30 contigs, 456 000 overlap records; the reference needs ~25 s for it (its
    O(N) nodeIndex scan).  No input fixture is stored: telomeri_amd.synth regenerates the dataset from its seed
    (tests/test_reference_main_on_hip.py writes the same text).  What is frozen: the report, the exit code, the scaffold's
    size and md5."""
    ds = synth.make_dataset("mid", **MID)
    rr, cr = os.path.join(tmp, "mid_rr.paf"), os.path.join(tmp, "mid_cr.paf")
    reads, contigs = os.path.join(tmp, "mid_reads.fa"), os.path.join(tmp, "mid_contigs.fa")
    synth.write_paf(ds, ds.rr, rr)
    synth.write_paf(ds, ds.cr, cr)
    synth.write_fasta(ds, reads, contigs)
    det_only = run_program(["--rb-att", "0"], [rr, cr, reads, contigs], os.path.join(tmp, "det.fa"))
    inputs = {os.path.basename(f): hashlib.md5(open(f, "rb").read()).hexdigest() for f in (rr, cr, reads, contigs)}
    json.dump({"dataset": MID, "records": [len(ds.rr), len(ds.cr)], "inputs_md5": inputs, "program_det_only": det_only},
              open(os.path.join(HERE, "mid.json"), "w"), indent=1, sort_keys=True)
    print("mid", det_only["exit"], det_only.get("bytes"), len(det_only["stdout"]), "report lines")


def main() -> None:
    assert po.have_ref(), "build the reference first: make -C oracle ref"
    if "--mid" in sys.argv:         # only the mid-size program record
        tmp = tempfile.mkdtemp(prefix="tlm_golden_")
        make_mid(tmp)
        shutil.rmtree(tmp)
        return
    tmp = tempfile.mkdtemp(prefix="tlm_golden_")
    empty = os.path.join(tmp, "empty.paf")
    open(empty, "w").close()

    # --- 1. the bundled fixture and its known-answer table
    gz_write(REF_FIXTURE, os.path.join(HERE, "ecoli_cr.paf.gz"))
    known = {}
    settings = {
        "avg": ["--filter-avg"], "sum": ["--filter-sum"], "min": ["--filter-min"], "max": ["--filter-max"],
        "strict": ["--min-oll", "5000", "--max-ohp", "0.1", "--min-olp", "0.05", "--max-ohl", "3000"],
    }
    for key, args in settings.items():
        d = po.run_ref_dump(empty, REF_FIXTURE, os.path.join(tmp, "fx.dump"), args + ["--rb-att", "20"])
        os_bits = d.adj[:, 7].astype(np.uint32).view(np.float32)
        es_bits = d.adj[:, 9].astype(np.uint32).view(np.float32)
        known[key] = {
            "args": args, "kept": int(d.adj.shape[0] // 2), "nodes": len(d.node_name),
            "anchors": int(d.node_anchor.sum()), "max_con": int(d.node_deg.max()),
            "min_os": float(os_bits.min()), "max_os": float(os_bits.max()),
            "min_es": float(es_bits.min()), "max_es": float(es_bits.max()),
            "graph_sha256": graph_digest(d),
            "paths": {str(b): len(p) for b, p in d.builders.items()},
        }
    # the whole program on the fixture (empty rr file: no paths, "No consensus found!", exit 1 -- SURVEY.md §4)
    nofile = os.path.join(tmp, "none.fa")
    open(nofile, "w").close()
    known["program"] = run_program([], [empty, REF_FIXTURE, nofile, nofile], os.path.join(tmp, "fx.fa"))
    json.dump(known, open(os.path.join(HERE, "ecoli_cr_known.json"), "w"), indent=1, sort_keys=True)

    # --- 2. synthetic sets with full dumps
    sets = {
        "tiny": (synth.make_config("tiny"), ["--rb-att", "60"]),
        "sparse": (synth.make_config("sparse"), ["--rb-att", "80", "--bt-att", "3"]),
        "quirks": (H.quirks_dataset(), ["--filter-min", "--rb-att", "50", "--bt-att", "5", "--sqrt"]),
        # the hostile generator at a size the reference runs in a second (repeat families, chimeric reads, skewed contigs)
        "hard_tiny": (synth.make_config("hard_tiny"), ["--rb-att", "40"]),
        # 40 Mb paths: the int accumulator of PathGroup.cpp:66-68 overflows (f3's quirks that only bite at size)
        "overflow": (H.overflow_dataset(), ["--len-thr", "100000000", "--rb-att", "200"]),
    }
    for name, (ds, args) in sets.items():
        rr, cr = os.path.join(tmp, name + "_rr.paf"), os.path.join(tmp, name + "_cr.paf")
        synth.write_paf(ds, ds.rr, rr)
        synth.write_paf(ds, ds.cr, cr)
        dump = os.path.join(tmp, name + ".dump")
        d = po.run_ref_dump(rr, cr, dump, args)
        # the whole reference program on the same inputs + FASTA (telomeri_amd.synth.write_fasta, seed 0): the scaffold
        reads, contigs, fasta = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "out.fa")
        synth.write_fasta(ds, reads, contigs)
        cli = [a for a in args if a != "--sqrt"]
        res = subprocess.run([po.REF_MAIN] + cli + [rr, cr, reads, contigs, fasta], capture_output=True, text=True)
        scaffold = {"exit": res.returncode}
        if res.returncode == 0:
            body = open(fasta, "rb").read()
            chain = [ln for ln in res.stdout.splitlines() if ln.startswith(("ctg", "read")) and "-" in ln]
            scaffold.update({"md5": hashlib.md5(body).hexdigest(), "bytes": len(body), "chain_md5": hashlib.md5(chain[-1].encode()).hexdigest(),
                             "consensus_lines": [ln for ln in res.stdout.splitlines() if ln.startswith(("Consensus length", "Final consensus"))]})
        # the same program with the Monte-Carlo builders switched off (--rb-att 0: no trials): everything it prints and
        # writes then depends on the deterministic builders alone, which the device reproduces bit for bit -- the report
        # and the scaffold of tests/test_reference_main_on_hip.py (the reference's own main() on the HIP path)
        det_cli = list(cli)
        det_cli[det_cli.index("--rb-att") + 1] = "0"
        det_only = run_program(det_cli, [rr, cr, reads, contigs], os.path.join(tmp, "det.fa"))
        # grouping with parameters that force several windows / groups (independent of the walk parameters)
        groups = {}
        for gp in (("2000", "500", "0.9"), ("500", "200", "0.5"), ("100", "50", "0.99")):
            gd = os.path.join(tmp, name + "_g.dump")
            subprocess.run([po.REF_DUMP] + args + ["--no-graph", "--groups", *gp, rr, cr, gd], check=True, stdout=subprocess.DEVNULL)
            txt = open(gd).read()
            groups[" ".join(gp)] = txt[txt.index("\nG ") + 1:]
        gz_write(rr, os.path.join(HERE, name + "_rr.paf.gz"))
        gz_write(cr, os.path.join(HERE, name + "_cr.paf.gz"))
        gz_write(dump, os.path.join(HERE, name + ".dump.gz"))
        json.dump({"args": args, "paths": {str(b): len(p) for b, p in d.builders.items()},
                   "nodes": len(d.node_name), "adj": int(d.adj.shape[0]), "scaffold": scaffold, "groups": groups,
                   "program_det_only": det_only},
                  open(os.path.join(HERE, name + ".json"), "w"), indent=1, sort_keys=True)
        print(name, len(d.node_name), d.adj.shape[0], {b: len(p) for b, p in d.builders.items()})

    make_mid(tmp)

    # --- 3. libstdc++ RNG vectors
    src = os.path.join(tmp, "rng.cpp")
    open(src, "w").write(r'''
#include <random>
#include <cstdio>
#include <cstring>
int main() {
    std::mt19937 g;                       // default seed 5489 (PathManager.cpp:13)
    std::printf("{\"raw\": [");
    for (int i = 0; i < 16; i++) std::printf("%s%u", i ? ", " : "", (unsigned) g());
    std::mt19937 g2;
    g2.discard(9999);
    std::printf("], \"raw_10000th\": %u, \"canonical_bits\": [", (unsigned) g2());
    std::mt19937 g3;
    std::uniform_real_distribution<> dis(0., 1.);   // PathManager.cpp:14
    for (int i = 0; i < 700; i++) {        // crosses a twist (624 words = 312 doubles)
        double d = dis(g3); unsigned long long u; std::memcpy(&u, &d, 8);
        std::printf("%s%llu", i ? ", " : "", u);
    }
    std::mt19937 g4(123456789u);
    std::printf("], \"seed\": 123456789, \"seed_canonical_bits\": [");
    for (int i = 0; i < 8; i++) {
        double d = dis(g4); unsigned long long u; std::memcpy(&u, &d, 8);
        std::printf("%s%llu", i ? ", " : "", u);
    }
    std::printf("]}\n");
}
''')
    exe = os.path.join(tmp, "rng")
    subprocess.run(["g++", "-O1", "-o", exe, src], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    json.dump(json.loads(out), open(os.path.join(HERE, "mt19937.json"), "w"))
    shutil.rmtree(tmp)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
