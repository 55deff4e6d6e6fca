"""TEST / MEASUREMENT INFRASTRUCTURE (build container only: needs oracle/_ref, i.e. /root/reference).

The sanity ratio SURVEY.md §8(d) asks for: the UNMODIFIED reference program (oracle/_ref/Telomeri, one thread by design)
timed beside the oracle -- the CPU restatement that bench.py's cpu_baseline leg times on the GPU box -- on the same small
inputs, same walks (the oracle in SEED_SHARED mode draws from one default-seeded std::mt19937 per builder call exactly as
the reference does, so both make the same steps; tests/test_oracle_vs_reference.py checks the paths).

    python oracle/ref_vs_oracle_timing.py > profiles/r03_reference_vs_oracle.json

Reference time of the path build = its own "Done (x s)" lap after the four builders (Main.cpp:230-248); evals = the
oracle's count for those same walks (the reference counts nothing).  Load is timed the same way (lap after graph.load)."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po          # noqa: E402
from telomeri_amd import synth             # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "Telomeri")


def one(config: str, reps: int = 3) -> dict:
    ds = synth.make_config(config)
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), os.path.join(tmp, "cr.paf")
        reads, contigs, out = os.path.join(tmp, "reads.fasta"), os.path.join(tmp, "contigs.fasta"), os.path.join(tmp, "out.fasta")
        synth.write_paf(ds, ds.rr, rr)
        synth.write_paf(ds, ds.cr, cr)
        synth.write_fasta(ds, reads, contigs)
        ref_load, ref_build, ref_total = [], [], []
        for _ in range(reps):
            t0 = time.time()
            res = subprocess.run([REF, rr, cr, reads, contigs, out], capture_output=True, text=True)
            ref_total.append(time.time() - t0)
            laps = [float(x) for x in re.findall(r"Done \(([0-9.e+-]+)s\)", res.stdout)]
            ref_load.append(laps[0])
            ref_build.append(laps[1])
        fp = po.FilterParams(2, 0, 0.0, (1 << 64) - 1, 1.0)        # Main.cpp:56-62 defaults (AVG)
        wp = po.WalkParams(2000, 60, 4_700_000)                     # Main.cpp:63-66
        ora_load, ora_build, evals = [], [], 0
        for _ in range(reps):
            t0 = time.time()
            g = po.Graph()
            assert g.load_paf(rr, False, fp) and g.load_paf(cr, True, fp)
            ora_load.append(time.time() - t0)
            t0 = time.time()
            evals = 0
            for m in (po.METRIC_ES, po.METRIC_OS):
                evals += g.walk_mc(m, wp, po.SEED_SHARED, threads=1)[1].evals
            for m in (po.METRIC_ES, po.METRIC_OS):
                evals += g.walk_det(m, threads=1)[1].evals
            ora_build.append(time.time() - t0)
    rb, ob = min(ref_build), min(ora_build)
    return {"config": config, "records": len(ds.rr) + len(ds.cr), "evals_four_builders": int(evals),
            "reference_path_build_s": rb, "oracle_path_build_s_1thread": ob,
            "reference_evals_per_s": evals / rb, "oracle_evals_per_s_1thread": evals / ob,
            "oracle_over_reference_path_build": rb / ob,
            "reference_load_s": min(ref_load), "oracle_load_s": min(ora_load), "reference_program_wall_s": min(ref_total)}


if __name__ == "__main__":
    model = ""
    for ln in open("/proc/cpuinfo"):
        if ln.startswith("model name"):
            model = ln.split(":", 1)[1].strip()
            break
    print(json.dumps({"where": "build container", "cpu_model": model, "threads": 1,
                      "reference": "oracle/_ref/Telomeri = g++ -O2 on the unmodified /root/reference/src/*.cpp",
                      "runs": [one(c) for c in (sys.argv[1:] or ["tiny", "toy"])]}, indent=1))
