#!/usr/bin/env python3
"""The second half of BASELINE.json's metric -- scaffold wall-clock, PAF + FASTA in, scaffold FASTA out -- for the C++
host program `telomeri_hip` (reference argv, reference defaults: 2000 trials per anchor, 60 backtracks) on a synthetic
dataset of the given shape.  Run on the GPU box:  python profiles/probes/scaffold_bench.py [config]   (default ecoli)
Writes one JSON line: wall-clock of the whole program, its own per-stage times, output size."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from telomeri_amd import synth
from load_bench import write_paf


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "ecoli"
    binary = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_hip")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "telomeri_amd", "host")], check=True)
    t0 = time.time()
    ds = synth.make_config(config)
    out = {"config": config, "records": len(ds.rr) + len(ds.cr), "reads": ds.n_reads, "contigs": ds.n_contigs}
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), os.path.join(tmp, "cr.paf")
        reads, contigs, fasta = os.path.join(tmp, "reads.fa"), os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "scaffold.fa")
        write_paf(ds, ds.rr, rr)
        write_paf(ds, ds.cr, cr)
        synth.write_fasta(ds, reads, contigs)
        out["input_bytes"] = sum(os.path.getsize(p) for p in (rr, cr, reads, contigs))
        out["dataset_written_s"] = time.time() - t0
        t0 = time.time()
        res = subprocess.run([binary, rr, cr, reads, contigs, fasta], capture_output=True, text=True, env=dict(os.environ, TLM_HOST_TIMING="1"))
        out["host_timing"] = [ln for ln in res.stderr.splitlines() if ln.startswith("[host timing]")]
        out["wall_s"] = time.time() - t0
        out["exit"] = res.returncode
        done = [float(x) for x in re.findall(r"Done \(([0-9.e+-]+)s\)", res.stdout)]
        if len(done) >= 2:
            out["load_s"], out["paths_s"] = done[0], done[1]
        m = re.search(r"load breakdown: tokenise\+intern ([0-9.e+-]+) ms \((\w+)\), device graph build ([0-9.e+-]+) ms, "
                      r"download\+materialise ([0-9.e+-]+) ms", res.stdout)
        if m:
            out["load_breakdown_ms"] = {"tokenise": float(m.group(1)), "device_build": float(m.group(3)),
                                        "download_materialise": float(m.group(4))}
        m = re.search(r"edge-extension evals: (\d+) in ([0-9.e+-]+) ms of walk kernels", res.stdout)
        if m:
            out["evals"], out["walk_kernel_ms"] = int(m.group(1)), float(m.group(2))
        m = re.search(r"Host stages after the path build: ([0-9.e+-]+)s", res.stdout)
        if m:
            out["post_stages_s"] = float(m.group(1))
        if res.returncode == 0:
            out["scaffold_bytes"] = os.path.getsize(fasta)
            chain = [ln for ln in res.stdout.splitlines() if ln.startswith("ctg") and "-" in ln]
            out["scaffold_nodes"] = chain[-1].count("-") + 1 if chain else 0
        else:
            out["stderr_tail"] = res.stderr[-300:]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
