#!/usr/bin/env python3
"""SURVEY.md §8f row 2 measurement: wall-clock of PAF ingestion on E. coli-shaped text (4.9 M records, ~0.5 GB).
  * C++ host tokeniser alone, 1 thread and all threads (`telomeri_hip --parse-only`)
    (the CPU comparison -- the oracle's sequential loader -- is a test-tier matter: tests/test_host_ingest.py; nothing
    outside tests/, smoke() and bench.py's cpu_baseline leg touches oracle/)
  * the full host load (tokenise + device graph build + download) through `telomeri_hip`
Writes one JSON line.  Run on the GPU box: python profiles/probes/load_bench.py [config]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import pandas as pd

from telomeri_amd import synth


def write_paf(ds, rec, path):
    c = {k: getattr(rec, k).cpu().numpy() for k in ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end",
                                                     "matches", "strand")}
    def names(ids):
        out = np.char.add("read", np.char.zfill((ids + 1).astype(str), 7))
        ctg = ids >= ds.n_reads
        out[ctg] = np.char.add("ctg", (ids[ctg] - ds.n_reads + 1).astype(str))
        return out
    ovl = c["q_end"] - c["q_start"]
    df = pd.DataFrame({"q": names(c["q_id"]), "ql": c["q_len"], "qs": c["q_start"], "qe": c["q_end"],
                       "s": np.where(c["strand"] != 0, "-", "+"), "t": names(c["t_id"]), "tl": c["t_len"], "ts": c["t_start"],
                       "te": c["t_end"], "m": c["matches"], "b": ovl, "mq": 60, "x1": "tp:A:S", "x2": "cm:i:1", "x3": "s1:i:1",
                       "x4": "dv:f:0.0100"})
    df.to_csv(path, sep="\t", header=False, index=False)


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "ecoli"
    binary = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_hip")
    ds = synth.make_config(config)
    out = {"config": config, "records": len(ds.rr) + len(ds.cr)}
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), os.path.join(tmp, "cr.paf")
        write_paf(ds, ds.rr, rr)
        write_paf(ds, ds.cr, cr)
        out["bytes"] = os.path.getsize(rr) + os.path.getsize(cr)
        for label, env in (("parse_1thread", {"TLM_PARSE_THREADS": "1"}), ("parse_16threads", {"TLM_PARSE_THREADS": "16"}),
                           ("parse_allcores", {})):
            e = dict(os.environ); e.update(env)
            res = subprocess.run([binary, "--parse-only", rr, cr], capture_output=True, text=True, env=e, check=True)
            ms = sum(float(m) for m in re.findall(r" in ([0-9.e+-]+) ms", res.stdout))
            out[label + "_ms"] = ms
            out[label + "_records_per_s"] = out["records"] / (ms * 1e-3)
            out[label + "_GBps"] = out["bytes"] / (ms * 1e-3) / 1e9
        if os.path.exists("/dev/kfd"):
            t0 = time.time()
            res = subprocess.run([binary, "--rb-att", "1", rr, cr, "r", "c", "o"], capture_output=True, text=True, check=True)
            out["host_cli_wall_s"] = time.time() - t0
            m = re.search(r"load breakdown: tokenise\+intern ([0-9.e+-]+) ms \((\w+)\), device graph build ([0-9.e+-]+) ms, "
                          r"download\+materialise ([0-9.e+-]+) ms", res.stdout)
            out["host_load"] = {"tokenise_ms": float(m.group(1)), "mode": m.group(2), "device_build_ms": float(m.group(3)),
                                "download_materialise_ms": float(m.group(4))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
