#!/usr/bin/env python3
"""profiles/probes/tok_probe.py <config> <lib.so> [<lib.so> ...] -- the device tokeniser alone on the config's read-read text:
the text is written once, every library (ablation builds of csrc/paf_parse.hip) tokenises it in a process of its own and
reports tlm_paf_timing.  Child mode: tok_probe.py --child <path>."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(path):
    import ctypes as C
    from telomeri_amd import capi
    ctx = capi.Context(0)
    out = []
    for rep in range(3):
        paf = capi.PafTokens(ctx)
        t0 = time.time()
        n, clean = paf.add_file(path, anchors=False)
        wall = time.time() - t0
        up, pa, fi = C.c_double(), C.c_double(), C.c_double()
        capi.lib().tlm_paf_timing(paf._h, C.byref(up), C.byref(pa), C.byref(fi))
        out.append({"records": int(n), "clean": bool(clean), "upload_ms": up.value, "tokenise_ms": pa.value, "wall_s": wall})
        del paf
    print(json.dumps(out))


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    from telomeri_amd import scaffold_bench as sb, synth
    config, libs = sys.argv[2 - 1], sys.argv[2:]
    ds = synth.make_config(config, device="cuda")
    tmp = sb.scratch_dir(len(ds.rr) * 110)
    rr = os.path.join(tmp, "rr_overlaps.paf")
    try:
        nb = sb.write_paf(ds, ds.rr, rr)
        del ds
        import torch
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        for lib in libs:
            env = dict(os.environ, TLM_LIB=os.path.abspath(lib))
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", rr], env=env, capture_output=True, text=True,
                                 timeout=300)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("[")]
            print(json.dumps({"lib": lib, "bytes": nb, "runs": json.loads(line[-1]) if line else res.stderr[-500:]}), flush=True)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
