"""Reading the committed fixtures of tests/golden/ (written by tests/golden/make_golden.py)."""
import gzip
import json
import os
import shutil
import tempfile

import numpy as np

from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gunzip_to(tmpdir: str, name: str) -> str:
    dst = os.path.join(tmpdir, name[:-3])
    with gzip.open(os.path.join(GOLDEN, name), "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    return dst


def load_json(name: str):
    return json.load(open(os.path.join(GOLDEN, name)))


def args_to_params(args):
    """ref_dump / reference CLI flags -> (oracle FilterParams, WalkParams) (reference Main.cpp:56-69,107-136)."""
    mode, min_oll, min_olp, max_ohl, max_ohp = "avg", 0, 0.0, po.ULONG_MAX, 1.0
    rb, bt, thr = 2000, 60, 4_700_000
    it = iter(args)
    for a in it:
        if a.startswith("--filter-"):
            mode = a[len("--filter-"):]
        elif a == "--min-oll":
            min_oll = int(next(it))
        elif a == "--min-olp":
            min_olp = float(next(it))
        elif a == "--max-ohl":
            max_ohl = int(next(it))
        elif a == "--max-ohp":
            max_ohp = float(next(it))
        elif a == "--rb-att":
            rb = int(next(it))
        elif a == "--bt-att":
            bt = int(next(it))
        elif a == "--len-thr":
            thr = int(next(it))
    return po.FilterParams.default(mode, min_oll, min_olp, max_ohl, max_ohp), po.WalkParams.default(rb, bt, thr)


def csr_as_dump_rows(c) -> np.ndarray:
    """The oracle's / device's CSR in the row format of ref_dump's `a` records."""
    deg = np.diff(c.row_ptr).astype(np.int64)
    owner = np.repeat(np.arange(c.n_nodes), deg)
    slot = np.arange(c.n_adj) - c.row_ptr[owner].astype(np.int64)
    return np.stack([owner, slot, c.nbr, c.q_start, c.q_end, c.t_start, c.t_end, c.os.view(np.uint32),
                     c.si.view(np.uint32), c.es.view(np.uint32), c.strand], 1).astype(np.int64)


def parse_paf_columns(path: str):
    """A 16-column PAF as integer columns + a first-seen name table (what a host tokeniser hands the library)."""
    names, ids = [], {}
    cols = {k: [] for k in ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches", "strand")}
    with open(path) as fh:
        for ln in fh:
            f = ln.split()
            if len(f) < 12:
                continue
            for key, nm in (("q_id", f[0]), ("t_id", f[5])):
                if nm not in ids:
                    ids[nm] = len(names)
                    names.append(nm)
                cols[key].append(ids[nm])
            for key, v in (("q_len", f[1]), ("q_start", f[2]), ("q_end", f[3]), ("t_len", f[6]), ("t_start", f[7]),
                           ("t_end", f[8]), ("matches", f[9])):
                cols[key].append(int(v))
            cols["strand"].append(1 if f[4] == "-" else 0)
    return names, {k: np.array(v, dtype=np.int32) for k, v in cols.items()}
