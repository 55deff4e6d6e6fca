"""ctypes binding of oracle/libtelomeri_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/telomeri_oracle.h).  It also holds the parser for the text dumps written by
oracle/_ref/ref_dump (the unmodified reference behind our dump driver).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtelomeri_oracle.so")
REF_DUMP = os.path.join(HERE, "_ref", "ref_dump")
REF_MAIN = os.path.join(HERE, "_ref", "Telomeri")

MODE = {"min": 0, "max": 1, "avg": 2, "sum": 3}
METRIC_ES, METRIC_OS, METRIC_ES_SQRT, METRIC_OS_SQRT = 0, 1, 2, 3
SEED_SHARED, SEED_PER_WALK = 0, 1
ULONG_MAX = (1 << 64) - 1


class FilterParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_overlap_length", C.c_uint64), ("min_overlap_percentage", C.c_float),
                ("max_overhang_length", C.c_uint64), ("max_overhang_percentage", C.c_float)]

    @staticmethod
    def default(mode: str = "avg", min_oll: int = 0, min_olp: float = 0.0, max_ohl: int = ULONG_MAX,
                max_ohp: float = 1.0) -> "FilterParams":   # reference Main.cpp:56-62
        return FilterParams(MODE[mode], min_oll, min_olp, max_ohl, max_ohp)


class WalkParams(C.Structure):
    _fields_ = [("rebuild_attempts", C.c_int32), ("backtrack_attempts", C.c_int32), ("len_threshold", C.c_int64)]

    @staticmethod
    def default(rb_att: int = 2000, bt_att: int = 60, len_thr: int = 4_700_000) -> "WalkParams":  # Main.cpp:63-66
        return WalkParams(rb_att, bt_att, len_thr)


class Counters(C.Structure):
    _fields_ = [("units", C.c_uint64), ("steps", C.c_uint64), ("evals", C.c_uint64), ("draws", C.c_uint64),
                ("accepted", C.c_uint64)]


class _Csr(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_adj", C.c_uint64), ("row_ptr", C.POINTER(C.c_uint64)),
                ("nbr", C.POINTER(C.c_uint32)), ("q_start", C.POINTER(C.c_uint32)), ("q_end", C.POINTER(C.c_uint32)),
                ("t_start", C.POINTER(C.c_uint32)), ("t_end", C.POINTER(C.c_uint32)), ("os", C.POINTER(C.c_float)),
                ("si", C.POINTER(C.c_float)), ("es", C.POINTER(C.c_float)), ("strand", C.POINTER(C.c_uint8)),
                ("node_len", C.POINTER(C.c_uint32)), ("node_anchor", C.POINTER(C.c_uint8)), ("n_edges", C.c_uint64)]


class _Paths(C.Structure):
    _fields_ = [("n_paths", C.c_uint64), ("unit", C.POINTER(C.c_uint64)), ("length", C.POINTER(C.c_int64)),
                ("off", C.POINTER(C.c_uint64)), ("nodes", C.POINTER(C.c_uint32)), ("slots", C.POINTER(C.c_uint32))]


UNIT_STAT_DTYPE = np.dtype([("status", np.uint8), ("n_nodes", np.uint32), ("steps", np.uint32), ("evals", np.uint64),
                            ("draws", np.uint32), ("length", np.int64)], align=True)


@dataclasses.dataclass
class Csr:
    n_nodes: int
    n_adj: int
    n_edges: int
    row_ptr: np.ndarray
    nbr: np.ndarray
    q_start: np.ndarray
    q_end: np.ndarray
    t_start: np.ndarray
    t_end: np.ndarray
    os: np.ndarray
    si: np.ndarray
    es: np.ndarray
    strand: np.ndarray
    node_len: np.ndarray
    node_anchor: np.ndarray


@dataclasses.dataclass
class PathSet:
    """Packed Path set: path i = nodes[off[i]:off[i+1]], slots likewise (last slot 0xFFFFFFFF)."""
    unit: np.ndarray
    length: np.ndarray
    off: np.ndarray
    nodes: np.ndarray
    slots: np.ndarray

    def __len__(self) -> int:
        return int(self.unit.shape[0])

    def path(self, i: int) -> Tuple[int, np.ndarray, np.ndarray]:
        a, b = int(self.off[i]), int(self.off[i + 1])
        return int(self.length[i]), self.nodes[a:b], self.slots[a:b - 1]


_lib = None


def build() -> None:
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(HERE, "telomeri_oracle.c")
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        build()
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.tlo_graph_new.restype = C.c_void_p
    L.tlo_graph_free.argtypes = [C.c_void_p]
    L.tlo_filter_score.argtypes = [P(FilterParams), C.c_size_t] + [C.c_void_p] * 12
    L.tlo_graph_load_paf.argtypes = [C.c_void_p, C.c_char_p, C.c_int, P(FilterParams), C.c_uint32, P(C.c_uint64)]
    L.tlo_graph_add_records.argtypes = [C.c_void_p, C.c_int, P(FilterParams), C.c_size_t] + [C.c_void_p] * 11
    L.tlo_graph_from_csr.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8
    L.tlo_graph_csr.argtypes = [C.c_void_p, P(_Csr)]
    L.tlo_graph_node_name.argtypes = [C.c_void_p, C.c_uint32]
    L.tlo_graph_node_name.restype = C.c_char_p
    L.tlo_graph_node_ext_id.argtypes = [C.c_void_p, C.c_uint32]
    L.tlo_graph_node_ext_id.restype = C.c_int64
    L.tlo_mc_units.argtypes = [C.c_void_p, P(WalkParams)]
    L.tlo_mc_units.restype = C.c_uint64
    L.tlo_det_units.argtypes = [C.c_void_p]
    L.tlo_det_units.restype = C.c_uint64
    L.tlo_walk_seed.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    L.tlo_walk_seed.restype = C.c_uint32
    L.tlo_walk_mc.argtypes = [C.c_void_p, C.c_int, P(WalkParams), C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                              P(Counters), C.c_void_p]
    L.tlo_walk_mc.restype = P(_Paths)
    L.tlo_walk_det.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, P(Counters), C.c_void_p]
    L.tlo_walk_det.restype = P(_Paths)
    L.tlo_paths_free.argtypes = [P(_Paths)]
    L.tlo_mt19937_raw.argtypes = [C.c_uint32, C.c_size_t, C.c_void_p]
    L.tlo_mt19937_canonical.argtypes = [C.c_uint32, C.c_size_t, C.c_void_p]
    _lib = L
    return L


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def filter_score(fp: FilterParams, q_len, q_start, q_end, t_len, t_start, t_end, matches, same_name=None):
    cols = [_i32(x) for x in (q_len, q_start, q_end, t_len, t_start, t_end, matches)]
    n = cols[0].shape[0]
    same = None if same_name is None else np.ascontiguousarray(same_name, dtype=np.uint8)
    keep = np.zeros(n, np.uint8)
    si = np.zeros(n, np.float32)
    os_ = np.zeros(n, np.float32)
    es = np.zeros(n, np.float32)
    lib().tlo_filter_score(C.byref(fp), n, *[_ptr(c) for c in cols], _ptr(same), _ptr(keep), _ptr(si), _ptr(os_),
                           _ptr(es))
    return keep, si, os_, es


def _take_paths(pp) -> PathSet:
    L = lib()
    p = pp.contents
    n = int(p.n_paths)
    unit = np.ctypeslib.as_array(p.unit, shape=(max(n, 1),))[:n].copy()
    length = np.ctypeslib.as_array(p.length, shape=(max(n, 1),))[:n].copy()
    off = np.ctypeslib.as_array(p.off, shape=(n + 1,)).copy()
    tot = int(off[n])
    nodes = np.ctypeslib.as_array(p.nodes, shape=(max(tot, 1),))[:tot].copy()
    slots = np.ctypeslib.as_array(p.slots, shape=(max(tot, 1),))[:tot].copy()
    L.tlo_paths_free(pp)
    return PathSet(unit, length, off, nodes, slots)


class Graph:
    def __init__(self) -> None:
        self._g = lib().tlo_graph_new()
        self.records: List[int] = []

    def __del__(self) -> None:
        try:
            if getattr(self, "_g", None):
                lib().tlo_graph_free(self._g)
                self._g = None
        except Exception:      # interpreter shutdown
            pass

    def load_paf(self, path: str, anchors: bool, fp: FilterParams, test_load_num: int = 0) -> bool:
        n = C.c_uint64(0)
        rc = lib().tlo_graph_load_paf(self._g, path.encode(), int(anchors), C.byref(fp), test_load_num, C.byref(n))
        self.records.append(int(n.value))
        return rc == 0

    def add_records(self, anchors: bool, fp: FilterParams, name_is_ctg, q_id, t_id, q_len, q_start, q_end, t_len,
                    t_start, t_end, matches, strand) -> None:
        cols = [_i32(x) for x in (q_id, t_id)]
        ctg = np.ascontiguousarray(name_is_ctg, dtype=np.uint8)
        rest = [_i32(x) for x in (q_len, q_start, q_end, t_len, t_start, t_end, matches, strand)]
        rc = lib().tlo_graph_add_records(self._g, int(anchors), C.byref(fp), cols[0].shape[0], _ptr(cols[0]),
                                         _ptr(cols[1]), _ptr(ctg), *[_ptr(c) for c in rest])
        if rc != 0:
            raise ValueError("tlo_graph_add_records failed")

    @staticmethod
    def from_csr(row_ptr, nbr, q_start, t_start, os_, es, node_len, node_anchor) -> "Graph":
        g = Graph()
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        arrs = [np.ascontiguousarray(nbr, np.uint32), np.ascontiguousarray(q_start, np.uint32),
                np.ascontiguousarray(t_start, np.uint32), np.ascontiguousarray(os_, np.float32),
                np.ascontiguousarray(es, np.float32), np.ascontiguousarray(node_len, np.uint32),
                np.ascontiguousarray(node_anchor, np.uint8)]
        rc = lib().tlo_graph_from_csr(g._g, row_ptr.shape[0] - 1, arrs[0].shape[0], _ptr(row_ptr),
                                      *[_ptr(a) for a in arrs])
        if rc != 0:
            raise ValueError("tlo_graph_from_csr failed")
        return g

    def csr(self) -> Csr:
        c = _Csr()
        if lib().tlo_graph_csr(self._g, C.byref(c)) != 0:
            raise RuntimeError("tlo_graph_csr failed")
        n, a = int(c.n_nodes), int(c.n_adj)

        def arr(p, k):
            if k == 0:
                return np.zeros(0, dtype=np.ctypeslib.as_array(p, shape=(1,)).dtype)
            return np.ctypeslib.as_array(p, shape=(k,)).copy()
        return Csr(n, a, int(c.n_edges), arr(c.row_ptr, n + 1), arr(c.nbr, a), arr(c.q_start, a), arr(c.q_end, a),
                   arr(c.t_start, a), arr(c.t_end, a), arr(c.os, a), arr(c.si, a), arr(c.es, a), arr(c.strand, a),
                   arr(c.node_len, n), arr(c.node_anchor, n))

    def node_name(self, i: int) -> str:
        return lib().tlo_graph_node_name(self._g, i).decode()

    def node_ext_id(self, i: int) -> int:
        return int(lib().tlo_graph_node_ext_id(self._g, i))

    def mc_units(self, wp: WalkParams) -> int:
        return int(lib().tlo_mc_units(self._g, C.byref(wp)))

    def det_units(self) -> int:
        return int(lib().tlo_det_units(self._g))

    def walk_mc(self, metric: int, wp: WalkParams, seed_mode: int = SEED_PER_WALK, unit_begin: int = 0,
                unit_end: Optional[int] = None, threads: int = 1):
        total = self.mc_units(wp)
        unit_end = total if unit_end is None else min(unit_end, total)
        unit_begin = min(unit_begin, unit_end)
        stats = np.zeros(max(unit_end - unit_begin, 1), UNIT_STAT_DTYPE)
        ctr = Counters()
        pp = lib().tlo_walk_mc(self._g, metric, C.byref(wp), seed_mode, unit_begin, unit_end, threads, C.byref(ctr),
                               _ptr(stats))
        if not pp:
            raise RuntimeError("tlo_walk_mc failed")
        return _take_paths(pp), ctr, stats[:unit_end - unit_begin]

    def walk_det(self, metric: int, unit_begin: int = 0, unit_end: Optional[int] = None, threads: int = 1):
        total = self.det_units()
        unit_end = total if unit_end is None else min(unit_end, total)
        unit_begin = min(unit_begin, unit_end)
        stats = np.zeros(max(unit_end - unit_begin, 1), UNIT_STAT_DTYPE)
        ctr = Counters()
        pp = lib().tlo_walk_det(self._g, metric, unit_begin, unit_end, threads, C.byref(ctr), _ptr(stats))
        if not pp:
            raise RuntimeError("tlo_walk_det failed")
        return _take_paths(pp), ctr, stats[:unit_end - unit_begin]


def walk_seed(metric: int, anchor_node: int, r: int) -> int:
    return int(lib().tlo_walk_seed(metric, anchor_node, r))


def mt19937_raw(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint32)
    lib().tlo_mt19937_raw(seed, n, _ptr(out))
    return out


def mt19937_canonical(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.float64)
    lib().tlo_mt19937_canonical(seed, n, _ptr(out))
    return out


# ---------------------------------------------------------------------------------------------
# ref_dump text format (oracle/ref_dump.cpp)

def concat_paths(sets) -> PathSet:
    """paths_ after several builder calls on one PathManager (reference Main.cpp:237-243): plain concatenation.
    `unit` keeps the builder position in its top byte."""
    unit = np.concatenate([s.unit.astype(np.uint64) | (np.uint64(k) << np.uint64(56)) for k, s in enumerate(sets)])
    length = np.concatenate([s.length for s in sets])
    counts = np.concatenate([np.diff(s.off.astype(np.int64)) for s in sets])
    off = np.zeros(counts.shape[0] + 1, np.uint64)
    off[1:] = np.cumsum(counts)
    return PathSet(unit, length, off, np.concatenate([s.nodes for s in sets]), np.concatenate([s.slots for s in sets]))


def filter_unique(ps: PathSet) -> PathSet:
    """PathManager::filterUnique (reference PathManager.cpp:432-438): std::unique with Path::operator==
    (Path.cpp:46-49: same node sequence; edges and length are not compared) -- ADJACENT duplicates only."""
    keep = []
    last = None
    for i in range(len(ps)):
        a, b = int(ps.off[i]), int(ps.off[i + 1])
        seq = ps.nodes[a:b]
        if last is not None and seq.shape == last.shape and np.array_equal(seq, last):
            continue
        keep.append(i)
        last = seq
    keep = np.array(keep, dtype=np.int64)
    counts = np.diff(ps.off.astype(np.int64))[keep] if len(keep) else np.zeros(0, np.int64)
    off = np.zeros(len(keep) + 1, np.uint64)
    off[1:] = np.cumsum(counts)
    idx = np.concatenate([np.arange(int(ps.off[i]), int(ps.off[i + 1])) for i in keep]) if len(keep) else np.zeros(0, np.int64)
    return PathSet(ps.unit[keep], ps.length[keep], off, ps.nodes[idx], ps.slots[idx])


def paths_between_anchors(ps: PathSet):
    """PathManager::getPathsBetweenAnchors (reference PathManager.cpp:626-645): std::map keyed by (front Node*,
    back Node*) -- iteration order = (front index, back index) -- each bucket in paths_ order.
    -> list of (front, back, [path indices])."""
    buckets = {}
    for i in range(len(ps)):
        key = (int(ps.nodes[int(ps.off[i])]), int(ps.nodes[int(ps.off[i + 1]) - 1]))
        buckets.setdefault(key, []).append(i)
    return [(k[0], k[1], v) for k, v in sorted(buckets.items())]


@dataclasses.dataclass
class RefDump:
    node_anchor: np.ndarray
    node_len: np.ndarray
    node_deg: np.ndarray
    node_name: List[str]
    adj: Optional[np.ndarray]          # [A, 11] int64: owner slot nbr qs qe ts te os_bits si_bits es_bits strand
    builders: Dict[int, List[Tuple[int, np.ndarray, np.ndarray]]]   # id -> [(length, nodes, slots)]
    unique: Optional[List[Tuple[int, np.ndarray, np.ndarray]]] = None   # paths_ after the 4 builders + filterUnique
    pairs: Optional[List[Tuple[int, int, List[int]]]] = None            # getPathsBetweenAnchors over `unique`


def parse_ref_dump(path: str) -> RefDump:
    anchors: List[int] = []
    lens: List[int] = []
    degs: List[int] = []
    names: List[str] = []
    adj: List[List[int]] = []
    builders: Dict[int, List[Tuple[int, np.ndarray, np.ndarray]]] = {}
    unique = None
    pairs = None
    cur: Optional[List] = None
    with open(path) as fh:
        head = fh.readline().split()
        assert head == ["TLMDUMP", "1"], head
        for line in fh:
            tag = line[0]
            if tag == "n":
                f = line.split()
                anchors.append(int(f[2])); lens.append(int(f[3])); degs.append(int(f[4])); names.append(f[5])
            elif tag == "a":
                adj.append([int(x) for x in line.split()[1:]])
            elif tag == "B":
                f = line.split()
                cur = []
                builders[int(f[1])] = cur
            elif tag == "U":
                cur = []
                unique = cur
            elif tag == "K":
                pairs = []
            elif tag == "k":
                f = [int(x) for x in line.split()[1:]]
                pairs.append((f[0], f[1], f[3:3 + f[2]]))
            elif tag == "p":
                v = np.array(line.split()[1:], dtype=np.int64)
                n = int(v[1])
                cur.append((int(v[0]), v[2:2 + n].astype(np.uint32), v[2 + n:2 + n + n - 1].astype(np.uint32)))
    return RefDump(np.array(anchors, np.uint8), np.array(lens, np.uint32), np.array(degs, np.int64), names,
                   np.array(adj, dtype=np.int64) if adj else None, builders, unique, pairs)


def run_ref_dump(rr: str, cr: str, out: str, args: Optional[List[str]] = None) -> RefDump:
    subprocess.run([REF_DUMP] + list(args or []) + [rr, cr, out], check=True, stdout=subprocess.DEVNULL)
    return parse_ref_dump(out)


def have_ref() -> bool:
    return os.path.exists(REF_DUMP)
