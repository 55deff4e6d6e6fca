"""ctypes binding of libtelomeri_hip.so (include/telomeri_hip.h) for the Python harness.

This is plumbing for tests/ and bench.py: device memory comes from torch tensors, every compute call
goes through the C ABI into the HIP kernels.  There is no CPU fallback here: if the library is
missing or no gfx950 device is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# TLM_LIB selects another build of the same library (timing-only ablation builds under profiles/, never tests)
LIB_PATH = os.environ.get("TLM_LIB") or os.path.join(HERE, "lib", "libtelomeri_hip.so")
CSRC = os.path.join(HERE, "csrc")

MODE = {"min": 0, "max": 1, "avg": 2, "sum": 3}
METRIC_ES, METRIC_OS, METRIC_ES_SQRT, METRIC_OS_SQRT = 0, 1, 2, 3
ULONG_MAX = (1 << 64) - 1


class TlmError(RuntimeError):
    """A library call failed.  `code` = the TLM_E* status (include/telomeri_hip.h), None when no call was made."""

    def __init__(self, msg: str, code: Optional[int] = None) -> None:
        super().__init__(msg)
        self.code = code


EPEER = -5      # TLM_EPEER: another rank of the job failed, or this context's communicator was given up


class FilterParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_overlap_length", C.c_uint64), ("min_overlap_percentage", C.c_float),
                ("max_overhang_length", C.c_uint64), ("max_overhang_percentage", C.c_float)]

    @staticmethod
    def default(mode: str = "avg", min_oll: int = 0, min_olp: float = 0.0, max_ohl: int = ULONG_MAX,
                max_ohp: float = 1.0) -> "FilterParams":      # reference Main.cpp:56-62
        return FilterParams(MODE[mode], min_oll, min_olp, max_ohl, max_ohp)


class WalkParams(C.Structure):
    _fields_ = [("rebuild_attempts", C.c_int32), ("backtrack_attempts", C.c_int32), ("len_threshold", C.c_int64)]

    @staticmethod
    def default(rb_att: int = 2000, bt_att: int = 60, len_thr: int = 4_700_000) -> "WalkParams":   # Main.cpp:63-66
        return WalkParams(rb_att, bt_att, len_thr)


class Counters(C.Structure):
    _fields_ = [("units", C.c_uint64), ("steps", C.c_uint64), ("evals", C.c_uint64), ("draws", C.c_uint64),
                ("accepted", C.c_uint64), ("kernel_ms", C.c_double), ("kernel_launches", C.c_uint32),
                ("waves", C.c_uint32), ("kernel_busy_ms", C.c_double), ("layout_bytes", C.c_uint64), ("step_class", C.c_uint64 * 10)]


class _Records(C.Structure):
    _fields_ = [("n", C.c_uint64), ("q_id", C.c_void_p), ("t_id", C.c_void_p), ("q_len", C.c_void_p),
                ("q_start", C.c_void_p), ("q_end", C.c_void_p), ("t_len", C.c_void_p), ("t_start", C.c_void_p),
                ("t_end", C.c_void_p), ("matches", C.c_void_p), ("strand", C.c_void_p), ("anchors", C.c_int32)]


class GraphInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_anchors", C.c_uint32), ("n_adj", C.c_uint64), ("n_records", C.c_uint64),
                ("n_kept", C.c_uint64), ("n_kept_per_file", C.c_uint64 * 4), ("max_degree", C.c_uint32),
                ("unsafe_nodes", C.c_uint32 * 4), ("bfs_levels", C.c_uint32), ("build_ms", C.c_double), ("filter_ms", C.c_double),
                ("phase_ms", C.c_double * 12), ("build_wall_ms", C.c_double), ("build_alloc_ms", C.c_double),
                ("host_syncs", C.c_uint32), ("phase_host_ms", C.c_double * 12), ("sync_wait_ms", C.c_double)]

    def phases(self) -> dict:
        """{phase name: device-timeline ms} of the build (tlm_build_phase_name)."""
        return {lib().tlm_build_phase_name(i).decode(): float(self.phase_ms[i]) for i in range(12)}

    def host_phases(self) -> dict:
        """Host time spent queueing each phase (waits for the device excluded)."""
        return {lib().tlm_build_phase_name(i).decode(): float(self.phase_host_ms[i]) for i in range(12)}


class GraphStats(C.Structure):
    _fields_ = [("anchors", C.c_uint64), ("reads", C.c_uint64), ("min_len", C.c_uint64), ("max_len", C.c_uint64),
                ("min_con", C.c_uint64), ("max_con", C.c_uint64), ("edges", C.c_uint64), ("min_ol", C.c_uint64),
                ("max_ol", C.c_uint64), ("min_os", C.c_float), ("max_os", C.c_float), ("min_es", C.c_float),
                ("max_es", C.c_float), ("min_si", C.c_float), ("max_si", C.c_float)]


EDGE_DTYPE = np.dtype([("q_index", np.uint32), ("t_index", np.uint32), ("q_start", np.uint32), ("q_end", np.uint32),
                       ("t_start", np.uint32), ("t_end", np.uint32), ("overlap_score", np.float32),
                       ("sequence_identity", np.float32), ("extension_score", np.float32), ("relative_strand", np.uint8)],
                      align=True)


class MemoryInfo(C.Structure):
    _fields_ = [("live_bytes", C.c_uint64), ("cached_bytes", C.c_uint64), ("peak_live_bytes", C.c_uint64),
                ("driver_allocs", C.c_uint64), ("reuses", C.c_uint64), ("driver_alloc_ms", C.c_double)]


class ExchangeInfo(C.Structure):
    _fields_ = [("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("slot_bits", C.c_uint32),
                ("device_ms", C.c_double), ("wall_ms", C.c_double)]


UNIT_STAT_DTYPE = np.dtype([("status", np.uint8), ("n_nodes", np.uint32), ("steps", np.uint32), ("evals", np.uint64),
                            ("draws", np.uint32), ("length", np.int64)], align=True)

_lib = None


def build_library() -> None:
    """hipcc cross-compiles for gfx950 without a GPU (telomeri_amd/csrc/Makefile)."""
    subprocess.run(["make", "-s", "-C", CSRC, "-j4"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TlmError("libtelomeri_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C telomeri_amd/csrc` -- there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    V = C.c_void_p
    L.tlm_abi_version.restype = C.c_int
    L.tlm_last_error.restype = C.c_char_p
    L.tlm_ctx_create.argtypes = [C.c_int, P(V)]
    L.tlm_ctx_destroy.argtypes = [V]
    L.tlm_ctx_stream.argtypes = [V]
    L.tlm_ctx_stream.restype = V
    L.tlm_ctx_synchronize.argtypes = [V]
    L.tlm_ctx_set_walk_tuning.argtypes = [V, C.c_int, C.c_uint32]
    L.tlm_ctx_set_arena_limit.argtypes = [V, C.c_uint64]
    L.tlm_ctx_set_walk_accounting.argtypes = [V, C.c_int]
    L.tlm_ctx_memory_info.argtypes = [V, P(MemoryInfo)]
    L.tlm_ctx_trim.argtypes = [V]
    L.tlm_build_phase_name.argtypes = [C.c_int]
    L.tlm_build_phase_name.restype = C.c_char_p
    L.tlm_filter_score.argtypes = [V, P(FilterParams), P(_Records), V, V, V, V, P(C.c_uint64)]
    L.tlm_graph_build.argtypes = [V, P(FilterParams), P(_Records), C.c_int, C.c_uint32, V, P(V)]
    L.tlm_graph_build_host.argtypes = [V, P(FilterParams), P(_Records), C.c_int, C.c_uint32, V, P(V)]
    L.tlm_graph_kept_records.argtypes = [V, V]
    L.tlm_graph_from_csr.argtypes = [V, C.c_uint32, C.c_uint64] + [V] * 12 + [P(V)]
    L.tlm_graph_destroy.argtypes = [V]
    L.tlm_graph_get_info.argtypes = [V, P(GraphInfo)]
    L.tlm_graph_download.argtypes = [V] + [V] * 13
    L.tlm_mc_units.argtypes = [V, P(WalkParams)]
    L.tlm_mc_units.restype = C.c_uint64
    L.tlm_det_units.argtypes = [V]
    L.tlm_det_units.restype = C.c_uint64
    L.tlm_graph_anchor_table.argtypes = [V, V, V]
    L.tlm_walk_seed.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    L.tlm_walk_seed.restype = C.c_uint32
    L.tlm_walk_mc.argtypes = [V, V, C.c_int, P(WalkParams), C.c_uint64, C.c_uint64, P(V), P(Counters)]
    L.tlm_walk_det.argtypes = [V, V, C.c_int, C.c_uint64, C.c_uint64, P(V), P(Counters)]
    L.tlm_walk_mc2.argtypes = [V, V, C.c_int, C.c_int, P(WalkParams), C.c_uint64, C.c_uint64, P(V), P(Counters)]
    L.tlm_walk_det2.argtypes = [V, V, C.c_int, C.c_int, C.c_uint64, C.c_uint64, P(V), P(Counters)]
    L.tlm_walk_builders.argtypes = [V, V, C.c_int, C.c_int, P(WalkParams), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                    P(V), P(Counters)]
    L.tlm_walk_list.argtypes = [V, V, C.c_int, C.c_int, C.c_int, P(WalkParams), C.c_uint64, V, P(V), P(Counters)]
    L.tlm_walk_builders_list.argtypes = [V, V, C.c_int, C.c_int, P(WalkParams), C.c_uint64, V, C.c_uint64, V, P(V), P(Counters)]
    L.tlm_dist_unit_share.argtypes = [V, P(WalkParams), C.c_int, C.c_int, C.c_int, V, P(C.c_uint64)]
    L.tlm_pathset_sizes.argtypes = [V, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
    L.tlm_pathset_export.argtypes = [V, C.c_int, V, V, V, V, V]
    L.tlm_pathset_unit_stats.argtypes = [V, V]
    L.tlm_pathset_destroy.argtypes = [V]
    L.tlm_pathset_concat.argtypes = [V, C.c_int, P(V), P(V)]
    L.tlm_pathset_filter_unique.argtypes = [V, V, P(V)]
    L.tlm_pathset_filter_unique_take.argtypes = [V, V, P(V)]
    L.tlm_pathset_anchor_pairs.argtypes = [V, V, V, V]
    L.tlm_pathset_summary.argtypes = [V, V, V, V, V, V]
    L.tlm_pathset_fetch.argtypes = [V, V, C.c_uint64, V, V, V]
    L.tlm_graph_edges.argtypes = [V, C.c_uint64, V, V, V]
    L.tlm_graph_get_stats.argtypes = [V, P(GraphStats)]
    L.tlm_paf_create.argtypes = [V, P(V)]
    L.tlm_paf_destroy.argtypes = [V]
    L.tlm_paf_destroy.restype = None
    L.tlm_paf_add_file.argtypes = [V, C.c_char_p, C.c_int, P(C.c_uint64), P(C.c_int)]
    L.tlm_paf_add_text.argtypes = [V, C.c_char_p, C.c_uint64, C.c_int, P(C.c_uint64), P(C.c_int)]
    L.tlm_paf_finish.argtypes = [V, P(C.c_uint32), P(C.c_uint64)]
    L.tlm_paf_names.argtypes = [V, V, V, V]
    L.tlm_paf_records.argtypes = [V, C.c_int, P(_Records)]
    L.tlm_paf_build_graph.argtypes = [V, P(FilterParams), P(V)]
    L.tlm_paf_timing.argtypes = [V, P(C.c_double), P(C.c_double), P(C.c_double)]
    L.tlm_paf_share_info.argtypes = [V, P(ExchangeInfo)]
    L.tlm_dist_text_pieces.argtypes = [V, C.c_uint64, P(C.c_uint64), P(C.c_uint64)]
    L.tlm_dist_text_run.argtypes = [C.c_uint64, C.c_int, C.c_int, P(C.c_uint64), P(C.c_uint64)]
    L.tlm_dist_unique_id.argtypes = [V]
    L.tlm_dist_init.argtypes = [V, C.c_int, C.c_int, V]
    L.tlm_dist_finalize.argtypes = [V]
    L.tlm_dist_finalize.restype = None
    L.tlm_dist_rank.argtypes = [V]
    L.tlm_dist_world.argtypes = [V]
    L.tlm_dist_anchor_range.argtypes = [V, C.c_int, C.c_int, P(C.c_uint32), P(C.c_uint32)]
    L.tlm_pathset_allgather.argtypes = [V, C.c_int, P(V), P(V), P(ExchangeInfo)]
    L.tlm_exchange_layout.argtypes = [C.c_int, C.c_int, V, V, V, V]
    L.tlm_pathset_gather_summaries.argtypes = [V, C.c_int, P(V), P(V), P(ExchangeInfo)]
    L.tlm_gathered_destroy.argtypes = [V]
    L.tlm_gathered_destroy.restype = None
    L.tlm_gathered_sizes.argtypes = [V, P(C.c_uint64), P(C.c_uint64)]
    L.tlm_gathered_export.argtypes = [V] + [V] * 9
    L.tlm_gathered_fetch.argtypes = [V, V, C.c_uint64, V, V, V]
    _lib = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise TlmError("%s failed (%d): %s" % (what, rc, lib().tlm_last_error().decode()), rc)


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclasses.dataclass
class HostPaths:
    """Packed Path set on the host: path i = nodes[off[i]:off[i+1]], slots likewise (last slot 0xFFFFFFFF)."""
    unit: np.ndarray
    length: np.ndarray
    off: np.ndarray
    nodes: np.ndarray
    slots: np.ndarray

    def __len__(self) -> int:
        return int(self.unit.shape[0])

    def path(self, i: int):
        a, b = int(self.off[i]), int(self.off[i + 1])
        return int(self.length[i]), self.nodes[a:b], self.slots[a:b - 1]


class PathSet:
    def __init__(self, handle, counters: Counters, ctx: "Context" = None) -> None:
        self._h = handle
        self._ctx = ctx              # the context must outlive the path set
        self.counters = counters
        n_paths, n_nodes, n_units = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().tlm_pathset_sizes(self._h, C.byref(n_paths), C.byref(n_nodes), C.byref(n_units)), "tlm_pathset_sizes")
        self.n_paths, self.n_nodes_total, self.n_units = int(n_paths.value), int(n_nodes.value), int(n_units.value)

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().tlm_pathset_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown
            pass

    def to_host(self) -> HostPaths:
        unit = np.zeros(self.n_paths, np.uint64)
        length = np.zeros(self.n_paths, np.int64)
        off = np.zeros(self.n_paths + 1, np.uint64)
        nodes = np.zeros(self.n_nodes_total, np.uint32)
        slots = np.zeros(self.n_nodes_total, np.uint32)
        _check(lib().tlm_pathset_export(self._h, 0, _np_ptr(unit), _np_ptr(length), _np_ptr(off), _np_ptr(nodes),
                                        _np_ptr(slots)), "tlm_pathset_export")
        return HostPaths(unit, length, off, nodes, slots)

    def to_device_tensors(self, device):
        """(unit, length, off, nodes, slots) as torch tensors on `device` (int64 / int64 / int64 / int32 / int32)."""
        import torch
        unit = torch.empty(self.n_paths, dtype=torch.int64, device=device)
        length = torch.empty(self.n_paths, dtype=torch.int64, device=device)
        off = torch.empty(self.n_paths + 1, dtype=torch.int64, device=device)
        nodes = torch.empty(self.n_nodes_total, dtype=torch.int32, device=device)
        slots = torch.empty(self.n_nodes_total, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)
        _check(lib().tlm_pathset_export(self._h, 1, unit.data_ptr(), length.data_ptr(), off.data_ptr(), nodes.data_ptr(),
                                        slots.data_ptr()), "tlm_pathset_export")
        return unit, length, off, nodes, slots

    def filter_unique(self) -> "PathSet":
        """PathManager::filterUnique on the device."""
        h = C.c_void_p()
        _check(lib().tlm_pathset_filter_unique(self._ctx._h, self._h, C.byref(h)), "tlm_pathset_filter_unique")
        return PathSet(h, self.counters, self._ctx)

    def filter_unique_take(self) -> "PathSet":
        """tlm_pathset_filter_unique_take: consumes this set (it is the result itself when nothing was dropped)."""
        h = C.c_void_p()
        _check(lib().tlm_pathset_filter_unique_take(self._ctx._h, self._h, C.byref(h)), "tlm_pathset_filter_unique_take")
        self._h = None
        return PathSet(h, self.counters, self._ctx)

    def anchor_pairs(self):
        """PathManager::getPathsBetweenAnchors -> [(front node, back node, [path indices])] in the map's order."""
        order = np.zeros(self.n_paths, np.uint32)
        key = np.zeros(self.n_paths, np.uint64)
        _check(lib().tlm_pathset_anchor_pairs(self._ctx._h, self._h, _np_ptr(order), _np_ptr(key)), "tlm_pathset_anchor_pairs")
        out = []
        if self.n_paths:
            cut = np.nonzero(np.diff(key))[0] + 1
            for seg, k in zip(np.split(order, cut), key[np.concatenate([[0], cut])]):
                out.append((int(k >> np.uint64(32)), int(k & np.uint64(0xFFFFFFFF)), seg.tolist()))
        return out

    def summary(self):
        """-> (front, back, n_nodes uint32, hash uint64) per path (tlm_pathset_summary)."""
        front = np.zeros(self.n_paths, np.uint32)
        back = np.zeros(self.n_paths, np.uint32)
        count = np.zeros(self.n_paths, np.uint32)
        h = np.zeros(self.n_paths, np.uint64)
        _check(lib().tlm_pathset_summary(self._ctx._h, self._h, _np_ptr(front), _np_ptr(back), _np_ptr(count), _np_ptr(h)),
               "tlm_pathset_summary")
        return front, back, count, h

    def fetch(self, index, n_nodes):
        """Node and slot lists of the paths `index`, concatenated (n_nodes: the summary's counts)."""
        index = np.ascontiguousarray(index, np.uint64)
        total = int(np.asarray(n_nodes)[index.astype(np.int64)].sum()) if index.size else 0
        nodes = np.zeros(total, np.uint32)
        slots = np.zeros(total, np.uint32)
        _check(lib().tlm_pathset_fetch(self._ctx._h, self._h, index.shape[0], _np_ptr(index), _np_ptr(nodes), _np_ptr(slots)),
               "tlm_pathset_fetch")
        return nodes, slots

    @staticmethod
    def allgather(sets):
        """tlm_pathset_allgather: the one exchange of the multi-GPU build -> (PathSet of all ranks, ExchangeInfo)."""
        arr = (C.c_void_p * len(sets))(*[s._h for s in sets])
        h = C.c_void_p()
        info = ExchangeInfo()
        _check(lib().tlm_pathset_allgather(sets[0]._ctx._h, len(sets), arr, C.byref(h), C.byref(info)), "tlm_pathset_allgather")
        return PathSet(h, sets[0].counters, sets[0]._ctx), info

    @staticmethod
    def concat(sets) -> "PathSet":
        """paths_ after several builder calls (Main.cpp:237-243)."""
        arr = (C.c_void_p * len(sets))(*[s._h for s in sets])
        h = C.c_void_p()
        _check(lib().tlm_pathset_concat(sets[0]._ctx._h, len(sets), arr, C.byref(h)), "tlm_pathset_concat")
        return PathSet(h, sets[0].counters, sets[0]._ctx)

    def unit_stats(self) -> np.ndarray:
        st = np.zeros(max(self.n_units, 1), UNIT_STAT_DTYPE)
        _check(lib().tlm_pathset_unit_stats(self._h, _np_ptr(st)), "tlm_pathset_unit_stats")
        return st[:self.n_units]


@dataclasses.dataclass
class HostCsr:
    n_nodes: int
    n_adj: int
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
    node_name_id: np.ndarray


class Graph:
    def __init__(self, ctx: "Context", handle) -> None:
        self.ctx = ctx
        self._h = handle

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().tlm_graph_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown
            pass

    def info(self) -> GraphInfo:
        gi = GraphInfo()
        _check(lib().tlm_graph_get_info(self._h, C.byref(gi)), "tlm_graph_get_info")
        return gi

    def download(self, cold: bool = True) -> HostCsr:
        gi = self.info()
        n, a = int(gi.n_nodes), int(gi.n_adj)
        row_ptr = np.zeros(n + 1, np.uint64)
        nbr = np.zeros(a, np.uint32)
        qs = np.zeros(a, np.uint32)
        ts = np.zeros(a, np.uint32)
        qe = np.zeros(a, np.uint32) if cold else None
        te = np.zeros(a, np.uint32) if cold else None
        si = np.zeros(a, np.float32) if cold else None
        strand = np.zeros(a, np.uint8) if cold else None
        os_ = np.zeros(a, np.float32)
        es = np.zeros(a, np.float32)
        node_len = np.zeros(n, np.uint32)
        node_anchor = np.zeros(n, np.uint8)
        name_id = np.zeros(n, np.int32)
        _check(lib().tlm_graph_download(self._h, _np_ptr(row_ptr), _np_ptr(nbr), _np_ptr(qs), _np_ptr(qe), _np_ptr(ts),
                                        _np_ptr(te), _np_ptr(os_), _np_ptr(si), _np_ptr(es), _np_ptr(strand),
                                        _np_ptr(node_len), _np_ptr(node_anchor), _np_ptr(name_id)), "tlm_graph_download")
        return HostCsr(n, a, row_ptr, nbr, qs, qe, ts, te, os_, si, es, strand, node_len, node_anchor, name_id)

    def edges(self, nodes, slots) -> np.ndarray:
        """OverlapGraph::Edge records of the (node, slot) pairs (tlm_graph_edges) as a structured array."""
        nodes = np.ascontiguousarray(nodes, np.uint32)
        slots = np.ascontiguousarray(slots, np.uint32)
        out = np.zeros(nodes.shape[0], EDGE_DTYPE)
        _check(lib().tlm_graph_edges(self._h, nodes.shape[0], _np_ptr(nodes), _np_ptr(slots), _np_ptr(out)), "tlm_graph_edges")
        return out

    def stats(self) -> GraphStats:
        st = GraphStats()
        _check(lib().tlm_graph_get_stats(self._h, C.byref(st)), "tlm_graph_get_stats")
        return st

    def anchor_range(self, rank: int, world: int):
        a0, a1 = C.c_uint32(), C.c_uint32()
        _check(lib().tlm_dist_anchor_range(self._h, rank, world, C.byref(a0), C.byref(a1)), "tlm_dist_anchor_range")
        return int(a0.value), int(a1.value)

    def unit_share(self, rank: int, world: int, wp: "WalkParams", deterministic: bool) -> np.ndarray:
        """tlm_dist_unit_share: the units (ascending) of rank's share under the sliced partition."""
        n = C.c_uint64()
        _check(lib().tlm_dist_unit_share(self._h, C.byref(wp), rank, world, int(deterministic), None, C.byref(n)), "tlm_dist_unit_share")
        units = np.zeros(int(n.value), np.uint64)
        _check(lib().tlm_dist_unit_share(self._h, C.byref(wp), rank, world, int(deterministic), _np_ptr(units), C.byref(n)),
               "tlm_dist_unit_share")
        return units

    def walk_list(self, deterministic: bool, metric_a: int, metric_b: int, wp: "WalkParams", units: np.ndarray):
        """tlm_walk_list: one builder over a list of units under one (metric_b < 0) or two metrics."""
        units = np.ascontiguousarray(units, np.uint64)
        h = (C.c_void_p * 2)()
        ctr = (Counters * 2)()
        _check(lib().tlm_walk_list(self.ctx._h, self._h, int(deterministic), metric_a, metric_b, C.byref(wp), units.shape[0],
                                   _np_ptr(units), h, ctr), "tlm_walk_list")
        sets = tuple(PathSet(C.c_void_p(h[k]), ctr[k], self.ctx) for k in range(1 if metric_b < 0 else 2))
        return sets[0] if metric_b < 0 else sets

    def walk_builders_list(self, metric_a: int, metric_b: int, wp: "WalkParams", mc_units: np.ndarray, det_units: np.ndarray):
        """tlm_walk_builders_list: MC a, MC b, DET a, DET b over lists of units."""
        mc_units = np.ascontiguousarray(mc_units, np.uint64)
        det_units = np.ascontiguousarray(det_units, np.uint64)
        h = (C.c_void_p * 4)()
        ctr = (Counters * 4)()
        _check(lib().tlm_walk_builders_list(self.ctx._h, self._h, metric_a, metric_b, C.byref(wp), mc_units.shape[0],
                                            _np_ptr(mc_units), det_units.shape[0], _np_ptr(det_units), h, ctr), "tlm_walk_builders_list")
        return tuple(PathSet(C.c_void_p(h[k]), ctr[k], self.ctx) for k in range(4))

    def kept_records(self) -> np.ndarray:
        out = np.zeros(int(self.info().n_kept), np.uint32)
        _check(lib().tlm_graph_kept_records(self._h, _np_ptr(out)), "tlm_graph_kept_records")
        return out

    def mc_units(self, wp: WalkParams) -> int:
        return int(lib().tlm_mc_units(self._h, C.byref(wp)))

    def det_units(self) -> int:
        return int(lib().tlm_det_units(self._h))

    def anchor_table(self):
        """-> (anchor node indices uint32 [n_anchors], DET unit prefix uint64 [n_anchors+1])"""
        na = int(self.info().n_anchors)
        anchors = np.zeros(na, np.uint32)
        det_off = np.zeros(na + 1, np.uint64)
        _check(lib().tlm_graph_anchor_table(self._h, _np_ptr(anchors), _np_ptr(det_off)), "tlm_graph_anchor_table")
        return anchors, det_off

    def walk_mc(self, metric: int, wp: WalkParams, unit_begin: int = 0, unit_end: Optional[int] = None) -> PathSet:
        if unit_end is None:
            unit_end = self.mc_units(wp)
        h = C.c_void_p()
        ctr = Counters()
        _check(lib().tlm_walk_mc(self.ctx._h, self._h, metric, C.byref(wp), unit_begin, unit_end, C.byref(h),
                                 C.byref(ctr)), "tlm_walk_mc")
        return PathSet(h, ctr, self.ctx)

    def walk_mc2(self, metric_a: int, metric_b: int, wp: WalkParams, unit_begin: int = 0, unit_end: Optional[int] = None):
        """Both metrics in one persistent launch -> (PathSet_a, PathSet_b); kernel time is booked on the first."""
        if unit_end is None:
            unit_end = self.mc_units(wp)
        h = (C.c_void_p * 2)()
        ctr = (Counters * 2)()
        _check(lib().tlm_walk_mc2(self.ctx._h, self._h, metric_a, metric_b, C.byref(wp), unit_begin, unit_end, h, ctr),
               "tlm_walk_mc2")
        return PathSet(C.c_void_p(h[0]), ctr[0], self.ctx), PathSet(C.c_void_p(h[1]), ctr[1], self.ctx)

    def walk_det2(self, metric_a: int, metric_b: int, unit_begin: int = 0, unit_end: Optional[int] = None):
        if unit_end is None:
            unit_end = self.det_units()
        h = (C.c_void_p * 2)()
        ctr = (Counters * 2)()
        _check(lib().tlm_walk_det2(self.ctx._h, self._h, metric_a, metric_b, unit_begin, unit_end, h, ctr), "tlm_walk_det2")
        return PathSet(C.c_void_p(h[0]), ctr[0], self.ctx), PathSet(C.c_void_p(h[1]), ctr[1], self.ctx)

    def walk_builders(self, metric_a: int, metric_b: int, wp: "WalkParams", mc_range=None, det_range=None):
        """tlm_walk_builders: MC a, MC b, DET a, DET b (the deterministic launch fills the Monte-Carlo launch's tail)."""
        mc0, mc1 = mc_range if mc_range is not None else (0, self.mc_units(wp))
        d0, d1 = det_range if det_range is not None else (0, self.det_units())
        h = (C.c_void_p * 4)()
        ctr = (Counters * 4)()
        _check(lib().tlm_walk_builders(self.ctx._h, self._h, metric_a, metric_b, C.byref(wp), mc0, mc1, d0, d1, h, ctr),
               "tlm_walk_builders")
        return tuple(PathSet(C.c_void_p(h[k]), ctr[k], self.ctx) for k in range(4))

    def walk_det(self, metric: int, unit_begin: int = 0, unit_end: Optional[int] = None) -> PathSet:
        if unit_end is None:
            unit_end = self.det_units()
        h = C.c_void_p()
        ctr = Counters()
        _check(lib().tlm_walk_det(self.ctx._h, self._h, metric, unit_begin, unit_end, C.byref(h), C.byref(ctr)),
               "tlm_walk_det")
        return PathSet(h, ctr, self.ctx)


def _records_struct(rec, anchors: bool) -> _Records:
    """rec: telomeri_amd.synth.Records (or anything with those int32 device tensors)."""
    import torch
    cols = {}
    for name in ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches"):
        t = getattr(rec, name)
        assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous(), name
        cols[name] = t.data_ptr()
    strand = rec.strand
    if strand.dtype != torch.uint8:
        strand = strand.to(torch.uint8)
        rec._strand_u8 = strand          # keep it alive
    return _Records(len(rec), cols["q_id"], cols["t_id"], cols["q_len"], cols["q_start"], cols["q_end"], cols["t_len"],
                    cols["t_start"], cols["t_end"], cols["matches"], strand.data_ptr(), int(anchors))


class Context:
    def __init__(self, device: int = 0) -> None:
        # Harness detail: PyTorch's wheel carries its own copy of the HIP / HSA runtime, this library uses the system's.
        # The two coexist in one process when PyTorch's comes up first; the other way round its device discovery fails
        # ("No HIP GPUs are available") and the process aborts at exit.  The C++ hosts never load PyTorch.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        h = C.c_void_p()
        _check(lib().tlm_ctx_create(device, C.byref(h)), "tlm_ctx_create")
        self._h = h
        self.device = device

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().tlm_ctx_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown
            pass

    def set_walk_tuning(self, waves_per_cu: int = 0, path_capacity: int = 0) -> None:
        _check(lib().tlm_ctx_set_walk_tuning(self._h, waves_per_cu, path_capacity), "tlm_ctx_set_walk_tuning")

    def dist_init(self, rank: int, world: int, unique_id: bytes) -> None:
        """One RCCL communicator per context; `unique_id` = the 128 bytes rank 0 got from dist_unique_id()."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(lib().tlm_dist_init(self._h, rank, world, buf), "tlm_dist_init")

    def dist_finalize(self) -> None:
        lib().tlm_dist_finalize(self._h)

    def set_walk_accounting(self, on: bool) -> None:
        """Walk calls launch the builds of the kernels that also count layout bytes (Counters.layout_bytes); never timed."""
        _check(lib().tlm_ctx_set_walk_accounting(self._h, int(bool(on))), "tlm_ctx_set_walk_accounting")

    def release_workspaces(self) -> None:
        """tlm_ctx_release_workspaces: the walk workspaces go back to the library's cache (after the last walk call)."""
        _check(lib().tlm_ctx_release_workspaces(self._h), "tlm_ctx_release_workspaces")

    def set_arena_limit(self, nbytes: int) -> None:
        _check(lib().tlm_ctx_set_arena_limit(self._h, nbytes), "tlm_ctx_set_arena_limit")

    def memory_info(self) -> MemoryInfo:
        mi = MemoryInfo()
        _check(lib().tlm_ctx_memory_info(self._h, C.byref(mi)), "tlm_ctx_memory_info")
        return mi

    def trim(self) -> None:
        _check(lib().tlm_ctx_trim(self._h), "tlm_ctx_trim")

    def filter_score(self, fp: FilterParams, rec, want_scores: bool = True):
        """-> (keep uint8, si, os, es float32) device tensors, n_kept."""
        import torch
        dev = rec.q_id.device
        n = len(rec)
        keep = torch.empty(n, dtype=torch.uint8, device=dev)
        si = torch.empty(n, dtype=torch.float32, device=dev) if want_scores else None
        os_ = torch.empty(n, dtype=torch.float32, device=dev) if want_scores else None
        es = torch.empty(n, dtype=torch.float32, device=dev) if want_scores else None
        rs = _records_struct(rec, False)
        torch.cuda.synchronize(dev)
        nk = C.c_uint64()
        _check(lib().tlm_filter_score(self._h, C.byref(fp), C.byref(rs), keep.data_ptr(),
                                      si.data_ptr() if want_scores else None, os_.data_ptr() if want_scores else None,
                                      es.data_ptr() if want_scores else None, C.byref(nk)), "tlm_filter_score")
        return keep, si, os_, es, int(nk.value)

    def build_graph(self, fp: FilterParams, files: Sequence, anchors: Sequence[bool], n_names: int, name_is_ctg) -> Graph:
        """files: Records on this device, in load order (reference Main.cpp:223: rr then cr)."""
        import torch
        arr = (_Records * len(files))(*[_records_struct(f, a) for f, a in zip(files, anchors)])
        ctg = name_is_ctg.to(device=files[0].q_id.device, dtype=torch.uint8).contiguous()
        torch.cuda.synchronize(files[0].q_id.device)
        h = C.c_void_p()
        _check(lib().tlm_graph_build(self._h, C.byref(fp), arr, len(files), n_names, ctg.data_ptr(), C.byref(h)),
               "tlm_graph_build")
        return Graph(self, h)

    def build_graph_host(self, fp: FilterParams, files: Sequence[dict], anchors: Sequence[bool], n_names: int,
                         name_is_ctg) -> Graph:
        """files: dicts of HOST int32 numpy columns (q_id, t_id, q_len, ..., matches) + uint8 strand."""
        keep = []
        recs = []
        for f, a in zip(files, anchors):
            cols = {k: np.ascontiguousarray(f[k], np.int32) for k in ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len",
                                                                       "t_start", "t_end", "matches")}
            strand = np.ascontiguousarray(f["strand"], np.uint8)
            keep.append((cols, strand))
            recs.append(_Records(cols["q_id"].shape[0], *[cols[k].ctypes.data for k in (
                "q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches")], strand.ctypes.data,
                int(a)))
        arr = (_Records * len(recs))(*recs)
        ctg = np.ascontiguousarray(name_is_ctg, np.uint8)
        h = C.c_void_p()
        _check(lib().tlm_graph_build_host(self._h, C.byref(fp), arr, len(recs), n_names, _np_ptr(ctg), C.byref(h)),
               "tlm_graph_build_host")
        return Graph(self, h)

    def graph_from_csr(self, row_ptr, nbr, q_start, t_start, os_, es, node_len, node_anchor, q_end=None, t_end=None,
                       si=None, strand=None) -> Graph:
        row_ptr = np.ascontiguousarray(row_ptr, np.uint64)
        u32 = lambda x: None if x is None else np.ascontiguousarray(x, np.uint32)
        f32 = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)
        u8 = lambda x: None if x is None else np.ascontiguousarray(x, np.uint8)
        nbr, q_start, t_start, q_end, t_end = u32(nbr), u32(q_start), u32(t_start), u32(q_end), u32(t_end)
        os_, es, si = f32(os_), f32(es), f32(si)
        node_len, node_anchor, strand = u32(node_len), u8(node_anchor), u8(strand)
        h = C.c_void_p()
        _check(lib().tlm_graph_from_csr(self._h, row_ptr.shape[0] - 1, nbr.shape[0], _np_ptr(row_ptr), _np_ptr(nbr),
                                        _np_ptr(q_start), _np_ptr(q_end), _np_ptr(t_start), _np_ptr(t_end), _np_ptr(os_),
                                        _np_ptr(si), _np_ptr(es), _np_ptr(strand), _np_ptr(node_len), _np_ptr(node_anchor),
                                        C.byref(h)), "tlm_graph_from_csr")
        return Graph(self, h)


class Gathered:
    """tlm_pathset_gather_summaries: the lazy exchange -- summaries of every path of the job, node lists on demand."""

    def __init__(self, sets):
        self.ctx = sets[0]._ctx
        arr = (C.c_void_p * len(sets))(*[s._h for s in sets])
        h = C.c_void_p()
        self.info = ExchangeInfo()
        _check(lib().tlm_pathset_gather_summaries(self.ctx._h, len(sets), arr, C.byref(h), C.byref(self.info)),
               "tlm_pathset_gather_summaries")
        self._h = h
        n, nl = C.c_uint64(), C.c_uint64()
        _check(lib().tlm_gathered_sizes(self._h, C.byref(n), C.byref(nl)), "tlm_gathered_sizes")
        self.n_paths, self.n_local_paths = int(n.value), int(nl.value)

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().tlm_gathered_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def export(self) -> dict:
        n = self.n_paths
        out = {"unit": np.zeros(n, np.uint64), "length": np.zeros(n, np.int64), "front": np.zeros(n, np.uint32),
               "back": np.zeros(n, np.uint32), "n_nodes": np.zeros(n, np.uint32), "hash": np.zeros(n, np.uint64),
               "owner": np.zeros(n, np.uint32), "dup_of_prev": np.zeros(n, np.uint8), "block_first": np.zeros(n, np.uint8)}
        _check(lib().tlm_gathered_export(self._h, *[_np_ptr(out[k]) for k in (
            "unit", "length", "front", "back", "n_nodes", "hash", "owner", "dup_of_prev", "block_first")]), "tlm_gathered_export")
        return out

    def fetch(self, index, n_nodes):
        index = np.ascontiguousarray(index, np.uint64)
        total = int(np.asarray(n_nodes)[index.astype(np.int64)].sum()) if index.size else 0
        nodes, slots = np.zeros(total, np.uint32), np.zeros(total, np.uint32)
        _check(lib().tlm_gathered_fetch(self.ctx._h, self._h, index.shape[0], _np_ptr(index), _np_ptr(nodes), _np_ptr(slots)),
               "tlm_gathered_fetch")
        return nodes, slots


class PafTokens:
    """tlm_paf_*: the PAF files of one graph, tokenised on the device."""

    def __init__(self, ctx: "Context") -> None:
        self.ctx = ctx
        h = C.c_void_p()
        _check(lib().tlm_paf_create(ctx._h, C.byref(h)), "tlm_paf_create")
        self._h = h
        self.n_files = 0

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().tlm_paf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def add_file(self, path: str, anchors: bool):
        """-> (records, clean)"""
        n, clean = C.c_uint64(), C.c_int()
        _check(lib().tlm_paf_add_file(self._h, path.encode(), int(anchors), C.byref(n), C.byref(clean)), "tlm_paf_add_file")
        self.n_files += bool(clean.value)
        return int(n.value), bool(clean.value)

    def add_text(self, text: bytes, anchors: bool):
        n, clean = C.c_uint64(), C.c_int()
        _check(lib().tlm_paf_add_text(self._h, text, len(text), int(anchors), C.byref(n), C.byref(clean)), "tlm_paf_add_text")
        self.n_files += bool(clean.value)
        return int(n.value), bool(clean.value)

    def finish(self):
        """-> (names list[str], is_ctg uint8 array)"""
        n, nb = C.c_uint32(), C.c_uint64()
        _check(lib().tlm_paf_finish(self._h, C.byref(n), C.byref(nb)), "tlm_paf_finish")
        off = np.zeros(n.value + 1, np.uint32)
        blob = np.zeros(max(int(nb.value), 1), np.uint8)
        ctg = np.zeros(max(n.value, 1), np.uint8)
        _check(lib().tlm_paf_names(self._h, _np_ptr(off), _np_ptr(blob), _np_ptr(ctg)), "tlm_paf_names")
        raw = blob.tobytes()
        return [raw[off[i]:off[i + 1]].decode("latin-1") for i in range(n.value)], ctg[:n.value]

    def records(self, file: int) -> "_Records":
        r = _Records()
        _check(lib().tlm_paf_records(self._h, file, C.byref(r)), "tlm_paf_records")
        return r

    def columns(self, file: int) -> dict:
        """The file's columns copied to the host (tests)."""
        import torch
        r = self.records(file)
        out = {}
        for name in ("q_id", "t_id", "q_len", "q_start", "q_end", "t_len", "t_start", "t_end", "matches", "strand"):
            dt = torch.uint8 if name == "strand" else torch.int32
            t = torch.empty(int(r.n), dtype=dt, device="cuda:%d" % self.ctx.device)
            src = getattr(r, name)
            if r.n:
                hip = C.CDLL("libamdhip64.so")
                hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                rc = hip.hipMemcpy(t.data_ptr(), src, int(r.n) * t.element_size(), 3)       # device to device
                assert rc == 0, rc
            out[name] = t.cpu().numpy()
        return out

    def build_graph(self, fp: FilterParams) -> "Graph":
        h = C.c_void_p()
        _check(lib().tlm_paf_build_graph(self._h, C.byref(fp), C.byref(h)), "tlm_paf_build_graph")
        return Graph(self.ctx, h)

    def share_info(self) -> "ExchangeInfo":
        """What the sharded load (ABI 5) moved between the ranks for this handle's files (zeros at world 1)."""
        info = ExchangeInfo()
        _check(lib().tlm_paf_share_info(self._h, C.byref(info)), "tlm_paf_share_info")
        return info


def dist_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    _check(lib().tlm_dist_unique_id(buf), "tlm_dist_unique_id")
    return bytes(buf)


def exchange_layout(counts: np.ndarray):
    """counts[world, n_sets, 2] (paths, nodes) -> path_base[world, n_sets], node_base[world, n_sets], (paths, nodes) totals."""
    counts = np.ascontiguousarray(counts, np.uint64)
    world, n_sets, _ = counts.shape
    pb = np.zeros((world, n_sets), np.uint64)
    nb = np.zeros((world, n_sets), np.uint64)
    tot = np.zeros(2, np.uint64)
    _check(lib().tlm_exchange_layout(world, n_sets, _np_ptr(counts), _np_ptr(pb), _np_ptr(nb), _np_ptr(tot)), "tlm_exchange_layout")
    return pb, nb, (int(tot[0]), int(tot[1]))


def walk_seed(metric: int, anchor_node: int, r: int) -> int:
    return int(lib().tlm_walk_seed(metric, anchor_node, r))


def declared_symbols() -> List[str]:
    """Every function include/telomeri_hip.h declares."""
    import re
    hdr = os.path.join(HERE, "..", "include", "telomeri_hip.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tlm_[a-z0-9_]+)\s*\(", txt)))
