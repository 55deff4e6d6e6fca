"""CPU tier: the RCCL stand-in behind the multi-rank GPU tests (tests/fake_rccl, TEST INFRASTRUCTURE) is itself tested --
its host-only build (buffers in host memory, no HIP call) is driven by several processes through ctypes: all-gather,
broadcast with several roots in one group, grouped send / receive between every pair, and the checks real RCCL does not make
for you: a size mismatch and an unmatched send are errors on BOTH sides, a rank that never arrives is a time-out, a rank that
failed takes its peers' next call down with it."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl_host.so")


@pytest.fixture(scope="module")
def lib_path():
    subprocess.run(["make", "-s", "-C", os.path.dirname(LIB), "host"], check=True)
    return LIB


class UniqueId(C.Structure):           # ncclUniqueId: passed BY VALUE to ncclCommInitRank
    _fields_ = [("internal", C.c_char * 128)]


def _api(path):
    L = C.CDLL(path)
    V = C.c_void_p
    L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(V), C.c_int, UniqueId, C.c_int]
    for f in ("ncclCommDestroy", "ncclCommAbort"):
        getattr(L, f).argtypes = [V]
    L.ncclAllGather.argtypes = [V, V, C.c_size_t, C.c_int, V, V]
    L.ncclBroadcast.argtypes = [V, V, C.c_size_t, C.c_int, C.c_int, V, V]
    L.ncclSend.argtypes = [V, C.c_size_t, C.c_int, C.c_int, V, V]
    L.ncclRecv.argtypes = [V, C.c_size_t, C.c_int, C.c_int, V, V]
    L.ncclGetErrorString.restype = C.c_char_p
    L.ncclGetErrorString.argtypes = [C.c_int]
    return L


U8 = 1      # ncclUint8


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _rank(path, uid, rank, world, scenario, q):
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "3"
    L = _api(path)
    comm = C.c_void_p()
    ident = UniqueId.from_buffer_copy(uid)
    if scenario == "absent" and rank == world - 1:
        q.put((rank, "stayed away"))
        return
    rc = L.ncclCommInitRank(C.byref(comm), world, ident, rank)
    if rc != 0:
        q.put((rank, "init:%d" % rc))
        return
    out = []
    try:
        # ---- all-gather
        mine = np.full(5, rank + 1, np.uint8)
        allv = np.zeros(5 * world, np.uint8)
        out.append(L.ncclAllGather(_ptr(mine), _ptr(allv), 5, U8, comm, None))
        out.append(allv.tolist() == sum(([r + 1] * 5 for r in range(world)), []))
        # ---- grouped broadcasts, every rank a root once (in place)
        bufs = [np.full(3 + r, 10 * (rank + 1) + r, np.uint8) for r in range(world)]
        L.ncclGroupStart()
        for r in range(world):
            L.ncclBroadcast(_ptr(bufs[r]), _ptr(bufs[r]), bufs[r].size, U8, r, comm, None)
        out.append(L.ncclGroupEnd())
        out.append(all(bufs[r].tolist() == [10 * (r + 1) + r] * (3 + r) for r in range(world)))
        # ---- grouped send / receive between every pair, two messages each way, different sizes
        sends = {p: [np.full(2 + rank, 100 + 10 * rank + p, np.uint8), np.full(7, 200 + rank, np.uint8)] for p in range(world) if p != rank}
        recvs = {p: [np.zeros(2 + p, np.uint8), np.zeros(7, np.uint8)] for p in range(world) if p != rank}
        if scenario == "size_mismatch" and rank == 0:
            recvs[1][1] = np.zeros(9, np.uint8)             # expects 9 bytes where rank 1 sends 7
        L.ncclGroupStart()
        for p in sends:
            for k in range(2):
                if scenario == "unmatched_send" and rank == 1 and p == 0 and k == 1:
                    continue                                 # rank 1 never posts its second send to rank 0 ...
                L.ncclSend(_ptr(sends[p][k]), sends[p][k].size, U8, p, comm, None)
            for k in range(2):
                if scenario == "unmatched_recv" and rank == 0 and p == 1 and k == 1:
                    continue                                 # ... or rank 0 never posts its second receive from rank 1
                L.ncclRecv(_ptr(recvs[p][k]), recvs[p][k].size, U8, p, comm, None)
        rc = L.ncclGroupEnd()
        out.append(rc)
        if rc == 0:
            out.append(all(recvs[p][0].tolist() == [100 + 10 * p + rank] * (2 + p) and recvs[p][1].tolist() == [200 + p] * 7 for p in recvs))
        else:
            out.append(L.ncclGetErrorString(rc).decode())
            # after a failure the next collective fails at once on every rank
            out.append(L.ncclAllGather(_ptr(mine), _ptr(allv), 5, U8, comm, None))
    finally:
        L.ncclCommDestroy(comm)
    q.put((rank, out))


def _run(path, world, scenario):
    L = _api(path)
    uid = UniqueId()
    assert L.ncclGetUniqueId(C.byref(uid)) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(path, bytes(uid), r, world, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    leftovers = [f for f in os.listdir("/dev/shm") if f.startswith(bytes(uid).split(b"\0")[0].decode())]
    return res, leftovers


@pytest.mark.parametrize("world", [2, 3])
def test_collectives_between_processes(lib_path, world):
    res, leftovers = _run(lib_path, world, "ok")
    for r in range(world):
        assert res[r] == [0, True, 0, True, 0, True], (r, res[r])
    assert leftovers == []


@pytest.mark.parametrize("scenario,needle", [("size_mismatch", "size mismatch"), ("unmatched_send", "no matching operation"),
                                             ("unmatched_recv", "without a matching ncclRecv")])
def test_pairing_is_checked_on_both_sides(lib_path, scenario, needle):
    res, _ = _run(lib_path, 2, scenario)
    texts = []
    for r in (0, 1):
        assert res[r][:4] == [0, True, 0, True]
        assert res[r][4] != 0 and res[r][6] != 0, (r, res[r])       # the group failed, and so did the next call
        texts.append(res[r][5])
    assert any(needle in t for t in texts), texts
    assert all("fake_rccl" in t for t in texts)


def test_a_rank_that_never_arrives_is_a_timeout(lib_path):
    res, _ = _run(lib_path, 2, "absent")
    assert res[1] == "stayed away"
    assert res[0].startswith("init:") and res[0] != "init:0"
