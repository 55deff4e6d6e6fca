"""The multi-rank path on CPU: world_size 2 over gloo.  Each rank walks its anchor bundle
(here with the oracle standing in for the device, as the checker), the packed Path sets cross ONE
all_gather, and every rank must hold exactly the Path set a single rank produces."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _packed(ps):
    from telomeri_amd.pathgather import PackedPaths
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
    return PackedPaths(t(ps.unit, np.int64), t(ps.length, np.int64), t(ps.off, np.int64), t(ps.nodes, np.int32),
                       t(ps.slots, np.int32))


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers as H
    from oracle import pyoracle as po
    from telomeri_amd import pathgather, synth
    ds = synth.make_config("tiny")
    og = H.oracle_graph(ds, po.FilterParams.default())
    wp = po.WalkParams.default(24)
    c = og.csr()
    anchors = np.nonzero(c.node_anchor)[0]
    a0, a1 = pathgather.shard_range(len(anchors), rank, world)
    # DET unit prefix per anchor, as tlm_graph_anchor_table reports it
    det_cnt = []
    for a in anchors:
        r0, r1 = int(c.row_ptr[a]), int(c.row_ptr[a + 1])
        hit = np.nonzero(c.node_anchor[c.nbr[r0:r1]])[0]
        det_cnt.append(int(hit[0]) + 1 if hit.size else r1 - r0)
    det_off = np.concatenate([[0], np.cumsum(det_cnt)])
    local = []
    for metric in (po.METRIC_ES, po.METRIC_OS):
        local.append(_packed(og.walk_mc(metric, wp, po.SEED_PER_WALK, a0 * 24, a1 * 24)[0]))
    for metric in (po.METRIC_ES, po.METRIC_OS):
        local.append(_packed(og.walk_det(metric, int(det_off[a0]), int(det_off[a1]))[0]))
    full = pathgather.allgather_paths(local)
    # reference result: all units on one rank
    want = [og.walk_mc(m, wp, po.SEED_PER_WALK)[0] for m in (0, 1)] + [og.walk_det(m)[0] for m in (0, 1)]
    ok = True
    for got, w in zip(full, want):
        ok &= np.array_equal(got.unit.numpy(), w.unit.astype(np.int64))
        ok &= np.array_equal(got.length.numpy(), w.length)
        ok &= np.array_equal(got.off.numpy(), w.off.astype(np.int64))
        ok &= np.array_equal(got.nodes.numpy(), w.nodes.astype(np.int32))
        ok &= np.array_equal(got.slots.numpy(), w.slots.view(np.int32))
    n_paths = sum(p.n_paths for p in full)
    open(os.path.join(out_dir, "rank%d" % rank), "w").write("%d %d" % (int(ok), n_paths))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_allgather_paths_world2_gloo():
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_worker, args=(2, _free_port(), out_dir), nprocs=2, join=True)
        res = [open(os.path.join(out_dir, "rank%d" % r)).read().split() for r in range(2)]
    assert res[0] == res[1]
    assert res[0][0] == "1" and int(res[0][1]) > 20


def test_pack_unpack_roundtrip_and_empty():
    sys.path.insert(0, ROOT)
    from telomeri_amd.pathgather import PackedPaths, concat, pack, unpack, shard_range
    a = PackedPaths(torch.tensor([3, 9]), torch.tensor([100, -5]), torch.tensor([0, 3, 5]),
                    torch.tensor([1, 2, 3, 7, 8], dtype=torch.int32), torch.tensor([0, 4, -1, 2, -1], dtype=torch.int32))
    e = PackedPaths(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), torch.zeros(1, dtype=torch.int64),
                    torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32))
    sizes, payload = pack([a, e, a])
    back = unpack(sizes, payload)
    for x, y in zip(back, [a, e, a]):
        for f in ("unit", "length", "off", "nodes", "slots"):
            assert torch.equal(getattr(x, f), getattr(y, f)), f
    c = concat([a, e, a])
    assert c.off.tolist() == [0, 3, 5, 8, 10] and c.n_paths == 4
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
