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
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]


def test_exchange_layout_of_the_abi_matches_the_gather_order():
    """tlm_exchange_layout (host arithmetic of the product's RCCL exchange, telomeri_amd/csrc/exchange.hip): placing
    every (rank, builder) block where it says yields the same Path set as this module's gather + per-builder concat,
    concatenated builder by builder (tlm_pathset_concat order).  World sizes 1..8, empty blocks included."""
    sys.path.insert(0, ROOT)
    from telomeri_amd import capi
    from telomeri_amd.pathgather import PackedPaths, concat
    rng = np.random.default_rng(7)
    for world in (1, 2, 3, 8):
        n_sets = 4
        blocks = {}
        counts = np.zeros((world, n_sets, 2), np.uint64)
        for r in range(world):
            for b in range(n_sets):
                n_p = int(rng.integers(0, 5)) if (r + b) % 3 else 0
                lens = rng.integers(2, 9, size=n_p)
                off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                nodes = rng.integers(0, 1000, size=int(off[-1])).astype(np.int32)
                slots = rng.integers(0, 300, size=int(off[-1])).astype(np.int32)
                unit = (np.arange(n_p) + 1000 * r).astype(np.int64)
                blocks[r, b] = PackedPaths(torch.from_numpy(unit), torch.from_numpy(unit * 7), torch.from_numpy(off),
                                           torch.from_numpy(nodes), torch.from_numpy(slots))
                counts[r, b] = (n_p, int(off[-1]))
        pb, nb, (tot_p, tot_n) = capi.exchange_layout(counts)
        unit = np.full(tot_p, -1, np.int64)
        off = np.full(tot_p + 1, -1, np.int64)
        nodes = np.full(tot_n, -1, np.int32)
        for (r, b), blk in blocks.items():
            p0, n0 = int(pb[r, b]), int(nb[r, b])
            unit[p0:p0 + blk.n_paths] = blk.unit.numpy() | (b << 56)
            off[p0:p0 + blk.n_paths] = blk.off.numpy()[:-1] + n0
            nodes[n0:n0 + blk.n_nodes] = blk.nodes.numpy()
        off[tot_p] = tot_n
        per_builder = [concat([blocks[r, b] for r in range(world)]) for b in range(n_sets)]
        want = concat(per_builder)
        want_unit = np.concatenate([pbk.unit.numpy() | (b << 56) for b, pbk in enumerate(per_builder)])
        assert tot_p == want.n_paths and tot_n == want.n_nodes
        np.testing.assert_array_equal(unit, want_unit)
        np.testing.assert_array_equal(off, want.off.numpy())
        np.testing.assert_array_equal(nodes, want.nodes.numpy())
