"""The one exchange of the path: an all-gather of the packed Path sets.

Anchors (and with them their Monte-Carlo trial bundles / first-edge walks) are dealt to the ranks in
contiguous ranges, every rank walks its units on its own replica of the CSR with no communication, and
then one variable-length all-gather hands every rank the whole Path set, ordered as the reference's
`paths_` vector is: builder by builder ([MC/ES][MC/OS][DET/ES][DET/OS], Main.cpp:237-243), inside a
builder by unit (anchor in node order, then trial / first-edge position).

This module is the harness's rehearsal of that exchange over torch.distributed ("gloo" in the CPU tests): the
payload of all builders travels in ONE all_gather of a padded int64 buffer per rank; a second, tiny all_gather
carries the sizes.  The PRODUCT's exchange is tlm_pathset_allgather in the C ABI (telomeri_amd/csrc/exchange.hip:
RCCL send/recv of the packed device buffers, every block received in its final place, 6 bytes per path node on the
wire); bench.py uses that one on the GPUs, and both produce the same Path set (tests/test_pathgather_gloo.py checks
this module against the ABI's placement rule, tlm_exchange_layout).
"""
from __future__ import annotations

import dataclasses
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclasses.dataclass
class PackedPaths:
    """Packed Path set as tensors (any device): path i = nodes[off[i]:off[i+1]]."""
    unit: torch.Tensor      # int64 [P]
    length: torch.Tensor    # int64 [P]
    off: torch.Tensor       # int64 [P+1]
    nodes: torch.Tensor     # int32 [T]
    slots: torch.Tensor     # int32 [T]

    @property
    def n_paths(self) -> int:
        return int(self.unit.shape[0])

    @property
    def n_nodes(self) -> int:
        return int(self.nodes.shape[0])


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous bundles of ceil(n_items / world) anchors: the rule of tlm_dist_anchor_range."""
    per = (n_items + world - 1) // world
    return min(n_items, rank * per), min(n_items, rank * per + per)


def _pad64(n32: int) -> int:
    return (n32 + 1) // 2


def pack(sets: Sequence[PackedPaths]) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (sizes int64 [2*B], payload int64 [L]) for B builders."""
    dev = sets[0].unit.device
    sizes = torch.tensor([v for s in sets for v in (s.n_paths, s.n_nodes)], dtype=torch.int64, device=dev)
    parts: List[torch.Tensor] = []
    for s in sets:
        counts = (s.off[1:] - s.off[:-1]).to(torch.int64)
        parts += [s.unit.to(torch.int64), s.length.to(torch.int64), counts]
        for t in (s.nodes, s.slots):
            t = t.to(torch.int32).contiguous()
            if t.shape[0] == 0:
                continue
            if t.shape[0] % 2:
                t = torch.cat([t, torch.zeros(1, dtype=torch.int32, device=dev)])
            parts.append(t.view(torch.int64))
    payload = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
    return sizes, payload


def unpack(sizes: torch.Tensor, payload: torch.Tensor) -> List[PackedPaths]:
    out: List[PackedPaths] = []
    sz = sizes.tolist()
    p = 0
    for b in range(len(sz) // 2):
        n_p, n_n = sz[2 * b], sz[2 * b + 1]
        unit = payload[p:p + n_p]; p += n_p
        length = payload[p:p + n_p]; p += n_p
        counts = payload[p:p + n_p]; p += n_p
        w = _pad64(n_n)
        nodes = payload[p:p + w].view(torch.int32)[:n_n]; p += w
        slots = payload[p:p + w].view(torch.int32)[:n_n]; p += w
        off = torch.zeros(n_p + 1, dtype=torch.int64, device=payload.device)
        torch.cumsum(counts, 0, out=off[1:])
        out.append(PackedPaths(unit, length, off, nodes, slots))
    return out


def concat(parts: Sequence[PackedPaths]) -> PackedPaths:
    """Concatenate per-rank sets of one builder in rank order (= unit order for contiguous shards)."""
    unit = torch.cat([p.unit for p in parts])
    length = torch.cat([p.length for p in parts])
    nodes = torch.cat([p.nodes for p in parts])
    slots = torch.cat([p.slots for p in parts])
    counts = torch.cat([p.off[1:] - p.off[:-1] for p in parts])
    off = torch.zeros(unit.shape[0] + 1, dtype=torch.int64, device=unit.device)
    torch.cumsum(counts, 0, out=off[1:])
    return PackedPaths(unit, length, off, nodes, slots)


def allgather_paths(local: Sequence[PackedPaths], group=None) -> List[PackedPaths]:
    """Every rank contributes its Path sets (one per builder); every rank gets the whole set per builder."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(local)
    world = dist.get_world_size(group)
    sizes, payload = pack(local)
    meta = torch.cat([sizes, torch.tensor([payload.shape[0]], dtype=torch.int64, device=sizes.device)])
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_len = max(int(m[-1]) for m in metas)
    padded = torch.zeros(max(max_len, 1), dtype=torch.int64, device=payload.device)
    padded[:payload.shape[0]] = payload
    # one receive buffer, rank r's shard at [r * max_len, (r + 1) * max_len): RCCL writes it in place
    recv = torch.empty(world * padded.shape[0], dtype=torch.int64, device=payload.device)
    bufs = list(recv.chunk(world))
    # the variant is chosen from the backend, up front and identically on every rank: a rank that fell back to another
    # collective after an error would leave the others waiting in the first one
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, padded, group=group)      # the single exchange of the data path
    else:
        dist.all_gather(bufs, padded, group=group)
    per_rank = [unpack(m[:-1], b[:int(m[-1])]) for m, b in zip(metas, bufs)]
    return [concat([per_rank[r][b] for r in range(world)]) for b in range(len(local))]
