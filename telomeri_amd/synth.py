"""Seeded synthetic overlap datasets shaped like the reference's inputs.

A linear genome of length G carries disjoint contigs ``ctg<i>`` separated by gaps and reads tiled
over the whole genome (gaps AND contig bodies: a walk that leaves contig B leftwards reaches
contig A only through A's body).  Overlaps are analytic:

* rr: every read pair overlapping by >= ``min_ovl`` bp, both directions (minimap2 ``--dual=yes``,
  reference run.sh:41), grouped by query in read order (read order is random w.r.t. position);
* cr: read(query) -> contig(target) for every read overlapping a contig by >= ``min_ovl``.

Records are the 12 PAF columns the reference parses (OverlapGraph.cpp:109-121) as integer
columns plus name ids; ``write_paf`` renders them as 16-column text (the reference discards
exactly 4 further tokens, OverlapGraph.cpp:124-127).

Everything is integer arithmetic on torch tensors so the same code runs on the CPU (tests) and on
the GPU (bench.py builds the A. thaliana-shaped input in HBM); per-pair "random" attributes come
from a counter hash, not from a stateful generator, so the result does not depend on the device.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional

import numpy as np
import torch

_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _mix(x: torch.Tensor, salt: int) -> torch.Tensor:
    """splitmix64-style finaliser on int64 tensors (wrapping multiply); returns non-negative int64."""
    c1 = -7046029254386353131  # 0x9E3779B97F4A7C15 as signed
    c2 = -4658895280553007687  # 0xBF58476D1CE4E5B9
    c3 = -7723592293110705685  # 0x94D049BB133111EB
    z = x + _s64(salt * c1)
    z = (z ^ ((z >> 30) & 0x3FFFFFFFF)) * c2
    z = (z ^ ((z >> 27) & 0x1FFFFFFFFF)) * c3
    z = z ^ ((z >> 31) & 0x1FFFFFFFF)
    return z & 0x7FFFFFFFFFFFFFFF


@dataclasses.dataclass
class Records:
    """One PAF file as integer columns (all int32 tensors of equal length R)."""
    q_id: torch.Tensor      # name id of the query (index into ``names``)
    t_id: torch.Tensor      # name id of the target
    q_len: torch.Tensor
    q_start: torch.Tensor
    q_end: torch.Tensor
    t_len: torch.Tensor
    t_start: torch.Tensor
    t_end: torch.Tensor
    matches: torch.Tensor
    strand: torch.Tensor    # 1 = '-'

    def __len__(self) -> int:
        return int(self.q_id.shape[0])

    def to(self, device) -> "Records":
        return Records(**{f.name: getattr(self, f.name).to(device) for f in dataclasses.fields(self)})

    def cat(self, other: "Records") -> "Records":
        return Records(**{f.name: torch.cat([getattr(self, f.name), getattr(other, f.name)])
                          for f in dataclasses.fields(self)})

    def take(self, idx: torch.Tensor) -> "Records":
        return Records(**{f.name: getattr(self, f.name)[idx] for f in dataclasses.fields(self)})


@dataclasses.dataclass
class Dataset:
    name: str
    n_reads: int
    n_contigs: int
    genome_len: int
    read_start: torch.Tensor     # int64 [n_reads]
    read_len: torch.Tensor       # int64 [n_reads]
    contig_start: torch.Tensor   # int64 [n_contigs]
    contig_len: torch.Tensor     # int64 [n_contigs]
    rr: Records
    cr: Records

    # name ids: reads are 0..n_reads-1, contigs n_reads..n_reads+n_contigs-1
    def name_of(self, name_id: int) -> str:
        if name_id < self.n_reads:
            return "read%07d" % (name_id + 1)
        return "ctg%d" % (name_id - self.n_reads + 1)

    def name_is_ctg(self) -> torch.Tensor:
        """bool per name id: does the name start with 'ctg' (Utils.hpp:17-27)?"""
        m = torch.zeros(self.n_reads + self.n_contigs, dtype=torch.bool)
        m[self.n_reads:] = True
        return m


def _pairs_sorted(start: torch.Tensor, end: torch.Tensor, min_ovl: int):
    """All (a, b), a < b in start order, with start[b] <= end[a] - min_ovl. start must be sorted."""
    n = start.shape[0]
    hi = torch.searchsorted(start, end - min_ovl, right=True)
    idx = torch.arange(n, device=start.device)
    cnt = torch.clamp(hi - (idx + 1), min=0)
    a = torch.repeat_interleave(idx, cnt)
    first = torch.cumsum(cnt, 0) - cnt
    b = torch.arange(a.shape[0], device=start.device) - torch.repeat_interleave(first, cnt) + a + 1
    return a, b


def _overlap_records(qs, ql, qid, ts, tl, tid, seed: int, jitter: int, minus_frac: float) -> Records:
    """Analytic overlap of query interval [qs, qs+ql) with target [ts, ts+tl); jittered ends."""
    lo = torch.maximum(qs, ts)
    hi = torch.minimum(qs + ql, ts + tl)
    key = qid.to(torch.int64) * 0x100000001 + tid.to(torch.int64)
    h1 = _mix(key, seed * 4 + 1)
    h2 = _mix(key, seed * 4 + 2)
    h3 = _mix(key, seed * 4 + 3)
    q_start = lo - qs
    q_end = hi - qs
    t_start = lo - ts
    t_end = hi - ts
    if jitter > 0:
        # alignment ends wobble by a few bp, independently on the two sequences (clamped to the reads)
        j = 2 * jitter + 1
        q_start = torch.clamp(q_start + (h1 % j) - jitter, min=0)
        t_start = torch.clamp(t_start + ((h1 >> 8) % j) - jitter, min=0)
        q_end = torch.minimum(q_end + ((h1 >> 16) % j) - jitter, ql)
        t_end = torch.minimum(t_end + ((h1 >> 24) % j) - jitter, tl)
        q_end = torch.maximum(q_end, q_start + 1)
        t_end = torch.maximum(t_end, t_start + 1)
    ovl = torch.minimum(q_end - q_start, t_end - t_start)
    # matches = floor(ovl * U(0.80, 0.95)) with U on a 2^20 grid
    frac_num = 838861 + (h2 % 157286)          # 0.80*2^20 .. 0.95*2^20
    matches = (ovl * frac_num) >> 20
    thr = int(minus_frac * (1 << 20))
    strand = ((h3 % (1 << 20)) < thr)
    i32 = lambda t: t.to(torch.int32)
    return Records(q_id=i32(qid), t_id=i32(tid), q_len=i32(ql), q_start=i32(q_start), q_end=i32(q_end),
                   t_len=i32(tl), t_start=i32(t_start), t_end=i32(t_end), matches=i32(matches),
                   strand=i32(strand))


def make_dataset(name: str = "toy", *, seed: int = 1, genome_len: int = 600_000, n_contigs: int = 3,
                 contig_len: int = 100_000, n_reads: int = 1200, read_len_lo: int = 5_000,
                 read_len_hi: int = 15_000, min_ovl: int = 1_000, jitter: int = 3,
                 minus_frac: float = 0.5, device: str = "cpu") -> Dataset:
    """Contigs are laid out evenly: gap, contig, gap, contig, ..., gap."""
    rng = np.random.default_rng(seed)
    gap = (genome_len - n_contigs * contig_len) // (n_contigs + 1)
    assert gap >= 0, "contigs do not fit the genome"
    c_start = np.array([gap + i * (contig_len + gap) for i in range(n_contigs)], dtype=np.int64)
    c_len = np.full(n_contigs, contig_len, dtype=np.int64)
    r_len = rng.integers(read_len_lo, read_len_hi + 1, size=n_reads, dtype=np.int64)
    r_start = (rng.random(n_reads) * (genome_len - r_len)).astype(np.int64)

    dev = torch.device(device)
    rs = torch.from_numpy(r_start).to(dev)
    rl = torch.from_numpy(r_len).to(dev)
    cs = torch.from_numpy(c_start).to(dev)
    cl = torch.from_numpy(c_len).to(dev)

    # --- read/read, both directions, grouped by query in read-id order
    order = torch.argsort(rs, stable=True)
    s_sorted = rs[order]
    e_sorted = s_sorted + rl[order]
    a, b = _pairs_sorted(s_sorted, e_sorted, min_ovl)
    # keep only pairs that really overlap by min_ovl (b may end before a does)
    ra, rb = order[a], order[b]
    ovl = torch.minimum(rs[ra] + rl[ra], rs[rb] + rl[rb]) - torch.maximum(rs[ra], rs[rb])
    keep = ovl >= min_ovl
    ra, rb = ra[keep], rb[keep]
    q = torch.cat([ra, rb])
    t = torch.cat([rb, ra])
    # group by query (read order = file order of a reads file, random w.r.t. position), then by target
    k = torch.argsort(q * n_reads + t, stable=True)
    q, t = q[k], t[k]
    rr = _overlap_records(rs[q], rl[q], q, rs[t], rl[t], t, seed, jitter, minus_frac)

    # --- contig/read: read is the query, contig the target (reference default, OverlapGraph.cpp:133)
    cq: List[torch.Tensor] = []
    ct: List[torch.Tensor] = []
    for ci in range(n_contigs):
        lo = torch.maximum(rs, cs[ci])
        hi = torch.minimum(rs + rl, cs[ci] + cl[ci])
        idx = torch.nonzero(hi - lo >= min_ovl).flatten()
        cq.append(idx)
        ct.append(torch.full_like(idx, ci))
    cqi = torch.cat(cq)
    cti = torch.cat(ct)
    k = torch.argsort(cqi * n_contigs + cti, stable=True)
    cqi, cti = cqi[k], cti[k]
    cr = _overlap_records(rs[cqi], rl[cqi], cqi, cs[cti], cl[cti], cti + n_reads, seed + 7919, jitter, minus_frac)
    return Dataset(name=name, n_reads=n_reads, n_contigs=n_contigs, genome_len=genome_len,
                   read_start=rs, read_len=rl, contig_start=cs, contig_len=cl, rr=rr, cr=cr)


# ---------------------------------------------------------------------------------------------
# Named configs.  Sizes of C2..C5 are what the generator produced, see DESIGN.md.
CONFIGS: Dict[str, dict] = {
    # C1 plumbing toy: CPU-runnable by the unmodified reference in ~3 s
    "toy": dict(seed=1, genome_len=600_000, n_contigs=3, contig_len=100_000, n_reads=1200,
                read_len_lo=5_000, read_len_hi=15_000),
    # smaller still, for golden fixtures that are committed as text
    "tiny": dict(seed=2, genome_len=200_000, n_contigs=3, contig_len=30_000, n_reads=260,
                 read_len_lo=4_000, read_len_hi=12_000),
    # low coverage: dead ends, backtracking, abandoned walks
    "sparse": dict(seed=3, genome_len=300_000, n_contigs=3, contig_len=40_000, n_reads=210,
                   read_len_lo=4_000, read_len_hi=12_000),
    # ~50x coverage: neighbour lists of several 64-entry chunks
    "dense": dict(seed=4, genome_len=120_000, n_contigs=2, contig_len=30_000, n_reads=700,
                  read_len_lo=5_000, read_len_hi=12_000),
    # C2 E. coli-like: 4.6 Mb, 3 contigs, 40 k reads mean 8 kb
    "ecoli": dict(seed=12, genome_len=4_600_000, n_contigs=3, contig_len=1_400_000, n_reads=40_000,
                  read_len_lo=3_000, read_len_hi=13_000),
    # C3 yeast-like: 12 Mb, 30 contigs, 200 k reads
    "yeast": dict(seed=13, genome_len=12_000_000, n_contigs=30, contig_len=300_000, n_reads=200_000,
                  read_len_lo=3_000, read_len_hi=13_000),
    # C4 A. thaliana-like: 120 Mb, 200 contigs, 1 M reads mean 10 kb
    "athaliana": dict(seed=14, genome_len=120_000_000, n_contigs=200, contig_len=450_000,
                      n_reads=1_000_000, read_len_lo=5_000, read_len_hi=15_000),
    # C5 human chr1-like: 248 Mb, 500 contigs, 2 M reads mean 20 kb
    "chr1": dict(seed=15, genome_len=248_000_000, n_contigs=500, contig_len=400_000,
                 n_reads=2_000_000, read_len_lo=10_000, read_len_hi=30_000),
}


def make_config(name: str, device: str = "cpu", **overrides) -> Dataset:
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    return make_dataset(name, device=device, **kw)


def write_paf(ds: Dataset, rec: Records, path: str, extra_lines: Optional[List[str]] = None) -> None:
    """16-column PAF text (12 parsed + 4 discarded tokens, OverlapGraph.cpp:109-127)."""
    cols = {f.name: getattr(rec, f.name).cpu().numpy() for f in dataclasses.fields(rec)}
    with open(path, "w") as fh:
        for i in range(len(rec)):
            ovl = int(cols["q_end"][i] - cols["q_start"][i])
            fh.write("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\ttp:A:S\tcm:i:%d\ts1:i:%d\tdv:f:0.0100\n" % (
                ds.name_of(int(cols["q_id"][i])), cols["q_len"][i], cols["q_start"][i], cols["q_end"][i],
                "-" if cols["strand"][i] else "+", ds.name_of(int(cols["t_id"][i])), cols["t_len"][i],
                cols["t_start"][i], cols["t_end"][i], cols["matches"][i], ovl, 60, ovl // 4, cols["matches"][i]))
        for ln in extra_lines or []:
            fh.write(ln.rstrip("\n") + "\n")


def write_fasta(ds: Dataset, reads_path: str, contigs_path: str, seed: int = 0) -> None:
    """Two-line FASTA records (what the reference's Scaffolder reads, Scaffolder.cpp:12-41): a random genome, reads and
    contigs cut out of it on the forward strand."""
    rng = np.random.default_rng(seed + 977)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=ds.genome_len)]
    rs, rl = ds.read_start.cpu().numpy(), ds.read_len.cpu().numpy()
    with open(reads_path, "w") as fh:
        for i in range(ds.n_reads):
            fh.write(">%s\n%s\n" % (ds.name_of(i), genome[rs[i]:rs[i] + rl[i]].tobytes().decode()))
    cs, cl = ds.contig_start.cpu().numpy(), ds.contig_len.cpu().numpy()
    with open(contigs_path, "w") as fh:
        for i in range(ds.n_contigs):
            fh.write(">%s\n%s\n" % (ds.name_of(ds.n_reads + i), genome[cs[i]:cs[i] + cl[i]].tobytes().decode()))
