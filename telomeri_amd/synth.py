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


def _overlap_records(qs, ql, qid, ts, tl, tid, seed: int, jitter: int, minus_frac: float, *, q_off=None, q_total=None,
                     t_off=None, t_total=None, salt=None) -> Records:
    """Analytic overlap of query interval [qs, qs+ql) with target [ts, ts+tl); jittered ends.
    The hostile generator's intervals are SEGMENTS of longer sequences: `q_off` / `t_off` = where the segment begins inside
    its sequence, `q_total` / `t_total` = the sequence's length (what the PAF's length columns hold); `salt` tells apart
    several records of one (query, target) pair."""
    lo = torch.maximum(qs, ts)
    hi = torch.minimum(qs + ql, ts + tl)
    key = qid.to(torch.int64) * 0x100000001 + tid.to(torch.int64)
    if salt is not None:
        key = key ^ (salt.to(torch.int64) * 0x51ED27)
    h1 = _mix(key, seed * 4 + 1)
    h2 = _mix(key, seed * 4 + 2)
    h3 = _mix(key, seed * 4 + 3)
    q_start = lo - qs
    q_end = hi - qs
    t_start = lo - ts
    t_end = hi - ts
    if q_off is not None:
        q_start, q_end, ql = q_start + q_off, q_end + q_off, q_total
    if t_off is not None:
        t_start, t_end, tl = t_start + t_off, t_end + t_off, t_total
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
# The hostile generator (round 4): what minimap2 `ava-pb` on real reads looks like where the friendly generator above is
# kind (reference run.sh:38-45 feeds the program real overlaps).  Differences, all seeded:
#   * REPEATS: `repeat_frac` of the genome lies in families of `copies_lo..copies_hi` copies of a `repeat_len_lo..hi` bp
#     sequence.  Two reads that cover DIFFERENT copies of a family overlap in the family's coordinates: a partial overlap
#     with long overhangs on both reads, between loci that are megabases apart.  A read inside a 50-copy family is a hub:
#     its list holds thousands of entries whose neighbours lie far outside any id window.
#   * read lengths are log-normal, contig lengths log-uniform over `contig_decades` decades (scaled so that the contigs cover
#     `contig_frac` of the genome), gaps uneven: per-anchor work is heavy-tailed;
#   * `chimera_frac` of the reads are CHIMERIC: two pieces from different loci in one read (each piece overlaps its own
#     locus' reads with the other piece as overhang);
#   * a few contig <-> contig records (a family copy at the end of one contig and at the start of another) and the
#     repeat-induced read -> contig records (nearly all dropped by the overhang filter, as in real data).
# Multi-edges between one pair of reads (a genomic overlap AND a repeat overlap, or two families) are kept, as the
# reference keeps them (OverlapGraph.cpp:233-248).

def _incidences(seg_g, seg_l, order, copy_pos, copy_len, min_ovl: int, max_len: int):
    """(segment, copy) pairs whose genomic intervals share >= min_ovl bp.  `order` sorts the segments by start."""
    s_sorted = seg_g[order]
    lo = torch.searchsorted(s_sorted, copy_pos - max_len, right=False)
    hi = torch.searchsorted(s_sorted, copy_pos + copy_len - min_ovl, right=True)
    cnt = torch.clamp(hi - lo, min=0)
    c = torch.repeat_interleave(torch.arange(copy_pos.shape[0], device=seg_g.device), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(c.shape[0], device=seg_g.device) - torch.repeat_interleave(first, cnt) + torch.repeat_interleave(lo, cnt)
    sg = order[k]
    a = torch.maximum(seg_g[sg], copy_pos[c])
    b = torch.minimum(seg_g[sg] + seg_l[sg], copy_pos[c] + copy_len[c])
    keep = b - a >= min_ovl
    return sg[keep], c[keep], a[keep], b[keep]


def make_hard_dataset(name: str = "hard", *, seed: int = 1, genome_len: int = 600_000, n_contigs: int = 4,
                      contig_frac: float = 0.75, contig_decades: float = 2.0, n_reads: int = 1200,
                      read_len_mean: int = 10_000, read_len_sigma: float = 0.5, read_len_min: int = 1_500,
                      read_len_max: int = 60_000, repeat_frac: float = 0.03, repeat_len_lo: int = 5_000,
                      repeat_len_hi: int = 20_000, copies_lo: int = 10, copies_hi: int = 50, chimera_frac: float = 0.015,
                      n_c2c: int = 6, min_ovl: int = 1_000, jitter: int = 3, minus_frac: float = 0.5,
                      device: str = "cpu") -> Dataset:
    rng = np.random.default_rng(seed)
    G = int(genome_len)
    # --- contigs: log-uniform sizes over `contig_decades` decades, scaled to contig_frac * G; uneven gaps
    raw = 10.0 ** (rng.random(n_contigs) * contig_decades)
    c_len = np.maximum((raw / raw.sum() * contig_frac * G).astype(np.int64), 4 * min_ovl)
    w = 0.2 + 1.6 * rng.random(n_contigs + 1)
    gaps = np.maximum((w / w.sum() * (G - c_len.sum())).astype(np.int64), 2 * min_ovl)
    assert c_len.sum() + gaps.sum() <= G + (n_contigs + 1), "contigs and gaps do not fit the genome"
    c_start = np.cumsum(gaps[:-1]) + np.concatenate([[0], np.cumsum(c_len[:-1])])
    c_start = c_start.astype(np.int64)
    # --- reads: log-normal lengths; a seeded subset is chimeric (two pieces from different loci)
    mu = np.log(read_len_mean) - 0.5 * read_len_sigma ** 2
    r_len = np.clip(np.exp(rng.normal(mu, read_len_sigma, n_reads)), read_len_min, read_len_max).astype(np.int64)
    r_start = (rng.random(n_reads) * (G - r_len)).astype(np.int64)
    chim = np.nonzero((rng.random(n_reads) < chimera_frac) & (r_len >= 2 * (min_ovl + 500)))[0]
    cut = (r_len[chim] * (0.3 + 0.4 * rng.random(chim.shape[0]))).astype(np.int64)
    locus2 = (rng.random(chim.shape[0]) * (G - (r_len[chim] - cut))).astype(np.int64)
    # segments: (read, genomic start, length, offset inside the read); a plain read is one segment
    seg_read = np.concatenate([np.arange(n_reads), chim])
    seg_g = np.concatenate([r_start, locus2])
    seg_l = np.concatenate([r_len, r_len[chim] - cut])
    seg_off = np.concatenate([np.zeros(n_reads, np.int64), cut])
    seg_l[chim] = cut                                   # the first piece of a chimeric read ends at the cut
    # --- repeat families and their copies
    fam_len, fam_copies, total = [], [], 0
    while total < repeat_frac * G:
        L = int(rng.integers(repeat_len_lo, repeat_len_hi + 1))
        C = int(rng.integers(copies_lo, copies_hi + 1))
        fam_len.append(L)
        fam_copies.append(C)
        total += L * C
    n_fam = len(fam_len) if repeat_frac > 0 else 0
    copy_fam = np.repeat(np.arange(n_fam), fam_copies[:n_fam]).astype(np.int64)
    copy_len = np.asarray(fam_len[:n_fam], np.int64)[copy_fam] if n_fam else np.zeros(0, np.int64)
    copy_pos = (rng.random(copy_fam.shape[0]) * (G - copy_len)).astype(np.int64)

    dev = torch.device(device)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int64)).to(dev)
    rs, rl, cs, cl = T(r_start), T(r_len), T(c_start), T(c_len)
    sr, sg, sl, so = T(seg_read), T(seg_g), T(seg_l), T(seg_off)
    st = rl[sr]                                         # the read's whole length (PAF length column)
    cf, cp, cpl = T(copy_fam), T(copy_pos), T(copy_len)
    max_len = int(max(read_len_max, int(c_len.max()))) + 1

    # --- rr, part 1: genomic overlaps of segments of different reads, both directions
    order = torch.argsort(sg, stable=True)
    a, b = _pairs_sorted(sg[order], sg[order] + sl[order], min_ovl)
    xa, xb = order[a], order[b]
    ovl = torch.minimum(sg[xa] + sl[xa], sg[xb] + sl[xb]) - torch.maximum(sg[xa], sg[xb])
    keep = (ovl >= min_ovl) & (sr[xa] != sr[xb])
    xa, xb = xa[keep], xb[keep]
    q, t = torch.cat([xa, xb]), torch.cat([xb, xa])
    rr = _overlap_records(sg[q], sl[q], sr[q], sg[t], sl[t], sr[t], seed, jitter, minus_frac,
                          q_off=so[q], q_total=st[q], t_off=so[t], t_total=st[t])
    # --- rr, part 2: overlaps in a family's coordinates between segments on DIFFERENT copies
    rep_parts = []
    cr_rep = None
    if n_fam:
        isg, icp, ia, ib = _incidences(sg, sl, order, cp, cpl, min_ovl, int(read_len_max) + 1)
        u0, u1 = ia - cp[icp], ib - cp[icp]             # family coordinates of the shared piece
        span = 1 << 26
        key0, key1 = cf[icp] * span + u0, cf[icp] * span + u1
        o2 = torch.argsort(key0, stable=True)
        a, b = _pairs_sorted(key0[o2], key1[o2], min_ovl)
        ya, yb = o2[a], o2[b]
        lo = torch.maximum(u0[ya], u0[yb])
        hi = torch.minimum(u1[ya], u1[yb])
        keep = (hi - lo >= min_ovl) & (cf[icp[ya]] == cf[icp[yb]]) & (icp[ya] != icp[yb]) & (sr[isg[ya]] != sr[isg[yb]])
        ya, yb = ya[keep], yb[keep]
        qi, ti = torch.cat([ya, yb]), torch.cat([yb, ya])
        # in family coordinates both pieces are intervals [u0, u1); inside its read a piece begins at (ia - seg start + offset)
        qo = ia[qi] - sg[isg[qi]] + so[isg[qi]]
        to = ia[ti] - sg[isg[ti]] + so[isg[ti]]
        rep = _overlap_records(u0[qi], u1[qi] - u0[qi], sr[isg[qi]], u0[ti], u1[ti] - u0[ti], sr[isg[ti]], seed + 101, jitter,
                               minus_frac, q_off=qo, q_total=st[isg[qi]], t_off=to, t_total=st[isg[ti]], salt=icp[qi] * 4096 + icp[ti])
        rep_parts.append(rep)
        # repeat-induced read -> contig records: a read on one copy, a contig that holds another copy
        corder = torch.argsort(cs, stable=True)
        jsg, jcp, ja, jb = _incidences(cs, cl, corder, cp, cpl, min_ovl, max_len)
        if jsg.shape[0] and isg.shape[0]:
            # pair every read incidence with the contig incidences of its family (few contigs per family: a dense join per family)
            fam_of_i, fam_of_j = cf[icp], cf[jcp]
            jo = torch.argsort(fam_of_j, stable=True)
            jf = fam_of_j[jo]
            lo_j = torch.searchsorted(jf, fam_of_i, right=False)
            hi_j = torch.searchsorted(jf, fam_of_i, right=True)
            cnt = hi_j - lo_j
            ii = torch.repeat_interleave(torch.arange(isg.shape[0], device=dev), cnt)
            first = torch.cumsum(cnt, 0) - cnt
            jj = jo[torch.arange(ii.shape[0], device=dev) - torch.repeat_interleave(first, cnt) + torch.repeat_interleave(lo_j, cnt)]
            v0, v1 = ja[jj] - cp[jcp[jj]], jb[jj] - cp[jcp[jj]]
            lo = torch.maximum(u0[ii], v0)
            hi = torch.minimum(u1[ii], v1)
            keep = (hi - lo >= min_ovl) & (icp[ii] != jcp[jj])
            ii, jj, v0, v1 = ii[keep], jj[keep], v0[keep], v1[keep]
            qo = ia[ii] - sg[isg[ii]] + so[isg[ii]]
            to = ja[jj] - cs[jsg[jj]]
            cr_rep = _overlap_records(u0[ii], u1[ii] - u0[ii], sr[isg[ii]], v0, v1 - v0, jsg[jj] + n_reads, seed + 7919 + 101,
                                      jitter, minus_frac, q_off=qo, q_total=st[isg[ii]], t_off=to, t_total=cl[jsg[jj]],
                                      salt=icp[ii] * 4096 + jcp[jj])
    for part in rep_parts:
        rr = rr.cat(part)
    # grouped by query in read-id order, then by target; records of one pair keep their order (genomic first)
    k = torch.argsort(rr.q_id.to(torch.int64) * n_reads + rr.t_id.to(torch.int64), stable=True)
    rr = rr.take(k)

    # --- cr: segment(query) -> contig(target), genomic
    cq: List[torch.Tensor] = []
    ct: List[torch.Tensor] = []
    corder = torch.argsort(cs, stable=True)
    jsg, jcp, _, _ = _incidences(sg, sl, order, cs[corder], cl[corder], min_ovl, int(read_len_max) + 1)
    cqi, cti = jsg, corder[jcp]
    cr = _overlap_records(sg[cqi], sl[cqi], sr[cqi], cs[cti], cl[cti], cti + n_reads, seed + 7919, jitter, minus_frac,
                          q_off=so[cqi], q_total=st[cqi])
    if cr_rep is not None:
        cr = cr.cat(cr_rep)
    # --- a few contig <-> contig records: the end of contig X against the start of contig Y (a family copy on both)
    if n_c2c > 0 and n_contigs >= 2:
        x = rng.integers(0, n_contigs, n_c2c)
        y = (x + 1 + rng.integers(0, n_contigs - 1, n_c2c)) % n_contigs
        ell = np.minimum(rng.integers(repeat_len_lo, repeat_len_hi + 1, n_c2c), np.minimum(c_len[x], c_len[y]) - 1)
        i32 = lambda v: torch.from_numpy(np.ascontiguousarray(v, np.int64)).to(dev).to(torch.int32)
        c2c = Records(q_id=i32(x + n_reads), t_id=i32(y + n_reads), q_len=i32(c_len[x]), q_start=i32(c_len[x] - ell),
                      q_end=i32(c_len[x]), t_len=i32(c_len[y]), t_start=i32(np.zeros(n_c2c)), t_end=i32(ell),
                      matches=i32((ell * 0.9).astype(np.int64)), strand=i32(np.zeros(n_c2c)))
        cr = cr.cat(c2c)
    k = torch.argsort(cr.q_id.to(torch.int64) * (n_reads + n_contigs) + cr.t_id.to(torch.int64), stable=True)
    cr = cr.take(k)
    # (read_start / read_len describe the sequence files: a chimeric read is written as one stretch of its first locus --
    # the Scaffolder only cuts substrings, their content does not matter to any comparison)
    rs_seq = torch.minimum(rs, torch.tensor(G, device=dev) - rl)
    return Dataset(name=name, n_reads=n_reads, n_contigs=n_contigs, genome_len=G, read_start=rs_seq, read_len=rl,
                   contig_start=cs, contig_len=cl, rr=rr, cr=cr)



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
    # --- hostile shapes (make_hard_dataset): repeats, skewed contigs, log-normal reads, chimeras, contig<->contig records
    # small enough for the unmodified reference (golden fixture, CPU tier)
    "hard_tiny": dict(kind="hard", seed=21, genome_len=260_000, n_contigs=4, n_reads=330, read_len_mean=8_000,
                      read_len_min=2_500, read_len_max=30_000, repeat_frac=0.08, repeat_len_lo=3_000, repeat_len_hi=6_000,
                      copies_lo=3, copies_hi=5, chimera_frac=0.03, n_c2c=3),
    # the oracle walks it in seconds: hubs of ~10^3 entries (the chunked selections), 8-byte rows, HBM-map probes
    "hard_small": dict(kind="hard", seed=22, genome_len=6_000_000, n_contigs=12, n_reads=50_000, repeat_frac=0.03,
                       copies_lo=6, copies_hi=20, n_c2c=4),
    # C4's size with everything hostile: 120 Mb, 200 contigs, 1 M reads
    "athaliana_hard": dict(kind="hard", seed=24, genome_len=120_000_000, n_contigs=200, n_reads=1_000_000,
                           repeat_frac=0.03, copies_lo=10, copies_hi=50, n_c2c=8),
}


def make_config(name: str, device: str = "cpu", **overrides) -> Dataset:
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    if kw.pop("kind", "friendly") == "hard":
        return make_hard_dataset(name, device=device, **kw)
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
