"""The product's multi-rank exchange (csrc/exchange.hip) EXECUTED with more than one rank on one GPU.

Several processes (tests/exchange_worker.py) share cuda:0; each builds the same graph, walks its anchor bundle and joins
tlm_pathset_allgather and tlm_pathset_gather_summaries / tlm_gathered_fetch through the library's own communicator.  RCCL
refuses two ranks on one device, so the ten RCCL entry points the library binds are provided by tests/fake_rccl (test
infrastructure, selected with TLM_RCCL_LIB): the library's code -- block placement, grouped send / receive pairing with
its skipping of empty blocks, 16-bit slots and the fall-back to 32 bits, the owners' broadcasts, the fail-together
protocol -- is what runs.  The result of every rank is compared with the one-process result byte for byte
(reference consumer: Main.cpp:237-252, paths_ after the four builder calls).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import exchange_worker as W
from telomeri_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
# the fault-injection switch (TLM_EXCHANGE_FAULT) exists in the CHECKED build of the library only (make EXTRA=-DTLM_CHECKED)
CHECKED_DIR = os.path.join(ROOT, "telomeri_amd", "lib_checked")
CHECKED = os.path.join(CHECKED_DIR, "libtelomeri_hip.so")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fake_rccl():
    if not os.path.exists(FAKE):
        subprocess.run(["make", "-s", "-C", os.path.dirname(FAKE)], check=True)
    return FAKE


def _async_env(on: bool) -> dict:
    return {"FAKE_RCCL_ASYNC": "1" if on else "0"}


def run_ranks(fake, dataset, world, tmp, extra_env=None, timeout=420):
    env = dict(os.environ, TLM_RCCL_LIB=fake, FAKE_RCCL_TIMEOUT_S="90", **(extra_env or {}))
    env.pop("TLM_EXCHANGE_SELFTEST", None)
    idfile = os.path.join(tmp, "id")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "exchange_worker.py"), dataset, str(r), str(world),
                               idfile, tmp], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)         # a rank that hangs fails the test here
            outs.append((p.returncode, out))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


@pytest.fixture(scope="module")
def ctx():
    return capi.Context(0)


def one_process(ctx, dataset):
    dg, wp = W.make_graph(ctx, dataset)
    sets = W.rank_sets(dg, wp, 0, 1)
    want = capi.PathSet.concat(sets)
    return dg, wp, sets, want


# "async": the stand-in returns from every call at once and completes the operation later, in stream order, behind a delay
# (FAKE_RCCL_ASYNC, tests/fake_rccl/fake_rccl.cpp): a missing stream dependency between a local kernel and a transfer, a
# host read of a gathered size before its stream has drained, a buffer released early -- all of which the synchronous mode
# executes "correctly" -- show here as wrong results on one GPU
CASES = [(d, w, p, "sync") for p in ("sliced", "contiguous") for d, w in (("toy", 2), ("quirks", 2), ("toy", 4), ("quirks", 3), ("hub", 2))
         if (d, w, p) != ("toy", 4, "contiguous")]     # (25 s; four ranks run sliced, three contiguous)
CASES += [("toy", 2, "sliced", "async"), ("quirks", 3, "sliced", "async"), ("hub", 2, "contiguous", "async")]


@pytest.mark.parametrize("dataset,world,partition,mode", CASES)
def test_exchange_with_several_ranks(ctx, fake_rccl, dataset, world, partition, mode):
    dg, wp, sets, want = one_process(ctx, dataset)
    h = want.to_host()
    front, back, count, hsh = want.summary()
    off = h.off.astype(np.int64)
    n = len(h)
    assert n > 0
    with tempfile.TemporaryDirectory() as tmp:
        outs = run_ranks(fake_rccl, dataset, world, tmp, dict(_async_env(mode == "async"), TLM_TEST_PARTITION=partition))
        for r, (rc, out) in enumerate(outs):
            err = os.path.join(tmp, "rank%d.err" % r)
            assert rc == 0, "rank %d: %s\n%s" % (r, open(err).read() if os.path.exists(err) else "", out[-2000:])
        res = [dict(np.load(os.path.join(tmp, "rank%d.npz" % r))) for r in range(world)]
    # which rank owns which path: MC unit -> anchor -> bundle; DET unit -> anchor through det_off
    _, det_off = dg.anchor_table()
    builder = (h.unit >> np.uint64(56)).astype(np.int64)
    unit = (h.unit & np.uint64((1 << 56) - 1)).astype(np.int64)
    anchor = np.where(builder < 2, unit // wp.rebuild_attempts, np.searchsorted(det_off.astype(np.int64), unit, side="right") - 1)
    if partition == "contiguous":
        ranges = [dg.anchor_range(r, world) for r in range(world)]
        owner = np.array([next(r for r, (a0, a1) in enumerate(ranges) if a0 <= a < a1) for a in anchor], np.uint32)
    else:       # sliced: the rank whose share holds the unit (tlm_dist_unit_share)
        owner = np.full(n, 0xFFFFFFFF, np.uint32)
        for r in range(world):
            for det in (False, True):
                mine = np.isin(unit, dg.unit_share(r, world, wp, det).astype(np.int64)) & ((builder >= 2) == det)
                assert np.all(owner[mine] == 0xFFFFFFFF)
                owner[mine] = r
        assert np.all(owner != 0xFFFFFFFF)
        # the shares are disjoint and cover every unit
        for det, total in ((False, dg.mc_units(wp)), (True, dg.det_units())):
            allu = np.concatenate([dg.unit_share(r, world, wp, det) for r in range(world)])
            assert allu.shape[0] == total and np.unique(allu).shape[0] == total
    seqs = [h.nodes[off[i]:off[i + 1]].tobytes() for i in range(n)]
    first = np.r_[True, (builder[1:] != builder[:-1]) | (owner[1:] != owner[:-1])]
    dup = np.array([(not first[i]) and seqs[i] == seqs[i - 1] for i in range(n)])
    pick = W.pick_indices(n).astype(np.int64)
    total_local = np.zeros(4, np.int64)
    for r, e in enumerate(res):
        # ---- the full exchange: paths_ of one PathManager, element for element
        np.testing.assert_array_equal(e["full_unit"], h.unit, err_msg="rank %d" % r)
        np.testing.assert_array_equal(e["full_length"], h.length)
        np.testing.assert_array_equal(e["full_off"], h.off)
        np.testing.assert_array_equal(e["full_nodes"], h.nodes)
        np.testing.assert_array_equal(e["full_slots"], h.slots)
        # ---- the lazy exchange
        assert int(e["lazy_info"][2]) == n and int(e["lazy_info"][3]) == int(e["local_paths"].sum())
        for k, w in (("unit", h.unit), ("length", h.length), ("front", front), ("back", back), ("n_nodes", count), ("hash", hsh),
                     ("owner", owner)):
            np.testing.assert_array_equal(e["lazy_" + k], w, err_msg="rank %d %s" % (r, k))
        np.testing.assert_array_equal(e["lazy_block_first"].astype(bool), first)
        np.testing.assert_array_equal(e["lazy_dup_of_prev"].astype(bool), dup)
        np.testing.assert_array_equal(e["fetch_nodes"], np.concatenate([h.nodes[off[i]:off[i + 1]] for i in pick]))
        np.testing.assert_array_equal(e["fetch_slots"], np.concatenate([h.slots[off[i]:off[i + 1]] for i in pick]))
        total_local += e["local_paths"]
        # bytes on the wire: everything but this rank's own blocks, 24 B a path + (4 + slot bytes) a path node
        mine = owner == r
        slot_bits = int(e["full_info"][2])
        other_nodes = int((count[~mine]).sum())
        assert int(e["full_info"][1]) == 24 * int((~mine).sum()) + (4 + slot_bits // 8) * other_nodes
        assert int(e["full_info"][0]) == (world - 1) * (24 * int(mine.sum()) + (4 + slot_bits // 8) * int(count[mine].sum()))
    np.testing.assert_array_equal(total_local, [s.n_paths for s in sets])
    wide = bool(((h.slots != 0xFFFFFFFF) & (h.slots >= 0xFFFF)).any())
    assert all(int(e["full_info"][2]) == (32 if wide else 16) for e in res)
    if dataset == "hub":
        assert wide, "the hub graph is there to force 32-bit slots"
        if partition == "contiguous":       # only the first rank's paths hold such a slot: the others learn it from the sizes exchange
            assert not ((h.slots != 0xFFFFFFFF) & (h.slots >= 0xFFFF))[np.repeat(owner != 0, count)].any()
    if dataset == "quirks":
        assert dup.any(), "quirks is there for its runs of identical paths"
    if (dataset, world, partition) == ("toy", 4, "contiguous"):
        assert any(e["local_paths"].sum() == 0 for e in res), "4 ranks over 3 anchors: one rank has nothing to send"
    if partition == "sliced" and world > 1:
        assert int(first.sum()) > 4 * world or n < 8 * world, "sliced shares interleave: many blocks"


@pytest.mark.parametrize("point", ["prep", "alloc", "group", "fetch"])
def test_ranks_fail_together(fake_rccl, point):
    """One rank pretends a local step failed (TLM_EXCHANGE_FAULT): the call ends on EVERY rank -- the failing one with
    its own error, the other with TLM_EPEER (-5) -- nobody hangs; after a failure inside the grouped transfers the
    communicator is aborted and later calls fail at once."""
    with tempfile.TemporaryDirectory() as tmp:
        # (rank 0 is the culprit: in the toy set the last anchor, rank 1's only one, has no accepted path -- nothing to send)
        outs = run_ranks(fake_rccl, "toy", 2, tmp, {"TLM_EXCHANGE_FAULT": "0:" + point, "TLM_LIB": CHECKED, "TLM_TEST_PARTITION": "contiguous"},
                         timeout=300)
        logs = []
        for r, (rc, out) in enumerate(outs):
            assert rc == 0, "rank %d: %s" % (r, out[-2000:])
            logs.append(open(os.path.join(tmp, "rank%d.log" % r)).read().splitlines())
    culprit, peer = logs[0], logs[1]
    if point == "fetch":
        assert culprit[0] == peer[0] == "allgather: ok" and culprit[1] == peer[1] == "gather_summaries: ok"
        for k in (2, 3):
            assert "injected failure at 'fetch' on rank 0" in culprit[k] and "(-2)" in culprit[k]
            assert "(-5)" in peer[k] and "rank 0 failed" in peer[k]
        return
    assert "injected failure at '%s' on rank 0" % point in culprit[0] and "(-2)" in culprit[0]
    assert "allgather" in peer[0] and "ok" not in peer[0]
    if point == "group":
        # the culprit left in the middle of the grouped transfers: both communicators are gone, the next call says so
        assert "aborted" in culprit[0]
        assert "(-5)" in culprit[1] and "aborted" in culprit[1]
        assert "(-5)" in peer[1] and "aborted" in peer[1]
    else:
        assert "(-5)" in peer[0] and "rank 0 failed" in peer[0]
        assert "injected failure" in culprit[1] and "(-5)" in peer[1]       # the communicator is intact: the protocol again
    assert len(culprit) == len(peer) == 2


# ---- the host program, two ranks (run.sh with TLM_GPUS=2; both on device 0) -------------------------------------------------
REPORT = ("-", "Paths", "Found", "Consensus", "Final", "ctg", "read", "Anchor pairs")


def _dataset_dir(tmp, name):
    import golden_util as G
    import helpers as H
    from telomeri_amd import synth
    meta = G.load_json(name + ".json")
    ds = H.dataset_by_name(name)
    d = os.path.join(tmp, name)
    os.makedirs(d)
    os.rename(G.gunzip_to(tmp, name + "_rr.paf.gz"), os.path.join(d, "rr_overlaps.paf"))
    os.rename(G.gunzip_to(tmp, name + "_cr.paf.gz"), os.path.join(d, "cr_overlaps.paf"))
    synth.write_fasta(ds, os.path.join(d, "reads.fasta"), os.path.join(d, "contigs.fasta"))
    return d, [a for a in meta["args"] if a != "--sqrt"]


def _run_sh(d, args, env):
    res = subprocess.run([os.path.join(ROOT, "run.sh"), d] + args, capture_output=True, text=True, timeout=420,
                         env=dict(os.environ, **env))
    out = os.path.join(d, "HERA_result.fasta")
    body = open(out, "rb").read() if os.path.exists(out) else None
    if body is not None:
        os.remove(out)
    return res, body


@pytest.mark.parametrize("name,mode,collectives", [("tiny", "lazy", "sync"), ("tiny", "full", "sync"), ("quirks", "lazy", "sync"),
                                                   ("quirks", "full", "sync"), ("tiny", "lazy", "async"), ("quirks", "full", "async")])
def test_host_program_with_two_ranks(fake_rccl, name, mode, collectives):
    """telomeri_hip as a two-rank job (TLM_WORLD=2: every rank loads and builds, walks its bundle, joins the exchange, runs
    the host stages; rank 0 writes the scaffold): same report, same scaffold bytes as the one-process run."""
    with tempfile.TemporaryDirectory() as tmp:
        d, args = _dataset_dir(tmp, name)
        one, body_one = _run_sh(d, args, {})
        assert one.returncode == 0, one.stderr[-2000:]
        two, body_two = _run_sh(d, args, dict(_async_env(collectives == "async"), TLM_GPUS="2", TLM_DEVICES="0 0", TLM_RCCL_LIB=fake_rccl,
                                              TLM_EXCHANGE=mode, FAKE_RCCL_TIMEOUT_S="90"))
        rank1 = open(os.path.join(d, "telomeri_rank1.log")).read()
        assert two.returncode == 0, two.stderr[-2000:] + rank1[-2000:]
        assert "Gathered" in two.stdout and "from 2 ranks" in two.stdout and "from 2 ranks" in rank1
        # "Found N paths." is what a builder call found among THIS rank's units: the ranks' counts add up to the one-process
        # figure, builder by builder; everything after the exchange is the same report on every rank
        found = lambda txt: [int(ln.split()[1]) for ln in txt.splitlines() if ln.startswith("Found ") and ln.endswith(" paths.")]
        keep = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(REPORT) and not ln.startswith("Found ")]
        assert len(found(one.stdout)) == 4
        assert [a + b for a, b in zip(found(two.stdout), found(rank1))] == found(one.stdout)
        assert keep(two.stdout) == keep(one.stdout)
        assert keep(rank1) == keep(one.stdout)          # every rank runs the same host stages on the same paths_
        assert body_two == body_one and body_one


def test_host_program_ranks_end_together(fake_rccl):
    """A rank that fails takes the job down: run.sh returns its status and leaves no rank behind waiting in a collective."""
    import time
    with tempfile.TemporaryDirectory() as tmp:
        d, args = _dataset_dir(tmp, "tiny")
        t0 = time.time()
        # (the host program finds the library next to itself: LD_LIBRARY_PATH puts the checked build in front of that)
        res, body = _run_sh(d, args, {"TLM_GPUS": "2", "TLM_DEVICES": "0 0", "TLM_RCCL_LIB": fake_rccl, "TLM_EXCHANGE_FAULT": "1:prep",
                                      "FAKE_RCCL_TIMEOUT_S": "90",
                                      "LD_LIBRARY_PATH": CHECKED_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", "")})
        assert res.returncode != 0 and body is None
        assert time.time() - t0 < 120
        rank1 = open(os.path.join(d, "telomeri_rank1.log")).read()
        assert "injected failure" in rank1
        assert "rank 0 failed" in res.stderr or "a rank ended with status" in res.stderr


# ---- bench.py as a two-rank job on one GPU (gloo for the harness's own small collectives, the library's exchange through
# ---- the stand-in): the N > 1 line is a measured line, and a rank that loses its communicator ends the job -----------------
def _bench(fake, extra_env, timeout=420, extra_args=(), scaffold=False):
    import json
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "toy", "--gpus", "2", "--share-gpu", "--dist-backend", "gloo",
           "--steps", "2", "--warmup", "1", "--rb-att", "40", "--cpu-seconds", "1"] + ([] if scaffold else ["--no-scaffold"]) + list(extra_args)
    env = dict(os.environ, TLM_RCCL_LIB=fake, FAKE_RCCL_TIMEOUT_S="60", **extra_env)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None), time.time() - t0


@pytest.mark.parametrize("partition,collectives", [("sliced", "sync"), ("contiguous", "sync"), ("sliced", "async")])
def test_bench_two_ranks_line_is_a_measured_line(fake_rccl, partition, collectives):
    res, line, _ = _bench(fake_rccl, _async_env(collectives == "async"), extra_args=["--partition", partition])
    assert res.returncode == 0, res.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert partition in line["config"]["partition"]
    assert "tlm_pathset_gather_summaries" in line["exchange"]["transport"]
    # rank 0 carries the CPU baseline at N > 1 too (the same bounded sample), with the reference / port ratio beside it
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["sample_matches_device"] is True
    assert 0.3 < cb["reference_over_port_1thread"] < 0.7
    # the dominant kernel of EVERY rank: min / max over the ranks
    rk = line["roofline"]["ranks"]
    assert 0 < rk["avg_launch_ms_min"] <= line["roofline"]["avg_launch_ms"] <= rk["avg_launch_ms_max"]
    assert 0 < rk["frac_min"] <= rk["frac_max"]
    if partition == "sliced":       # every rank walks half of every anchor's trials: equal unit counts, near-equal work
        assert rk["mc_evals_max"] / rk["mc_evals_min"] < 1.6


def test_bench_two_ranks_scaffold_leg(fake_rccl):
    """The second half of the metric at N > 1: every bench rank starts ITS rank of the host program on the text rank 0 has
    rendered (sharded load, sliced walks, lazy exchange, rank 0 writes the scaffold); the line carries the slowest rank's
    wall-clock.  Here two ranks on one GPU through the stand-in -- the code path the driver's SCALE run takes."""
    res, line, _ = _bench(fake_rccl, {}, scaffold=True)
    assert res.returncode == 0, res.stderr[-3000:]
    sc = line["scaffold"]
    assert sc["ranks"] == 2 and sc["load"] == "sharded", sc
    assert sc["all_ranks_ok"] is True and sc["exit"] == 0, sc
    assert line["scaffold_wall_s"] > 0 and sc["scaffold_bytes"] > 0 and sc["paths"] > 0
    assert any("sharded load" in ln for ln in sc["rank_timing"]), sc["rank_timing"]


def test_bench_rank_that_loses_its_communicator_ends_the_job(fake_rccl):
    """a failure in the middle of the grouped transfers aborts the communicator; the failing rank must LEAVE (the
    launcher ends the others) instead of waiting for its peers in the harness's gather until a collective timeout."""
    res, line, seconds = _bench(fake_rccl, {"TLM_EXCHANGE_FAULT": "1:group", "TLM_EXCHANGE": "full", "TLM_LIB": CHECKED}, timeout=300)
    assert res.returncode != 0, res.stdout[-500:]
    assert line is None
    assert seconds < 150, "the job took %.0f s to end" % seconds
    assert "the communicator is gone" in res.stderr or "ended with status" in res.stderr


# ---- the sharded load (ABI 5) ---------------------------------------------------------------------
# OverlapGraph::load reads every overlap file whole (OverlapGraph.cpp:94-131); in a job of N ranks every rank brings only its
# run of pieces of each file to its device and the ranks swap the runs (csrc/exchange.hip: tlm_text_share).  From the full
# text on, every rank runs the one-process code, so everything must EQUAL the one-process result: record counts, columns,
# names in first-seen order, the graph bit for bit.

def _load_files(tmp, name):
    import golden_util as G
    return G.gunzip_to(tmp, name + "_rr.paf.gz"), G.gunzip_to(tmp, name + "_cr.paf.gz")


@pytest.mark.parametrize("name,world,piece_log2,collectives", [("tiny", 2, 16, "sync"), ("hard_tiny", 3, 16, "sync"), ("quirks", 4, 14, "sync"),
                                                               ("hard_tiny", 2, 17, "async"), ("tiny", 2, 30, "sync")])
def test_sharded_load_equals_one_process(ctx, fake_rccl, name, world, piece_log2, collectives):
    """Pieces of 16-128 KB (checked build's TLM_TEXT_PIECE_LOG2) make a text of a few hundred KB a matter of many pieces: every
    rank uploads a run of them -- runs of different lengths, a last piece that is short, a cr file of ONE piece that only rank 0
    brings -- and receives the others' runs in place.  piece_log2 = 30 is the product's piece: rank 0 brings everything."""
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = _load_files(tmp, name)
        want = W.load_graph(ctx, [(rr, False), (cr, True)])
        assert want["share"].tolist() == [0, 0]
        outs = run_ranks(fake_rccl, "load:%s:%s" % (rr, cr), world, tmp,
                         dict(_async_env(collectives == "async"), TLM_LIB=CHECKED, TLM_TEXT_PIECE_LOG2=str(piece_log2)))
        for r, (rc, out) in enumerate(outs):
            err = os.path.join(tmp, "rank%d.err" % r)
            assert rc == 0, out[-3000:] + (open(err).read() if os.path.exists(err) else "")
        total = os.path.getsize(rr) + os.path.getsize(cr)
        received = 0
        for r in range(world):
            got = np.load(os.path.join(tmp, "rank%d.npz" % r))
            assert set(got.files) == set(want.keys())
            for k, v in want.items():
                if k != "share":
                    np.testing.assert_array_equal(got[k], v, err_msg="rank %d: %s" % (r, k))
            sent, recv = got["share"].tolist()
            # a rank receives what it did not bring itself; it sends what it brought to each of the others
            piece = 1 << piece_log2
            own = 0
            for path in (rr, cr):
                b = os.path.getsize(path)
                n = (b + piece - 1) // piece
                own += min(b, (n * (r + 1) // world) * piece) - min(b, (n * r // world) * piece)
            assert recv == total - own and sent == own * (world - 1), (r, sent, recv, own, total)
            received += recv
        assert received == (world - 1) * total


def test_sharded_load_fails_together(ctx, fake_rccl):
    """A rank that cannot bring its pieces (injected: TLM_EXCHANGE_FAULT=<rank>:text, checked build) takes every rank out of
    the load with an error; none is left waiting in the swap."""
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = _load_files(tmp, "tiny")
        outs = run_ranks(fake_rccl, "load:%s:%s" % (rr, cr), 3, tmp, dict(TLM_LIB=CHECKED, TLM_TEXT_PIECE_LOG2="16", TLM_EXCHANGE_FAULT="1:text"),
                         timeout=200)
        errs = [open(os.path.join(tmp, "rank%d.err" % r)).read() for r in range(3)]
        assert [rc for rc, _ in outs] == [3, 3, 3], outs
        assert "injected failure at 'text' on rank 1" in errs[1]
        assert all("rank 1 failed in the sharded load" in errs[r] for r in (0, 2)), errs


def test_sharded_load_refuses_ranks_that_see_different_files(ctx, fake_rccl):
    """The ranks swap the size of the text before any piece moves: a rank that sees another file (here: a shorter one) makes
    EVERY rank return an error -- sends and receives of different lengths would never meet."""
    with tempfile.TemporaryDirectory() as tmp:
        rr, cr = _load_files(tmp, "tiny")
        with open(rr, "rb") as f, open(rr + ".other", "wb") as o:
            o.write(f.read()[:100_000])
        outs = run_ranks(fake_rccl, "load:%s:%s" % (rr, cr), 2, tmp, dict(TLM_LIB=CHECKED, TLM_TEXT_PIECE_LOG2="16", TLM_TEST_OTHER_FILE_RANK="1"),
                         timeout=200)
        errs = [open(os.path.join(tmp, "rank%d.err" % r)).read() for r in range(2)]
        assert [rc for rc, _ in outs] == [3, 3], outs
        assert all("not the same file" in e for e in errs), errs


def test_host_program_sharded_load_of_several_pieces(fake_rccl):
    """telomeri_hip as a two-rank job on a text of seven 1 GiB pieces: rank 0's helper process brings pieces 0-2, rank 1's
    pieces 3-6 (protocol line "R") -- four pieces through the three slots of its window, so piece 6 waits for the word that
    piece 3, the FIRST of its run, has been taken over --, the ranks swap their runs (tlm_paf_text_share), and both load what
    one process loads."""
    import re
    import golden_util as G
    binary = os.path.join(ROOT, "telomeri_amd", "bin", "telomeri_hip")
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        rr, cr = os.path.join(tmp, "rr.paf"), G.gunzip_to(tmp, "tiny_cr.paf.gz")
        block = "".join("r%d\t9000\t10\t8000\t+\tr%d\t9500\t20\t8100\t7000\t8100\t60\ta\tb\tc\td\n" % (k, k + 1) for k in range(4096)).encode()
        reps = (6400 << 20) // len(block) + 1
        with open(rr, "wb") as fh:
            for _ in range(reps):
                fh.write(block)
        assert 6 * (1 << 30) < os.path.getsize(rr) < 7 * (1 << 30)
        common = dict(os.environ, TLM_HOST_TIMING="1")
        one = subprocess.run([binary, "--no-scaffold", "--rb-att", "2", rr, cr, "r", "c", "o"], capture_output=True, text=True, timeout=900, env=common)
        assert one.returncode == 0 and "(device)" in one.stdout, one.stderr[-2000:]
        rv = os.path.join(tmp, "rendezvous")
        # (outputs into files: a rank whose pipe is full while nobody reads it would stop, and its peer with it)
        logs = [(open(os.path.join(tmp, "rank%d.out" % r), "w+"), open(os.path.join(tmp, "rank%d.err" % r), "w+")) for r in range(2)]
        procs = [subprocess.Popen([binary, "--no-scaffold", "--rb-att", "2", rr, cr, "r", "c", "o"], stdout=logs[r][0], stderr=logs[r][1],
                                  env=dict(common, TLM_WORLD="2", TLM_RANK=str(r), TLM_DEVICE="0", TLM_RENDEZVOUS=rv, TLM_RCCL_LIB=fake_rccl,
                                           FAKE_RCCL_TIMEOUT_S="300")) for r in range(2)]
        try:
            for p in procs:
                p.wait(timeout=900)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        res = []
        for o, e in logs:
            o.seek(0), e.seek(0)
            res.append((o.read(), e.read()))
            o.close(), e.close()
        keep = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(("Loaded", "-"))]
        for r, (out, err) in enumerate(res):
            assert procs[r].returncode == 0, err[-3000:]
            m = re.search(r"sharded load of \S*rr\.paf: pieces \[(\d+), (\d+)\) brought by the helper process", err)
            assert m and (int(m.group(1)), int(m.group(2))) == ((0, 3) if r == 0 else (3, 7)), err[-3000:]
            assert "swap with the other ranks" in err
            assert keep(out) == keep(one.stdout) and keep(out), (out[:2000], one.stdout[:2000])
            m = re.search(r"Loaded (\d+)/(\d+) edges from \S*rr\.paf", out)
            assert m and int(m.group(2)) == reps * 4096
