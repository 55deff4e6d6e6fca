#!/usr/bin/env python3
"""bench.py -- edge-extension evals/sec of the path build (BASELINE.json metric) on MI355X.

A "step" is one pass of the hot path over this rank's batch: the four builders of reference
Main.cpp:237-243 (MC/ES, MC/OS, DET/ES, DET/OS) over the rank's anchor bundle on the resident CSR, then
(N > 1) the single all-gather of the Path sets.  Inputs are resident in HBM before the timed region:
the synthetic A. thaliana-shaped overlaps are generated on the GPU (telomeri_amd.synth), filtered,
scored and turned into the CSR by the library; graph-build timings are reported beside the metric.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Strong scaling (north_star: "anchor contigs and their Monte-Carlo trial bundles shard across the 8 GPUs"): the
workload is the WHOLE A. thaliana-shaped config -- all 200 anchors x the reference's default 2000 Monte-Carlo trials
x {ES, OS} + the deterministic builders -- and rank r of N walks, of EVERY anchor, slice r of its trial bundle and of
its first edges (`--partition sliced`, tlm_dist_unit_share: the ranks' work is equal whatever the anchors weigh;
N = 8: 250 trials of each of the 200 anchors per GPU).  `--partition contiguous`: round 3's bundles of ceil(200 / N)
whole anchors; `--anchors-per-gpu K` restricts every rank to K whole anchors (diagnostics; the line then says so).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def launch_ranks(n: int) -> int:
    """`bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): this process becomes the launcher.  It never
    touches a GPU (nothing that initialises HIP has been imported yet): it starts N fresh children of this same command, one
    rank per GPU, with the rendezvous variables torch.distributed.run would set, lets rank 0 write the JSON line straight to
    this process's stdout, ends the others as soon as one of them fails, and returns the first non-zero status."""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))
    try:
        while live and rc == 0:
            for r in sorted(live):
                st = procs[r].poll()
                if st is None:
                    continue
                live.discard(r)
                if st != 0:
                    rc = st if st > 0 else 128 - st
                    log("bench.py: rank %d ended with status %d; ending the other ranks" % (r, st))
                    break
            time.sleep(0.05)
    finally:
        for r in live:
            if procs[r].poll() is None:
                procs[r].send_signal(signal.SIGTERM)
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="athaliana", help="telomeri_amd.synth.CONFIGS entry")
    ap.add_argument("--anchors-per-gpu", type=int, default=0, help="0 = the whole config is walked; K = K whole anchors per rank")
    ap.add_argument("--partition", choices=("sliced", "contiguous"), default="sliced",
                    help="N > 1: a slice of every anchor's units per rank (default) or bundles of whole anchors")
    ap.add_argument("--no-scaffold", action="store_true", help="skip the PAF-in / FASTA-out wall-clock leg")
    ap.add_argument("--scaffold-config", default=None, help="config of the scaffold leg (default: --config)")
    ap.add_argument("--rb-att", type=int, default=2000)
    ap.add_argument("--bt-att", type=int, default=60)
    ap.add_argument("--len-thr", type=int, default=4_700_000)
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--no-det", action="store_true", help="skip the deterministic builders (diagnostics only)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--separate-launches", action="store_true", help="one launch per builder and metric (diagnostics)")
    ap.add_argument("--serial-builders", action="store_true",
                    help="the deterministic launch after the Monte-Carlo launch has ended instead of in its tail (diagnostics)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL); gloo only to rehearse N>1 on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layout-accounting", action="store_true",
                    help="skip the untimed accounting launch behind roofline.frac_layout (under a profiler: its kernels would "
                         "be counted with the product's)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the cpu_baseline leg (reported as cores)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if os.environ.get("TLM_BENCH_LAUNCH_ECHO"):      # launcher test (CPU tier): what a rank was started with, nothing else
        status = int(os.environ.get("TLM_BENCH_LAUNCH_ECHO_FAIL", "-1")) == int(os.environ.get("RANK", "0"))
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}), flush=True)
        if os.environ.get("TLM_BENCH_LAUNCH_ECHO_FAIL") and not status:
            time.sleep(30)      # the launcher must end this rank when the failing one has gone
        sys.exit(7 if status else 0)

    from telomeri_amd import cpu_share
    host_threads = cpu_share.limit_thread_pools()       # before numpy / torch start their pools (DESIGN.md "The CPU quota")
    import numpy as np
    import torch
    import torch.distributed as dist
    from telomeri_amd import capi, pathgather, synth
    torch.set_num_threads(max(1, min(torch.get_num_threads(), host_threads)))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    t0 = time.time()
    ctx = capi.Context(local_rank)
    if args.waves_per_cu:
        ctx.set_walk_tuning(args.waves_per_cu, 0)
    # the product's exchange: an RCCL communicator owned by the library context (tlm_dist_init); torch.distributed only
    # carries the 128-byte unique id to the other ranks and the final max / sum of the timing
    # TLM_EXCHANGE: "lazy" (default: path summaries to every rank, node lists on demand -- what the host stages consume),
    # "full" (every node list to every rank), "torch" (the harness's gather over torch.distributed)
    exchange_mode = os.environ.get("TLM_EXCHANGE", "lazy")
    # (with TLM_RCCL_LIB -- tests/fake_rccl, several ranks on one GPU -- the library's exchange also runs in a gloo rehearsal)
    use_abi_exchange = (world > 1 and exchange_mode in ("lazy", "full")
                        and (args.dist_backend == "nccl" or bool(os.environ.get("TLM_RCCL_LIB"))))
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")     # where the harness's own small collectives live
    if use_abi_exchange:
        # every rank tries; if ANY of them cannot bring the library's communicator up, all fall back together to the
        # harness's gather over torch.distributed (the line then says so)
        ok, why = 1, ""
        try:
            box = [capi.dist_unique_id() if rank == 0 else None]
        except Exception as e:
            box, ok, why = [None], 0, repr(e)
        dist.broadcast_object_list(box, src=0)
        if ok and box[0] is not None:
            try:
                ctx.dist_init(rank, world, box[0])
            except Exception as e:
                ok, why = 0, repr(e)
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            log("rank %d: library communicator unavailable (%s); falling back to torch.distributed for the exchange" % (rank, why))
            try:
                ctx.dist_finalize()
            except Exception:
                pass
            use_abi_exchange = False
            exchange_mode = "torch (fallback: %s)" % (why or "another rank failed")
    ds = synth.make_config(args.config, device=str(dev))
    torch.cuda.synchronize(dev)
    t_gen = time.time() - t0
    fp = capi.FilterParams.default()
    # the graph is built twice: the first build pays one-off costs (code-object loading of the rocPRIM kernels, first
    # hipMalloc of GB-sized buffers), the second one is what a build costs
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    cold_build_ms = float(g.info().build_ms)
    del g
    throttle0 = cpu_share.throttle_counters()
    t0 = time.time()
    g = ctx.build_graph(fp, [ds.rr, ds.cr], [False, True], ds.n_reads + ds.n_contigs, ds.name_is_ctg())
    t_build = time.time() - t0
    throttle1 = cpu_share.throttle_counters()
    gi = g.info()
    n_records = int(gi.n_records)
    del ds
    torch.cuda.empty_cache()
    if rank == 0:
        log("dataset %s: %d records generated in %.1fs; graph %d nodes / %d adjacency entries / %d anchors built in "
            "%.2fs (device %.1f ms, filter_score %.2f ms); max degree %d; unsafe nodes %s; BFS levels %d" % (
                args.config, n_records, t_gen, gi.n_nodes, gi.n_adj, gi.n_anchors, t_build, gi.build_ms, gi.filter_ms,
                gi.max_degree, list(gi.unsafe_nodes), gi.bfs_levels))

    wp = capi.WalkParams.default(args.rb_att, args.bt_att, args.len_thr)
    n_anchors = int(gi.n_anchors)
    apg = args.anchors_per_gpu if args.anchors_per_gpu > 0 else (n_anchors + world - 1) // world
    a0, a1 = min(rank * apg, n_anchors), min(rank * apg + apg, n_anchors)
    mc_range = (a0 * args.rb_att, a1 * args.rb_att)
    _, det_off = g.anchor_table()
    d0, d1 = int(det_off[a0]), int(det_off[a1])
    # the sliced partition: this rank's units as lists (a slice of every anchor's trials / first edges)
    sliced = (world > 1 and args.partition == "sliced" and args.anchors_per_gpu == 0
              and not (args.separate_launches or args.serial_builders or args.no_det))
    mc_units = g.unit_share(rank, world, wp, False) if sliced else None
    det_units = g.unit_share(rank, world, wp, True) if sliced else None

    def walk_mc_pair():         # this rank's Monte-Carlo units under both metrics (accounting, post-passes: untimed)
        return g.walk_list(False, capi.METRIC_ES, capi.METRIC_OS, wp, mc_units) if sliced else g.walk_mc2(capi.METRIC_ES, capi.METRIC_OS, wp, *mc_range)

    def walk_det_pair():
        return g.walk_list(True, capi.METRIC_ES, capi.METRIC_OS, wp, det_units) if sliced else g.walk_det2(capi.METRIC_ES, capi.METRIC_OS, d0, d1)

    exch = {}
    state = {"abi_exchange": use_abi_exchange, "fallback": None}

    def one_step():
        # the four builders of Main.cpp:237-243; the two metrics of a builder share one persistent launch
        sets, ctrs = [], []
        if sliced:
            sets = list(g.walk_builders_list(capi.METRIC_ES, capi.METRIC_OS, wp, mc_units, det_units))
            ctrs = [("mc", sets[0].counters), ("mc", sets[1].counters), ("det", sets[2].counters), ("det", sets[3].counters)]
        elif not (args.separate_launches or args.serial_builders or args.no_det):
            # one call: the deterministic launch is queued behind the Monte-Carlo launch on a second stream
            sets = list(g.walk_builders(capi.METRIC_ES, capi.METRIC_OS, wp, mc_range, (d0, d1)))
            ctrs = [("mc", sets[0].counters), ("mc", sets[1].counters), ("det", sets[2].counters), ("det", sets[3].counters)]
        elif args.separate_launches:
            for metric in (capi.METRIC_ES, capi.METRIC_OS):
                ps = g.walk_mc(metric, wp, *mc_range)
                sets.append(ps); ctrs.append(("mc", ps.counters))
        else:
            a, b = g.walk_mc2(capi.METRIC_ES, capi.METRIC_OS, wp, *mc_range)
            sets += [a, b]; ctrs += [("mc", a.counters), ("mc", b.counters)]
        if not args.no_det and (args.separate_launches or args.serial_builders):
            if args.separate_launches:
                for metric in (capi.METRIC_ES, capi.METRIC_OS):
                    ps = g.walk_det(metric, d0, d1)
                    sets.append(ps); ctrs.append(("det", ps.counters))
            else:
                a, b = g.walk_det2(capi.METRIC_ES, capi.METRIC_OS, d0, d1)
                sets += [a, b]; ctrs += [("det", a.counters), ("det", b.counters)]
        done = False
        if state["abi_exchange"]:       # ONE RCCL exchange through the library's own communicator
            try:
                if exchange_mode == "lazy":     # tlm_pathset_gather_summaries
                    gathered = capi.Gathered(sets)
                    xi = gathered.info
                else:                           # tlm_pathset_allgather
                    gathered, xi = capi.PathSet.allgather(sets)
                n_total = gathered.n_paths
                exch["bytes_sent"] = int(xi.bytes_sent); exch["bytes_received"] = int(xi.bytes_received)
                exch["ms"] = exch.get("ms", 0.0) + float(xi.device_ms); exch["calls"] = exch.get("calls", 0) + 1
                exch["slot_bits"] = int(xi.slot_bits)
                del gathered
                done = True
            except capi.TlmError as e:
                # A failure that every rank AGREED on before data moved (the failing rank's own code there, TLM_EPEER on the
                # others, communicator intact) is one all ranks see together: they go on with the harness's gather for this
                # and the remaining steps, and the line says so.  A failure in the MIDDLE of a collective sequence aborts the
                # communicator (include/telomeri_hip.h, "FAILING TOGETHER"): the peers may still sit inside RCCL, so this
                # rank must END -- the launcher then ends the others -- instead of waiting for them in another collective.
                broken = "aborted" in str(e)
                log("rank %d: the library's exchange failed (%s)%s" % (rank, e, "; the communicator is gone: leaving the job" if broken
                                                                        else "; the harness's gather takes over"))
                if broken:
                    sys.stdout.flush()
                    sys.stderr.flush()
                    os._exit(5)
                state["abi_exchange"] = False
                state["fallback"] = str(e)[:200]
        if not done and world > 1:      # the harness's own gather over torch.distributed (gloo rehearsals; fallback)
            packed = [pathgather.PackedPaths(*ps.to_device_tensors(dev)) for ps in sets]
            gathered = pathgather.allgather_paths(packed)
            n_total = sum(p.n_paths for p in gathered)
        elif not done:
            n_total = sum(ps.n_paths for ps in sets)
        return ctrs, n_total

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_step()
    sync()
    exch.pop("ms", None); exch.pop("calls", None)
    t0 = time.time()
    evals = 0
    mc_ms = mc_evals = mc_launches = 0.0
    det_ms = det_evals = det_launches = det_busy_ms = 0.0
    paths_total = 0
    for _ in range(args.steps):
        ctrs, paths_total = one_step()
        for kind, c in ctrs:
            evals += int(c.evals)
            if kind == "mc":
                mc_ms += c.kernel_ms; mc_evals += int(c.evals); mc_launches += c.kernel_launches
            else:
                det_ms += c.kernel_ms; det_evals += int(c.evals); det_launches += c.kernel_launches
                det_busy_ms += c.kernel_busy_ms
    sync()
    elapsed = time.time() - t0
    # every rank's dominant kernel: average launch and its fraction of the yardstick (min / max over the ranks go into the line)
    my_launch_ms = mc_ms / max(mc_launches, 1)
    my_frac = 8.0 * mc_evals / (mc_ms * 1e-3) / 8e12 if mc_ms > 0 else 0.0
    my_det_ms = det_busy_ms / max(args.steps, 1)
    spread = None
    if world > 1:
        t = torch.tensor([elapsed, float(evals)], dtype=torch.float64, device=cdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        evals_all = float(tsum[1])
        k = torch.tensor([my_launch_ms, my_frac, my_det_ms, float(mc_evals), -my_launch_ms, -my_frac, -my_det_ms, -float(mc_evals)],
                         dtype=torch.float64, device=cdev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)        # one collective: maxima, and minima as maxima of the negatives
        spread = {"avg_launch_ms_max": float(k[0]), "avg_launch_ms_min": -float(k[4]), "frac_max": float(k[1]), "frac_min": -float(k[5]),
                  "walk_det_ms_per_step_max": float(k[2]), "walk_det_ms_per_step_min": -float(k[6]),
                  "mc_evals_max": float(k[3]), "mc_evals_min": -float(k[7]),
                  "of": "walk_mc_kernel of every rank (rank 0's figures are the block's own); evals: this rank's share per %d steps" % args.steps}
    else:
        evals_all = float(evals)

    if rank != 0:
        if world > 1:
            if not args.no_scaffold:        # this rank's part of the scaffold leg (rank 0 gets there after its own reporting)
                g = None
                scaffold_leg_all_ranks(args, {}, ctx, dev, local_rank, rank, world, cdev, dist, torch)
            dist.barrier()
            dist.destroy_process_group()
        return

    value = evals_all / elapsed
    # roofline of the dominant kernel (walk_mc_kernel): 8 algorithmic bytes per eval
    ach = 8.0 * mc_evals / (mc_ms * 1e-3) / 1e9 if mc_ms > 0 else 0.0
    # ... and against the bytes the walk layout REALLY has to move: compact rows are 6 bytes per entry, lists are read in
    # whole lane widths, every step also reads a row record and a {q_start, t_start} pair.  Counted exactly by ONE untimed
    # launch of the accounting build of the same kernels over the same units (tlm_ctx_set_walk_accounting).
    layout = {}
    try:
        if args.no_layout_accounting:
            raise RuntimeError("switched off")
        ctx.set_walk_accounting(True)
        a, b = walk_mc_pair()
        lb_mc, ev_mc = int(a.counters.layout_bytes), int(a.counters.evals) + int(b.counters.evals)
        a_classes = list(a.counters.step_class)
        del a, b
        layout["mc"] = (lb_mc, ev_mc)
        layout["mc_classes"] = [int(x) for x in a_classes]
        if not args.no_det and (d1 > d0 or sliced):
            a, b = walk_det_pair()
            layout["det"] = (int(a.counters.layout_bytes), int(a.counters.evals) + int(b.counters.evals))
            layout["det_classes"] = [int(x) for x in a.counters.step_class]
            del a, b
    except Exception as e:
        log("layout accounting failed: %r" % (e,))
    finally:
        ctx.set_walk_accounting(False)
    # HBM traffic of the dominant kernel cannot be counted from inside this process (PMC counters need rocprofv3 around
    # it): it is taken from the tracked profile of THIS workload (profiles/run_profile.sh -> profiles/summarize.py) and
    # reported only when that profile's launch processed the same number of evals as the launches timed here
    traffic = traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    evals_per_launch = mc_evals / max(mc_launches, 1)
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath)).get(args.config, {})
            if t and abs(t.get("evals_per_launch", 0) - evals_per_launch) <= 1e-6 * evals_per_launch:
                traffic = t.get("walk_mc_kernel_hbm_bytes_per_launch")
                traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, source tree %s)" % (
                    t.get("source"), t.get("commit", "?"))
        except Exception:
            traffic = None
    out = {
        "metric": "edge-extension evals/sec in path build",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s-shaped synthetic overlaps (telomeri_amd.synth.CONFIGS['%s']): %d PAF records -> "
                               "%d nodes, %d adjacency entries; %d of its %d anchors walked (%d per GPU) x (%d MC trials x "
                               "{ES,OS} + DET first edges x {ES,OS})" % (args.config, args.config, n_records, gi.n_nodes,
                                                                         gi.n_adj, min(apg * world, n_anchors), n_anchors,
                                                                         apg, args.rb_att),
                   "anchors_total": n_anchors, "anchors_walked": min(apg * world, n_anchors), "anchors_per_gpu": apg, "rebuild_attempts": args.rb_att, "backtrack_attempts": args.bt_att,
                   "len_threshold": args.len_thr, "seed_policy": "per-walk mt19937",
                   "partition": ("sliced: every rank walks slice r of EVERY anchor's trial bundle and first edges (tlm_dist_unit_share)"
                                 if sliced else "contiguous bundles of whole anchors per rank")
                                + ", CSR replicated, one exchange of Paths (inside the timed step)"},
        "roofline": roofline_block(ach, traffic, traffic_source, mc_ms, mc_launches, evals_per_launch, layout.get("mc"),
                                   layout.get("mc_classes")),
        "kernels": {"walk_mc_ms_per_step": mc_ms / max(args.steps, 1),
                    # the deterministic launch is queued behind the Monte-Carlo one: from launch to end (queued) and from
                    # its first workgroup in to its last one out (busy, device clock)
                    "walk_det_queued_ms_per_step": det_ms / max(args.steps, 1),
                    "walk_det_ms_per_step": det_busy_ms / max(args.steps, 1),
                    "walk_mc_evals_per_s": mc_evals / (mc_ms * 1e-3) if mc_ms else None,
                    "walk_det_evals_per_s": det_evals / (det_busy_ms * 1e-3) if det_busy_ms else None,
                    "walk_det_roofline": det_roofline(det_evals, det_busy_ms, layout.get("det"), layout.get("det_classes")),
                    "graph_build_cold_ms": cold_build_ms, "graph_build_device_ms": gi.build_ms, "filter_score_ms": gi.filter_ms,
                    "graph_build_wall_ms": gi.build_wall_ms, "graph_build_alloc_ms": gi.build_alloc_ms,
                    "graph_build_host_syncs": gi.host_syncs,
                    "graph_build_phases_ms": {k: round(v, 3) for k, v in gi.phases().items()},
                    # the host's time between the same marks (waits for the device excluded) and those waits: a phase that
                    # is long on the device timeline and long HERE was held up on the host, not by its kernels
                    "graph_build_host_phases_ms": {k: round(v, 3) for k, v in gi.host_phases().items()},
                    "graph_build_sync_wait_ms": gi.sync_wait_ms,
                    "graph_build_cpu_throttled": {k: throttle1.get(k, 0) - throttle0.get(k, 0) for k in throttle1},
                    "host_threads": host_threads,
                    "filter_score_records_per_s": n_records / (gi.filter_ms * 1e-3) if gi.filter_ms else None,
                    "filter_score_GBps": (36.0 + 13.0) * n_records / (gi.filter_ms * 1e-3) / 1e9 if gi.filter_ms else None},
        "paths_per_step": paths_total,
    }
    if spread:
        out["roofline"]["ranks"] = spread
    if world > 1:
        out["exchange"] = {"transport": ("tlm_pathset_gather_summaries (RCCL all-gather of 45-byte path summaries; node lists "
                                         "stay with their owners until tlm_gathered_fetch asks)" if exchange_mode == "lazy" else
                                         "tlm_pathset_allgather (RCCL send/recv of every node list)") if state["abi_exchange"]
                           else "telomeri_amd.pathgather over torch.distributed/%s%s" % (
                               args.dist_backend, " (the library's exchange failed: %s)" % state["fallback"] if state["fallback"] else ""),
                           "allgather_bytes_sent_per_rank": exch.get("bytes_sent"),
                           "allgather_bytes_received_per_rank": exch.get("bytes_received"),
                           "slot_bits_on_the_wire": exch.get("slot_bits"),
                           "ms_per_step_rank0": exch["ms"] / max(exch.get("calls", 1), 1) if "ms" in exch else None}

    # SURVEY.md §8f row 1, outside the timed region: paths_ = concat of the four sets, filterUnique,
    # getPathsBetweenAnchors on the device
    try:
        sets = list(walk_mc_pair())
        if not args.no_det:
            sets += list(walk_det_pair())
        torch.cuda.synchronize(dev)
        # three times: the first pass obtains its two 5 GB results from the driver (the timed steps never concatenate; that
        # call takes 1 ms in one run and 250 ms in the next), the others find them in the library's cache -- the passes'
        # own times are the faster of those two
        first, best = None, None
        for rep in range(3):
            mi0 = ctx.memory_info()
            t0 = time.time()
            allp = capi.PathSet.concat(sets)
            t1 = time.time()
            uniq = allp.filter_unique()
            t2 = time.time()
            pairs = uniq.anchor_pairs()
            t3 = time.time()
            mi1 = ctx.memory_info()
            nbytes = 8.0 * allp.n_nodes_total          # nodes + slots of every path, read once
            res = {"paths_in": allp.n_paths, "paths_unique": uniq.n_paths, "anchor_pairs": len(pairs),
                   "path_nodes": allp.n_nodes_total, "concat_ms": 1e3 * (t1 - t0),
                   "filter_unique_ms": 1e3 * (t2 - t1), "anchor_pairs_ms": 1e3 * (t3 - t2),
                   "filter_unique_GBps": 2 * nbytes / max(t2 - t1, 1e-9) / 1e9,
                   "driver_allocations": int(mi1.driver_allocs - mi0.driver_allocs),
                   "driver_allocation_ms": float(mi1.driver_alloc_ms - mi0.driver_alloc_ms)}
            if rep == 0:
                first = res
            else:
                best = res if best is None else dict(res, **{k: min(res[k], best[k]) for k in ("concat_ms", "filter_unique_ms", "anchor_pairs_ms")})
            if rep < 2:
                del allp, uniq
        best["filter_unique_GBps"] = 2 * nbytes / max(best["filter_unique_ms"] * 1e-3, 1e-9) / 1e9
        out["post_passes"] = dict(best, first_pass={k: first[k] for k in ("concat_ms", "filter_unique_ms", "anchor_pairs_ms",
                                                                         "driver_allocations", "driver_allocation_ms")})
        del sets, allp, uniq
    except Exception as e:      # reported, never fatal for the metric line
        out["post_passes"] = {"error": str(e)}

    if not args.no_cpu_baseline:        # rank 0, outside the timed region; the same bounded sample at every N
        out["cpu_baseline"] = cpu_baseline(args, g, wp, g.download(cold=False), (0, g.mc_units(wp)) if sliced else mc_range, capi)
    if world == 1 and not args.no_scaffold:
        # the metric's second half: PAF + FASTA TEXT in -> scaffold FASTA out through the C++ host program, on the same
        # config, reference argv and defaults (telomeri_amd/scaffold_bench.py; outside the timed region above)
        # nothing of this process stays on the device while the program runs
        del g
        import gc
        gc.collect()
        ctx.trim()
        torch.cuda.empty_cache()
        mi = ctx.memory_info()
        out["device_memory_kept_during_scaffold_leg"] = {"library_live_bytes": int(mi.live_bytes), "library_cached_bytes": int(mi.cached_bytes),
                                                         "torch_reserved_bytes": int(torch.cuda.memory_reserved())}
        try:
            from telomeri_amd import scaffold_bench as sb
            cfg = args.scaffold_config or args.config
            sc = sb.run(cfg, device=str(dev))
            out["scaffold_wall_s"] = sc.get("scaffold_wall_s")
            out["scaffold"] = sc
        except Exception as e:      # reported, never fatal for the metric line
            out["scaffold_wall_s"] = None
            out["scaffold"] = {"error": repr(e)}
    if world > 1 and not args.no_scaffold:
        g = None
        scaffold_leg_all_ranks(args, out, ctx, dev, local_rank, rank, world, cdev, dist, torch)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def scaffold_leg_all_ranks(args, out, ctx, dev, local_rank, rank, world, cdev, dist, torch):
    """The metric's second half at N > 1: the C++ host program as an N-rank job, one process per GPU (rank r of this bench
    starts rank r of the job on ITS device), on the text rank 0 has rendered: sharded load (every rank uploads 1/N of the
    overlap text, one swap over xGMI), sliced walks, the lazy exchange, host stages on every rank, rank 0 writes the
    scaffold.  Outside the timed region; every phase's failure is reported in the line and never fatal; every rank takes
    every collective of this function whatever happened to it."""
    import gc
    import shutil
    from telomeri_amd import scaffold_bench as sb
    gc.collect()
    ctx.trim()
    torch.cuda.empty_cache()
    box = [None]
    if rank == 0:
        try:
            box = [sb.prepare(args.scaffold_config or args.config, device=str(dev))]
        except Exception as e:
            out["scaffold_wall_s"] = None
            out["scaffold"] = {"error": "rendering the input failed: " + repr(e)}
    dist.broadcast_object_list(box, src=0)
    prep = box[0]
    if prep is None:
        return
    res, wall = None, 0.0
    try:
        renv = {"TLM_WORLD": world, "TLM_RANK": rank, "TLM_DEVICE": local_rank, "TLM_RENDEZVOUS": os.path.join(prep["tmp"], "rendezvous"),
                "LOCAL_WORLD_SIZE": os.environ.get("LOCAL_WORLD_SIZE", str(world))}
        if args.share_gpu:      # a rehearsal on ONE GPU: every rank's processes pile up on it (the boxes allow six): no helper processes
            renv["TLM_NO_UPLOAD_HELPER"] = "1"
        res = sb.execute(prep, rank_env=renv, timeout_s=float(os.environ.get("TLM_SCAFFOLD_TIMEOUT_S", "240")))
        wall = float(res.get("scaffold_wall_s") or 0.0)
        ok = 1 if res.get("exit") == 0 else 0
    except Exception as e:
        res, ok = {"error": repr(e)}, 0
    t = torch.tensor([wall, float(ok), float(res.get("stages_s", {}).get("load_s", 0.0) if isinstance(res, dict) else 0.0)], dtype=torch.float64, device=cdev)
    tmax, tmin = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    if rank == 0:
        all_ok = tmin[1].item() > 0.5
        out["scaffold_wall_s"] = round(tmax[0].item(), 3) if all_ok else None       # the job's wall-clock = its slowest rank's
        out["scaffold"] = dict(res, ranks=world, all_ranks_ok=all_ok, load_s_min_max_over_ranks=[tmin[2].item(), tmax[2].item()],
                               is_="telomeri_hip as a job of %d ranks, one per GPU (rank 0's block; scaffold_wall_s = the slowest rank)" % world)
    dist.barrier()
    if rank == 0:
        shutil.rmtree(prep["tmp"], ignore_errors=True)


STEP_CLASS_NAMES = ("compact_k4", "compact_k6", "compact_k8", "compact_k12", "row8_k4", "row8_k6", "row8_k8", "row8_k12",
                    "long_list", "serial")


def class_shares(classes):
    tot = float(sum(classes)) if classes else 0.0
    return {n: round(c / tot, 5) for n, c in zip(STEP_CLASS_NAMES, classes) if c} if tot else None


def roofline_block(ach, traffic, traffic_source, mc_ms, mc_launches, evals_per_launch, layout, classes=None):
    """`frac` stays what SURVEY.md §8(d) defines: 8 algorithmic bytes per eval against 8 TB/s.  Since the compact rows
    (6 bytes per entry) that figure is no longer a lower bound on the bytes moved, so `frac` CAN exceed 1 on graphs whose
    steps run mostly on compact rows; `frac_layout` prices the same launch against the bytes its loads really ask the
    layout for (counted by the accounting build) and is the one that is bounded by 1."""
    avg_ms = mc_ms / max(mc_launches, 1)
    r = {"bound": "hbm", "kernel": "walk_mc_kernel", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
         "frac": ach / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
         "achieved_is": "algorithmic bytes (8 B per eval) per second, not HBM traffic",
         "launches": int(mc_launches), "avg_launch_ms": avg_ms,
         "algorithmic_bytes_per_eval": 8, "evals_per_launch": evals_per_launch}
    if traffic and avg_ms > 0:
        r["traffic_TBps"] = traffic / (avg_ms * 1e-3) / 1e12
    if layout and layout[1] > 0 and avg_ms > 0:
        lb, ev = layout
        r["layout_bytes_per_eval"] = lb / ev
        r["layout_bytes_per_launch"] = lb * (evals_per_launch / ev)
        r["achieved_layout"] = r["layout_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
        r["frac_layout"] = r["achieved_layout"] / 8000.0
        r["layout_is"] = ("bytes the steps' loads ask the walk layout for, counted by an untimed launch of the accounting "
                          "build over the same units: list entries as read (6 B compact / 8 B rows, whole lane widths), "
                          "8 B row record, 8 B {q_start, t_start}")
    if classes:
        r["step_classes"] = class_shares(classes)       # share of the steps by the form of their selection (accounting launch)
    if r.get("frac_layout", 0.0) > 1.0:
        r["frac_layout_note"] = ("frac_layout > 1: the steps ask the memory system for more bytes per second than HBM's 8 TB/s because "
                                 "the XCDs' L2s serve part of them (whole anchors are dealt to one XCD on graphs with long lists); "
                                 "traffic_TBps, where a PMC profile of this workload exists, is what left the L2s")
    if r["frac"] > 1.0:
        r["frac_note"] = ("frac > 1: the yardstick (8 B per eval) is SURVEY.md's figure, most steps of this graph read 6-byte "
                          "compact rows; frac_layout is the bounded figure")
    return r


def det_roofline(det_evals, det_busy_ms, layout, classes=None):
    if not det_busy_ms:
        return None
    r = {"kernel": "walk_det_kernel", "achieved": 8.0 * det_evals / (det_busy_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s"}
    r["frac"] = r["achieved"] / 8000.0
    if layout and layout[1] > 0:
        r["layout_bytes_per_eval"] = layout[0] / layout[1]
        r["frac_layout"] = r["layout_bytes_per_eval"] * det_evals / (det_busy_ms * 1e-3) / 8e12
    if classes:
        r["step_classes"] = class_shares(classes)
    return r


def cpu_baseline(args, g, wp, h, mc_range, capi):
    """The oracle (CPU restatement of the reference, kind "port") timed on this box's host cores on a
    bounded sample of the SAME workload: the first units of this rank's MC/ES bundle.  Also checks that
    the device reports identical per-unit evals for that sample."""
    import numpy as np
    from oracle import pyoracle as po
    og = po.Graph.from_csr(h.row_ptr, h.nbr, h.q_start, h.t_start, h.os, h.es, h.node_len, h.node_anchor)
    owp = po.WalkParams(wp.rebuild_attempts, wp.backtrack_attempts, wp.len_threshold)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, args.cpu_threads))   # a 1-GPU box's CPU share is 16 cores
    # calibrate on a few units, single thread
    u0 = mc_range[0]
    t0 = time.time()
    _, c1, _ = og.walk_mc(po.METRIC_ES, owp, po.SEED_PER_WALK, u0, u0 + 4, 1)
    dt = max(time.time() - t0, 1e-4)
    per_unit = dt / 4
    n1 = int(max(4, min(mc_range[1] - u0, args.cpu_seconds * 0.4 / per_unit)))
    t0 = time.time()
    _, c1, s1 = og.walk_mc(po.METRIC_ES, owp, po.SEED_PER_WALK, u0, u0 + n1, 1)
    dt1 = time.time() - t0
    nall = int(max(cores * 4, min(mc_range[1] - u0, args.cpu_seconds * 0.6 * cores / per_unit)))
    t0 = time.time()
    _, call, sall = og.walk_mc(po.METRIC_ES, owp, po.SEED_PER_WALK, u0, u0 + nall, cores)
    dtall = time.time() - t0
    # parity of the sample against the device
    ps = g.walk_mc(capi.METRIC_ES, wp, u0, u0 + nall)
    dstat = ps.unit_stats()
    same = bool(np.array_equal(dstat["evals"], sall["evals"]) and np.array_equal(dstat["status"], sall["status"])
                and np.array_equal(dstat["length"], sall["length"]))
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    ref_ratio = ref_src = None
    try:        # the unmodified reference against this port, one thread each, same walks (build container; oracle/ref_vs_oracle_timing.py)
        rp = os.path.join(ROOT, "profiles", "r03_reference_vs_oracle.json")
        run = [r for r in json.load(open(rp))["runs"] if r["config"] == "toy"][0]
        ref_ratio = round(float(run["reference_evals_per_s"]) / float(run["oracle_evals_per_s_1thread"]), 3)
        ref_src = "profiles/r03_reference_vs_oracle.json (toy set, build container: the unmodified reference %.3g evals/s, this port %.3g)" % (
            run["reference_evals_per_s"], run["oracle_evals_per_s_1thread"])
    except Exception:
        pass
    return {"value": call.evals / dtall, "unit": "evals/s", "cores": cores, "kind": "port",
            "reference_over_port_1thread": ref_ratio, "reference_over_port_source": ref_src,
            "sample": "MC/ES units [%d,%d) of the same graph on %d threads (%.1fs); 1 thread: %d units, %.3g evals/s" % (
                u0, u0 + nall, cores, dtall, n1, c1.evals / dt1),
            "value_1thread": c1.evals / dt1, "cpu_model": model, "sample_matches_device": same}


if __name__ == "__main__":
    main()
