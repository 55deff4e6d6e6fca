#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof/<pass>/...) into the small files kept under
profiles/: the --stats kernel table restricted to this library's kernels, and per-kernel PMC sums.

  python profiles/summarize.py gpurun_out/prof profiles/r01_v1 [config-name]

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are KiB, collected in separate
--pmc passes; on gfx950 FETCH_SIZE reads exactly half the bytes of coalesced loads (checked here on
filter_score_kernel, whose byte count is known: 36 B read + 13 B written per record), so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  With a config name the walk_mc figure is also
written to profiles/traffic.json, which bench.py reports as roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import sys

OURS = ("walk_mc_kernel", "walk_det_kernel", "filter_score_kernel", "gather_kernel", "first_seen_kernel",
        "make_entries_kernel", "make_nodes_kernel", "row_ptr_kernel", "node_flags_kernel", "scan_units_kernel",
        "pack_paths_kernel", "det_count_kernel", "kept_fields_kernel", "walk_layout_kernel", "bfs_expand_kernel",
        "dup_flags_kernel", "copy_kept_kernel")


def short(name: str) -> str:
    for k in OURS:
        if k in name:
            return k
    return name[:60]


def main(src: str, dst_prefix: str, config: str = "") -> None:
    out = {}
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if any(k in r["Name"] for k in OURS)]
        with open(dst_prefix + "_kernel_stats.csv", "w") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in keep:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
        out["kernel_stats"] = {short(r["Name"]): {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                  "total_ms": float(r["TotalDurationNs"]) / 1e6} for r in keep}
    pmc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in sorted(glob.glob(os.path.join(src, "*"))):
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k not in OURS:
                    continue
                pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    res = {}
    for k, cs in pmc.items():
        res[k] = {c: {"sum": v, "dispatches": len(calls[k][c])} for c, v in cs.items()}
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            nf, nw = len(calls[k]["FETCH_SIZE"]), len(calls[k]["WRITE_SIZE"])
            res[k]["hbm_bytes_per_launch"] = (2 * cs["FETCH_SIZE"] / nf + cs["WRITE_SIZE"] / nw) * 1024
        if "TCC_HIT_sum" in cs:
            res[k]["l2_hit_rate"] = cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])
    out["pmc"] = res
    for name in ("trace", "fetch", "write", "sq", "tcc"):      # the bench lines printed under the profiler
        p = os.path.join(src, name + ".json")
        if os.path.exists(p) and os.path.getsize(p):
            try:
                b = json.loads(open(p).read().strip().splitlines()[-1])
                out.setdefault("bench_lines", {})[name] = {"value": b["value"], "ms_per_step": b["ms_per_step"],
                                                          "roofline": b["roofline"], "kernels": b.get("kernels")}
            except Exception:
                pass
    json.dump(out, open(dst_prefix + "_summary.json", "w"), indent=1, sort_keys=True)
    if config and "walk_mc_kernel" in res and "hbm_bytes_per_launch" in res["walk_mc_kernel"]:
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
        t = json.load(open(tp)) if os.path.exists(tp) else {}
        import subprocess
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                    cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() + "+"
        except Exception:
            commit = "?"
        epl = None
        for b in out.get("bench_lines", {}).values():
            epl = b["roofline"].get("evals_per_launch", epl)
        t[config] = {"walk_mc_kernel_hbm_bytes_per_launch": res["walk_mc_kernel"]["hbm_bytes_per_launch"],
                     "evals_per_launch": epl, "commit": commit,
                     "source": os.path.basename(dst_prefix) + "_summary.json"}
        json.dump(t, open(tp, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
