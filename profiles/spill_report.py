#!/usr/bin/env python3
"""Where the walk kernels touch scratch memory: per kernel the register / spill figures of the code object's metadata, and
every scratch_load / scratch_store of the product builds attributed to the source function and line it was emitted for.

    python profiles/spill_report.py [--out profiles/r03_spills.json]

Two compilations of csrc/walk_kernels.hip for gfx950 with the library's flags: one as shipped (its metadata is what the
.so carries), one with -gline-tables-only (line tables do not change the code much, but the figures of THAT build are
reported beside it so that the attribution is read with the right caveat).  The function of a line = the nearest
preceding definition in the source (inlined callees keep their own lines).  Loop depth from the assembler's block comments.
How often a branch RUNS is not in the assembly: that comes from the accounting launch (bench.py roofline.step_classes)."""
import argparse
import collections
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "telomeri_amd", "csrc", "walk_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S"]
META = ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def compile_asm(extra, out):
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def metadata(asm):
    res = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", asm)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        res[demangle(name.group(1))] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in META if re.search(r"\.%s:\s+(\d+)" % k, blk)}
    return res


def demangle(n):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(walk_\w+_kernel)ILb(\d)ELb(\d)ELb(\d)EEEv8WalkArgs", n)
    return "%s<SQRT=%s, ACCOUNT=%s, LITE=%s>" % (m.group(1), m.group(2), m.group(3), m.group(4)) if m else n


def functions_of_source():
    starts = []
    pat = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|static|inline|extern|void|int)\b.*?\b(\w+)\s*\([^;]*$")
    pending_template = False
    for i, ln in enumerate(open(SRC), 1):
        if ln.startswith(("__device__", "__global__", "template", "void ", "int ")) and "(" in ln and not ln.rstrip().endswith(";"):
            m = re.search(r"\b(\w+)\s*\(", ln.split("__launch_bounds__")[-1] if "__launch_bounds__" not in ln else ln.split(")", 1)[-1])
            if m and m.group(1) not in ("__attribute__", "__launch_bounds__", "aligned"):
                starts.append((i, m.group(1)))
    return starts


def function_of(line, starts):
    name = "?"
    for s, n in starts:
        if s <= line:
            name = n
        else:
            break
    return name


def attribute(asm, starts):
    res = {}
    sites = {}
    cur = None
    line = 0
    depth = 0
    chain = []
    for ln in asm.splitlines():
        m = re.match(r"^(_ZN12_GLOBAL__N_1\d+walk_\w+_kernelILb\dELb\dELb\dEEEv8WalkArgs):", ln)
        if m:
            cur = demangle(m.group(1))
            res[cur] = collections.Counter()
            sites[cur] = collections.Counter()
            depth = 0
            continue
        if cur is None:
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"\s+\.loc\s+\d+\s+(\d+)", ln)
        if m:
            line = int(m.group(1))
            chain = [line] + [int(x) for x in re.findall(r"walk_kernels\.hip:(\d+):", ln.split(";", 1)[1] if ";" in ln else "")[1:]]
            continue
        m = re.search(r"Depth=(\d+)", ln)
        if m and "Loop" in ln:
            depth = int(m.group(1))
        elif re.match(r"^\.LBB\d+_\d+:", ln) and "Loop" not in ln and "Depth" not in ln:
            pass
        m = re.match(r"\s+(scratch_(?:load|store)_\w+)", ln)
        if m:
            res[cur][(function_of(line, starts), line, m.group(1).split("_")[1], depth)] += 1
            # the call site inside the walk function itself: which instantiation (entries per lane) the code belongs to
            for c in chain:
                if function_of(c, starts) in ("mc_walk", "det_walk"):
                    sites[cur][c] += 1
                    break
    return res, sites


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_spills.json"))
    args = ap.parse_args()
    starts = functions_of_source()
    with tempfile.TemporaryDirectory() as tmp:
        shipped = metadata(compile_asm([], os.path.join(tmp, "a.s")))
        asm_g = compile_asm(["-gline-tables-only"], os.path.join(tmp, "g.s"))
    meta_g = metadata(asm_g)
    att, sites = attribute(asm_g, starts)
    src_lines = open(SRC).read().splitlines()
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    out = {"source": "telomeri_amd/csrc/walk_kernels.hip", "commit": commit + "+", "kernels": {}}
    for k in sorted(shipped):
        by_fn = collections.defaultdict(lambda: {"loads": 0, "stores": 0, "lines": set(), "max_loop_depth": 0})
        for (fn, line, kind, depth), n in att.get(k, {}).items():
            e = by_fn[fn]
            e["loads" if kind == "load" else "stores"] += n
            e["lines"].add(line)
            e["max_loop_depth"] = max(e["max_loop_depth"], depth)
        out["kernels"][k] = {"as_shipped": shipped[k], "with_line_tables": meta_g.get(k),
                             "scratch_instructions_by_call_site_in_the_walk": {
                                 "%d: %s" % (ln, src_lines[ln - 1].strip()[:110]): n for ln, n in sorted(sites.get(k, {}).items(), key=lambda kv: -kv[1])},
                             "scratch_instructions_by_function": {
                                 fn: {"loads": e["loads"], "stores": e["stores"], "max_loop_depth": e["max_loop_depth"],
                                      "source_lines": sorted(e["lines"])[:40]} for fn, e in sorted(by_fn.items(), key=lambda kv: -(kv[1]["loads"] + kv[1]["stores"]))}}
    json.dump(out, open(args.out, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, v["as_shipped"])
        for fn, e in v["scratch_instructions_by_function"].items():
            print("    %-22s loads %3d stores %3d depth %d lines %s" % (fn, e["loads"], e["stores"], e["max_loop_depth"], e["source_lines"][:12]))


if __name__ == "__main__":
    main()
