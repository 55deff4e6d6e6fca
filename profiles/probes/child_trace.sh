#!/bin/bash
# profiles/probes/child_trace.sh <n> -- the host program's walk kernels under rocprofv3 --kernel-trace, n runs (fast / slow mode hunt)
O=$GRAFT_REPO_ROOT/gpurun_out/r02_bimodal; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for i in $(seq 1 $1); do
  rm -rf $O/prof_$i
  TLM_SCAFFOLD_ROCPROF=$O/prof_$i timeout -k 10 300 python3 $GRAFT_REPO_ROOT/telomeri_amd/scaffold_bench.py athaliana > $O/s_$i.json 2> $O/s_$i.err
  python3 - <<PY
import json, glob, csv
d=json.load(open("$O/s_$i.json")); print("run $i", d["scaffold_wall_s"], d.get("walk_kernel_ms"), flush=True)
for f in glob.glob("$O/prof_$i/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "walk_" in r["Kernel_Name"]:
            print("   ", r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms; start", int(r["Start_Timestamp"])%10**10/1e6, "scratch", r.get("Private_Segment_Size"), "grid", r.get("Grid_Size"), "queue", r.get("Queue_Id"))
PY
  find $O/prof_$i -name "*.csv" -size +512k -delete
done
