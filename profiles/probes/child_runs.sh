#!/bin/bash
# profiles/probes/child_runs.sh <n> [env...] -- n runs of the host program on the A. thaliana-shaped text: wall-clock and walk kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r02_child; mkdir -p $O
n=$1; shift
for i in $(seq 1 $n); do
  env "$@" timeout -k 10 300 python3 $GRAFT_REPO_ROOT/telomeri_amd/scaffold_bench.py athaliana > $O/s_$i.json 2> $O/s_$i.err
  python3 - <<PY
import json
d=json.load(open("$O/s_$i.json")); print("run $i", "$*", d["scaffold_wall_s"], d.get("walk_kernel_ms"), d["stages_s"], d["outside_stages_s"]["main_left_to_exit"], [h for h in d["host_timing"] if "unmapping" in h], flush=True)
PY
done
