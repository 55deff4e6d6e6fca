#!/bin/bash
# profiles/probes/tokeniser_pmc.sh -- rocprofv3 PMC passes over the host program's load phase (device tokeniser kernels) on the
# A. thaliana-shaped text.  One process (no upload helper), destructors at exit (the profiler writes its files there).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TOK_OUT:-r02_tokpmc}; mkdir -p $O
D=$(python3 - <<PY
import sys, json
sys.path.insert(0, "$R/profiles/probes"); sys.path.insert(0, "$R")
from telomeri_amd import scaffold_bench as sb
out = sb.run("athaliana", keep=True, extra_args=["--no-scaffold", "--rb-att", "1"])
print(out["dir"])
PY
)
echo "files in $D"; ls -la $D | head
cd /tmp; export TMPDIR=/tmp TLM_NO_UPLOAD_HELPER=1 TLM_CLEAN_EXIT=1
B="$R/telomeri_amd/bin/telomeri_hip --no-scaffold --rb-att 1 $D/rr_overlaps.paf $D/cr_overlaps.paf r c o"
run() { n=$1; shift; timeout -k 10 200 rocprofv3 "$@" --output-format csv -d $O/$n -- $B > $O/$n.out 2> $O/$n.err || echo "pass $n failed"; }
run trace --kernel-trace --stats
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run fetch --pmc FETCH_SIZE
rm -rf $D
for k in parse_tiles count_lines collect_slots; do echo "== $k"; python3 $R/profiles/probes/pmc_sum.py $O $k; done
find $O -name "*.csv" -size +1M -delete
