#!/bin/bash
# profiles/probes/tok_ablate.sh <values...> -- the load phase of the host program on the A. thaliana-shaped text under TLM_TOK_ABLATE
# values (1 = no dictionary probes, 2 = no token loops; 0 = timing only).  The switch is NOT in the shipped library: apply
# profiles/experiments/r02_tokeniser_ablation.patch to csrc/paf_parse.hip and rebuild first.  TOK_TRACE=1 adds a rocprofv3 kernel trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_tokabl; mkdir -p $O
D=$(python3 - <<PY
import sys, json
sys.path.insert(0, "$R/profiles/probes"); sys.path.insert(0, "$R")
from telomeri_amd import scaffold_bench as sb
out = sb.run("athaliana", keep=True, extra_args=["--no-scaffold", "--rb-att", "1"])
print(out["dir"])
PY
)
cd /tmp; export TMPDIR=/tmp TLM_NO_UPLOAD_HELPER=1 TLM_HOST_TIMING=1
B="$R/telomeri_amd/bin/telomeri_hip --no-scaffold --rb-att 1 $D/rr_overlaps.paf $D/cr_overlaps.paf r c o"
for a in "$@"; do
  TLM_TOK_ABLATE=$a timeout -k 10 200 $B > $O/a_$a.out 2> $O/a_$a.err; echo "ablate $a rc $?"; grep -h "ablate\|device tokeniser" $O/a_$a.err
done
if [ -n "$TOK_TRACE" ]; then
  TLM_CLEAN_EXIT=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.out 2> $O/trace.err || echo "trace failed"
  find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-150
  find $O/trace -name "*.csv" -size +1M -delete
fi
rm -rf $D
