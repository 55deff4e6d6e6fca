#!/usr/bin/env bash
# profiles/gpu_chain.sh <outdir> <<< "name|timeout_s|command" lines
# Runs GPU steps one after the other on the box, each under its own `timeout -k 10`, stdout+stderr to <outdir>/<name>.log.
# A step that FAILS lets the chain go on (its status is recorded); a step that is KILLED at its limit stops the chain:
# nothing further is started on a GPU that may be in a bad state.  The same goes for a step that CRASHED (abort, segfault) or whose
# log holds a GPU memory fault.
out=$1
mkdir -p "$out"
: > "$out/chain_status.txt"
while IFS='|' read -r name limit cmd
do
    [ -z "$name" ] && continue
    echo "=== $name (limit ${limit}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$limit" bash -o pipefail -c "$cmd" > "$out/$name.log" 2>&1
    st=$?
    echo "$name status=$st seconds=$(( $(date +%s) - start ))" | tee -a "$out/chain_status.txt"
    tail -n 4 "$out/$name.log"
    if [ $st -eq 124 ] || [ $st -eq 137 ] || [ $st -eq 134 ] || [ $st -eq 139 ] || grep -q "Memory access fault" "$out/$name.log"
    then
        echo "$name was killed at its limit, crashed or faulted on the GPU: stopping the chain" | tee -a "$out/chain_status.txt"
        exit 1
    fi
done
exit 0
