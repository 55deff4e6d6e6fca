#!/bin/bash
# profiles/run_profile.sh <out-subdir> [bench args...] -- run ON THE GPU BOX (through gpurun):
# rocprofv3 kernel trace + the PMC passes the roofline block needs (each in its own run, as the MI355X guide
# prescribes; --pmc is never combined with sys/runtime tracing).  Summaries: profiles/summarize.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-scaffold --no-layout-accounting $*"
KEEP='Kernel_Name|"Name"|walk_|filter_score|gather_kernel|kept_fields|first_seen|make_entries|make_nodes|row_ptr_kernel|node_flags|scan_units|pack_paths|det_count|walk_layout|padded_degree|bfs_|summary_kernel|dup_flags|copy_kept'
run() {
    name=$1; shift
    timeout -k 10 ${PASS_LIMIT:-280} rocprofv3 "$@" --output-format csv -d $OUT/$name -- $B > $OUT/$name.json 2> $OUT/$name.log || echo "pass $name failed"
    # gpurun copies back at most 64 MiB: keep only this library's kernels from the per-dispatch tables
    for f in $OUT/$name/*/*_kernel_trace.csv $OUT/$name/*/*_counter_collection.csv; do
        [ -f "$f" ] && grep -E "$KEEP" "$f" > "$f.tmp" && mv "$f.tmp" "$f"
    done
    tail -c 600 $OUT/$name.log > $OUT/$name.log.tail; mv $OUT/$name.log.tail $OUT/$name.log
}
# PASSES="trace fetch write ..." restricts the run (a pass of the chr1-shaped config takes minutes)
PASSES=${PASSES:-trace fetch write sq sq2 tcc tcp}
want() { case " $PASSES " in *" $1 "*) return 0;; esac; return 1; }
want trace && run trace --kernel-trace --stats
want fetch && run fetch --pmc FETCH_SIZE --kernel-trace
want write && run write --pmc WRITE_SIZE --kernel-trace
want sq && run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace
want sq2 && run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace
want tcc && run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace
want tcp && run tcp --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace
ls $OUT
