#!/usr/bin/env bash
# run.sh -- the reference's driver contract (reference run.sh:1-47) for the MI355X host program.
#
#   ./run.sh <dataset-dir>
#
#   <dataset-dir>
#          L  reads.fasta | reads.fastq
#          L  contigs.fasta | contigs.fastq
#          L  rr_overlaps.paf      (read-read overlaps;   minimap2 --dual=yes -x ava-pb reads reads)
#          L  cr_overlaps.paf      (contig-read overlaps; minimap2 --dual=yes -x ava-pb contigs reads)
#
# As in the reference, the overlap files are built with ./minimap2/minimap2 only when BOTH are missing (run.sh:38-45),
# otherwise the existing ones are used; the scaffold goes to <dataset-dir>/HERA_result.fasta.
# Extra (not in the reference): TLM_GPUS=N runs one process per GPU (anchors sharded, one RCCL exchange of the paths);
# further arguments are handed to the program (e.g. --rb-att 1000).
here="$(cd "$(dirname "$0")" && pwd)"
bin="$here/telomeri_amd/bin/telomeri_hip"

if [ $# -lt 1 ]
then
    echo "Please provide the dataset directory path!"
    echo "Usage: ./run.sh <dataset-dir>"
    echo "Dataset structure must match the following"
    echo "<dataset-dir>"
    echo "       L  reads.fasta"
    echo "       L  contigs.fasta"
    echo "       L  rr_overlaps.paf"
    echo "       L  cr_overlaps.paf"
    exit 1
fi

dir=$1
shift
out_file=$dir/"HERA_result.fasta"
reads_file=$(ls $dir/reads.fast? 2> /dev/null | head -n 1)
contigs_file=$(ls $dir/contigs.fast? 2> /dev/null | head -n 1)
rr_file=$dir/"rr_overlaps.paf"
cr_file=$dir/"cr_overlaps.paf"

if [ ! -f "$rr_file" -a ! -f "$cr_file" ]
then
    if [ ! -x "$here/minimap2/minimap2" ]
    then
        echo "No overlap files in $dir and no ./minimap2/minimap2 to build them with." >&2
        exit 1
    fi
    "$here/minimap2/minimap2" --dual=yes -x ava-pb "$reads_file" "$reads_file" > "$rr_file"
    "$here/minimap2/minimap2" --dual=yes -x ava-pb "$contigs_file" "$reads_file" > "$cr_file"
else
    echo "Using existing overlap files!"
fi

if [ ! -x "$bin" ]
then
    make -s -C "$here/telomeri_amd/csrc" && make -s -C "$here/telomeri_amd/host" || exit 1
fi

gpus=${TLM_GPUS:-1}
if [ "$gpus" -le 1 ]
then
    exec "$bin" "$@" "$rr_file" "$cr_file" "$reads_file" "$contigs_file" "$out_file"
fi

# one process per GPU; rank 0 writes the scaffold and its log goes to the terminal.  TLM_DEVICES="d0 d1 ..." names the device
# of every rank (default: rank r on device r).  The ranks meet in collectives: when one of them ends with an error the
# others are ended too instead of being left waiting for it.
rendezvous=$(mktemp -u "${TMPDIR:-/tmp}/telomeri_rccl_id.XXXXXX")
devices=(${TLM_DEVICES:-$(seq 0 $((gpus - 1)))})
if [ ${#devices[@]} -lt "$gpus" ]
then
    echo "TLM_DEVICES names ${#devices[@]} devices for $gpus ranks" >&2
    exit 1
fi
pids=()
for r in $(seq 1 $((gpus - 1)))
do
    TLM_WORLD=$gpus TLM_RANK=$r TLM_DEVICE=${devices[$r]} TLM_RENDEZVOUS=$rendezvous \
        "$bin" "$@" "$rr_file" "$cr_file" "$reads_file" "$contigs_file" "$out_file" > "$dir/telomeri_rank$r.log" 2>&1 &
    pids+=($!)
done
TLM_WORLD=$gpus TLM_RANK=0 TLM_DEVICE=${devices[0]} TLM_RENDEZVOUS=$rendezvous \
    "$bin" "$@" "$rr_file" "$cr_file" "$reads_file" "$contigs_file" "$out_file" &
pids+=($!)
rc=0
left=$gpus
while [ $left -gt 0 ]
do
    wait -n
    st=$?
    left=$((left - 1))
    if [ $st -ne 0 ]
    then
        rc=$st
        echo "run.sh: a rank ended with status $st; ending the other ranks" >&2
        kill "${pids[@]}" 2> /dev/null
        break
    fi
done
wait
rm -f "$rendezvous" "$rendezvous.tmp"
exit $rc
