set -u
T=$(mktemp -d)
export TLM_RCCL_LIB=tests/fake_rccl/libfake_rccl.so FAKE_RCCL_TIMEOUT_S=90 FAKE_RCCL_ASYNC=1 FAKE_RCCL_ASYNC_VERBOSE=1 TLM_TEST_PARTITION=contiguous
python tests/exchange_worker.py hub 0 2 $T/id $T > $T/r0.log 2>&1 &
P0=$!
python tests/exchange_worker.py hub 1 2 $T/id $T > $T/r1.log 2>&1 &
P1=$!
wait $P0; echo "rank0 rc=$?"; wait $P1; echo "rank1 rc=$?"
echo "--- rank0"; grep -v amdgpu.ids $T/r0.log | tail -30; echo "--- rank1"; grep -v amdgpu.ids $T/r1.log | tail -30
rm -rf $T
