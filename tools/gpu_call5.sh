#!/bin/bash
# round-3 GPU call 5: default = CU masks + two scan streams (post stream unmasked); GPU tier, pipe sweep, bench
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r3/c5_pipe.log; : > $L
for rows in 1000000 1250000 10000000; do
 for v in "-1 -1" "0 0"; do
  set -- $v
  echo "== rows $rows B 64 pipe_cu_mask $1 pipe_dual_scan $2 (-1 = default)" >> $L
  CMR_PIPE_CU_MASK=$1 CMR_PIPE_DUAL_SCAN=$2 timeout 120 python tools/pipe_only.py $rows 64 200 2>&1 | grep -v amdgpu >> $L
 done
done
for rows in 1250000 10000000; do
  echo "== rows $rows B 256 default" >> $L
  timeout 120 python tools/pipe_only.py $rows 256 60 2>&1 | grep -v amdgpu >> $L
done
cat $L
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > gpurun_out/r3/c5_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3/c5_pytest.log
tail -5 gpurun_out/r3/c5_pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r3/c5_bench.json 2> gpurun_out/r3/c5_bench.err
echo "bench rc=$?"; tail -c 300 gpurun_out/r3/c5_bench.err
