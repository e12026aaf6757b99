#!/bin/bash
# round-3 GPU call 4: pipelined step with explicit CU masks and two alternating scan streams
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r3/c4_pipe.log; : > $L
for rows in 1000000 1250000 10000000; do
 for v in "0 0" "1 0" "2 0" "1 1" "2 1" "0 1"; do
  set -- $v
  echo "== rows $rows mask $1 dual $2" >> $L
  CMR_PIPE_CU_MASK=$1 CMR_PIPE_DUAL_SCAN=$2 timeout 120 python tools/pipe_only.py $rows 64 200 2>&1 | grep -v amdgpu >> $L
 done
done
CMR_PIPE_CU_MASK=1 CMR_PIPE_DUAL_SCAN=1 timeout 120 python tools/pipe_only.py 1250000 256 100 2>&1 | grep -v amdgpu >> $L
CMR_PIPE_CU_MASK=0 CMR_PIPE_DUAL_SCAN=0 timeout 120 python tools/pipe_only.py 1250000 256 100 2>&1 | grep -v amdgpu >> $L
cat $L
