#!/bin/bash
# round-3 GPU call 3: 8-wave wide kernel (two waves per SIMD, 8 k-steps of a tile in LDS) — parity tests + timing; PPR routes
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python tools/ppr_bench.py > gpurun_out/r3/c3_ppr.log 2>&1; cat gpurun_out/r3/c3_ppr.log | grep -v amdgpu.ids
export COMORAG_HIP_LIB=$GRAFT_REPO_ROOT/build_exp/lib_w8.so
( timeout 600 python -m pytest tests/test_search_gpu.py tests/test_configs_gpu.py -m gpu -q -k "wide or batch or config3 or threads" -p no:cacheprovider ) > gpurun_out/r3/c3_w8_pytest.log 2>&1; tail -4 gpurun_out/r3/c3_w8_pytest.log
timeout 200 python tools/wide_bench.py 10000000 256 > gpurun_out/r3/c3_w8_wide.log 2>&1; cut -c1-330 gpurun_out/r3/c3_w8_wide.log | grep -v amdgpu.ids
timeout 100 python tools/wide_bench.py 1250000 256 >> gpurun_out/r3/c3_w8_wide.log 2>&1; tail -2 gpurun_out/r3/c3_w8_wide.log | cut -c1-330
unset COMORAG_HIP_LIB
timeout 100 python tools/wide_bench.py 1250000 256 > gpurun_out/r3/c3_w4_wide_shard.log 2>&1; tail -2 gpurun_out/r3/c3_w4_wide_shard.log | cut -c1-330
