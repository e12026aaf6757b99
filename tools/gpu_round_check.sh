#!/bin/bash
# What a round ends with on the GPU box (gpurun -- bash tools/gpu_round_check.sh): GPU tier, bench line, profile collection
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > gpurun_out/r3/check_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3/check_pytest.log
tail -4 gpurun_out/r3/check_pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r3/check_bench.json 2> gpurun_out/r3/check_bench.err
echo "bench rc=$?"; tail -c 300 gpurun_out/r3/check_bench.err
timeout 900 bash tools/collect_profiles.sh > gpurun_out/r3/check_profiles.log 2>&1; tail -4 gpurun_out/r3/check_profiles.log
