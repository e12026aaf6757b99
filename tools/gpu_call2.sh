#!/bin/bash
# round-3 GPU call 2: GPU tier, wide kernel after the wait fix, bench line, r3 profiles, MFMA probe in both operand modes
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > gpurun_out/r3/c2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3/c2_pytest.log
tail -5 gpurun_out/r3/c2_pytest.log
timeout 200 python tools/wide_bench.py 10000000 256 > gpurun_out/r3/c2_wide.log 2>&1; cut -c1-330 gpurun_out/r3/c2_wide.log
( tools/probe/mfma_probe unit; tools/probe/mfma_probe ) > gpurun_out/r3/c2_mfma_probe.txt 2>&1; grep -c TFLOP gpurun_out/r3/c2_mfma_probe.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r3/c2_bench.json 2> gpurun_out/r3/c2_bench.err
echo "bench rc=$?"; tail -c 400 gpurun_out/r3/c2_bench.err
timeout 900 bash tools/collect_profiles_r3.sh > gpurun_out/r3/c2_profiles.log 2>&1; tail -25 gpurun_out/r3/c2_profiles.log
