#!/bin/bash
# round-3 GPU call 6: GPU tier, bench line, r3 profiles with the final library
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > gpurun_out/r3/c6_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3/c6_pytest.log
tail -4 gpurun_out/r3/c6_pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r3/c6_bench.json 2> gpurun_out/r3/c6_bench.err
echo "bench rc=$?"; tail -c 300 gpurun_out/r3/c6_bench.err
timeout 900 bash tools/collect_profiles_r3.sh > gpurun_out/r3/c6_profiles.log 2>&1; tail -4 gpurun_out/r3/c6_profiles.log
