#!/bin/bash
# round-3 GPU call 1: full GPU tier, bench line, wide-kernel ring geometry A/B
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > gpurun_out/r3/c1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3/c1_pytest.log
tail -5 gpurun_out/r3/c1_pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r3/c1_bench.json 2> gpurun_out/r3/c1_bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r3/c1_bench.err
for lib in default g8n16 n9 g8n18; do
  if [ $lib = default ]; then unset COMORAG_HIP_LIB; else export COMORAG_HIP_LIB=$GRAFT_REPO_ROOT/build_exp/lib_$lib.so; fi
  echo "== wide $lib" >> gpurun_out/r3/c1_wide.log
  timeout 200 python tools/wide_bench.py 10000000 256 >> gpurun_out/r3/c1_wide.log 2>&1
done
unset COMORAG_HIP_LIB
cat gpurun_out/r3/c1_wide.log | grep -v "^$" | cut -c1-330
