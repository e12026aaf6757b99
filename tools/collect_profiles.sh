#!/bin/bash
# Profile collection on the GPU box (gpurun; ROUND=r4 by default names the outputs; bench.py runs with --no-pmc here: its own
# rocprofv3 passes must not nest inside these): kernel trace of bench.py, PMC traffic passes of the headline
# configuration, SQ counters + traffic of the wide kernel.  Everything lands in gpurun_out/<round>_profiles/: the summaries AND a
# compressed per-dispatch CSV of every database (tools/rocpd_dump.py), which is what gets copied into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r6}
O=$R/gpurun_out/${ROUND}_profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o w -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_under_trace.json 2> $O/kt.err
python $R/tools/rocpd_stats.py $O/kt/w_results.db > $O/${ROUND}_bench_kernel_stats.txt
python $R/tools/rocpd_dump.py $O/kt/w_results.db $O/${ROUND}_raw_bench_kernel_trace.csv.gz
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o w -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-pmc > /dev/null 2> $O/pmc_$c.err
  python $R/tools/rocpd_pmc.py $O/pmc_$c/w_results.db scan_kernel 1000 > $O/pmc_$c.jsonl
  python $R/tools/rocpd_dump.py $O/pmc_$c/w_results.db $O/${ROUND}_raw_pmc_$c.csv.gz
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace -d $O/pmc_wide_a -o w -- python $R/tools/pipe_only.py 10000000 256 12 > /dev/null 2> $O/pmc_wide_a.err
python $R/tools/rocpd_pmc.py $O/pmc_wide_a/w_results.db scan_wide 1000 > $O/pmc_wide_a.jsonl
python $R/tools/rocpd_dump.py $O/pmc_wide_a/w_results.db $O/${ROUND}_raw_pmc_wide_a.csv.gz
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace -d $O/pmc_wide_b -o w -- python $R/tools/pipe_only.py 10000000 256 12 > /dev/null 2> $O/pmc_wide_b.err
python $R/tools/rocpd_pmc.py $O/pmc_wide_b/w_results.db scan_wide 1000 > $O/pmc_wide_b.jsonl
python $R/tools/rocpd_dump.py $O/pmc_wide_b/w_results.db $O/${ROUND}_raw_pmc_wide_b.csv.gz
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_wide_$c -o w -- python $R/tools/pipe_only.py 10000000 256 12 > /dev/null 2> $O/pmc_wide_$c.err
  python $R/tools/rocpd_pmc.py $O/pmc_wide_$c/w_results.db scan_wide 1000 > $O/pmc_wide_$c.jsonl
  python $R/tools/rocpd_dump.py $O/pmc_wide_$c/w_results.db $O/${ROUND}_raw_pmc_wide_$c.csv.gz
done
ROUND=$ROUND python $R/tools/summarise_profiles.py $O > $O/summarise.log 2>&1
rm -rf $O/*/w_results.db.tmp $O/*/*.db; du -sh $O; ls $O; cat $O/summarise.log
