#!/bin/bash
# PMC passes over the wide-batch routes (tools/quad_ab.py): fabric-side fetch bytes and L2 hit / miss per scan launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/quad_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ROWS=${ROWS:-2500000}
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace -d $O/$tag -o w -- python $R/tools/quad_ab.py $ROWS 256 4 > $O/$tag.out 2> $O/$tag.err
  python $R/tools/rocpd_pmc.py $O/$tag/w_results.db scan_ 100 > $O/$tag.jsonl
done
rm -rf $O/*/*.db $O/*/w_results.db.tmp
cat $O/*.jsonl
