#!/bin/bash
# Same-box A/B of the pipelined streams in bench.py's own flow: headline (10 M rows, B = 64) and config 3 (B = 256) per setting.
for opt in "" "--index-option pipe_cu_mask=0" "--index-option pipe_cu_mask=1" ; do
  echo "== $opt"
  timeout 200 python bench.py --steps 20 --warmup 5 --only-config3 $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['extra']['config3_batch256']
print('headline', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], '| c3', round(c['value']), c['ms_per_step'], c['kernel_ms'])"
done
for rows in 1000000 1250000; do for o in "" "CMR_PIPE_CU_MASK=0"; do echo "== rows $rows $o"; env $o timeout 100 python tools/pipe_only.py $rows 64 200 | tail -1; done; done
