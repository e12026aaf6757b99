cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_multi_device_gpu.py -q -x 2>&1 | tail -3
for S in 1 2 4 8; do timeout 200 python bench.py --single-process --gpus $S --share-device --rows 4000000 --steps 60 --no-extra --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('S', d['n_gpus'], 'ms_per_step', d['ms_per_step'], 'value', d['value'], json.dumps(d.get('config'))[:600])
"; done
cd /tmp && export TMPDIR=/tmp
for ns in 0 1; do
CMR_SCAN_NO_SAMPLE=$ns rocprofv3 --kernel-trace -d /tmp/lt$ns -o w -- python $GRAFT_REPO_ROOT/tools/latency_trace.py 1000000 2000000 2>/dev/null | grep rows
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py /tmp/lt$ns/w_results.db 0 10
done
