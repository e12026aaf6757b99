cd $GRAFT_REPO_ROOT
timeout 200 python tools/fin_debug.py 200000 128 1 2>&1 | grep -v "^fin grid"
timeout 900 python -m pytest tests/test_scan_fin_gpu.py tests/test_search_gpu.py -q -x 2>&1 | tail -8
for f in "CMR_SCAN_FIN=1" "CMR_SCAN_FIN=0" "CMR_SCAN_FIN=0 CMR_SCAN_ASM_RING=0"; do echo "== $f"; env $f timeout 300 python tools/latency.py 1000000 2000000 2>&1 | grep rows; done
