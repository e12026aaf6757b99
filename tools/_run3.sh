cd $GRAFT_REPO_ROOT
timeout 200 python tools/fin_debug.py 1000003 64 8 2>&1 | grep -v "^fin grid" | grep "rep 2"
timeout 900 python -m pytest tests/test_scan_fin_gpu.py -q -x 2>&1 | tail -5
