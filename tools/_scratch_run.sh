cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "roi_crop or roi_pool" 2>&1 | tail -3
timeout 300 python tools/pool_crop_time.py 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v['us'] for k,v in d.items() if isinstance(v,dict)})"
echo -n "ring4 "; MI_LIB_OVERRIDE=.ab_r6/libmi_ring4.so timeout 300 python tools/pool_crop_time.py 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roi_pool_bwd']['us'])"
