cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "roi_pool" 2>&1 | tail -5
timeout 300 python tools/pool_crop_time.py 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v['us'] for k,v in d.items() if isinstance(v,dict)})"
for a in 2 4 1; do
  echo -n "ablate $a "; MI_LIB_OVERRIDE=.ab_r6/libmi_tuning.so MI_ROI_ALIGN_ABLATE=$a timeout 300 python tools/pool_crop_time.py 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roi_pool_bwd']['us'])"
done
