cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_reentrancy_gpu.py -q -x -k "roi_pool or pool" 2>&1 | tail -3
python tools/pool_bwd_c4.py 256; python tools/pool_bwd_c4.py 1024
python tools/pool_crop_time.py 100 | cut -c1-400
