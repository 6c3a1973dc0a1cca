cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "roi_pool" 2>&1 | tail -3
for v in "" .ab_r6/libmi_narrow0.so; do
echo -n "lib=$v "; MI_BENCH_C4=1 MI_LIB_OVERRIDE=$v timeout 300 python tools/pool_crop_time.py 100 2>&1 | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['roi_pool_fwd']['us'] if 'roi_pool_fwd' in d else d, end=' ')
print()"
done
