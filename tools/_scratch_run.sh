cd /root/repo
timeout 900 python -m pytest tests/test_reentrancy_gpu.py -q -x 2>&1 | tail -5
