cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "legacy or direct or roi_align_golden or seven or modules or misaligned or adversarial or channels_last" 2>&1 | tail -3
MI_ROI_ALIGN_IMPL=direct timeout 300 python tools/direct_time.py 2>&1 | tail -1
