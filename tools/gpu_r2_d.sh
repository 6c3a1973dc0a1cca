#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_d; mkdir -p $O; cd $R
bash tools/gpu_r2_perf.sh
timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "training_post_conv" > $O/pytest_e2e.log 2>&1; tail -4 $O/pytest_e2e.log
REPLAYS=40 timeout 200 python tools/capture_probe.py full_step > $O/cap_replays.log 2>&1; tail -12 $O/cap_replays.log
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
