#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_j; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "post_conv" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 400 python tools/e2e_probe.py --variants fp32_cl,fp32 --modes train,infer --steps 8 --warmup 3 > $O/probe_nowino.log 2>&1
timeout 300 python tools/e2e_probe.py --variants fp32_cl,fp32 --modes train,infer --steps 8 --warmup 3 > $O/probe_wino.log 2>&1
grep -h '"variant"' $O/probe_nowino.log $O/probe_wino.log | cut -c1-170
