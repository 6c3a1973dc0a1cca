#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_probe; mkdir -p $O; cd $R
for p in full_step fwd_bwd fwd_only body_fwd proposals labelling optimizer body_fwd_bwd; do
  timeout 240 python tools/capture_probe.py $p > $O/cap_$p.log 2>&1; grep -h "CAPTURE_" $O/cap_$p.log | head -2
  if grep -q CAPTURE_OK $O/cap_full_step.log; then break; fi
done
timeout 600 python tools/e2e_probe.py --variants fp32_bench,fp32_cl_bench,bf16,bf16_cl_bench --modes train --steps 5 --warmup 2 > $O/probe_variants.log 2>&1
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 300 python tools/e2e_probe.py --variants fp32,fp32_cl --modes train --steps 5 --warmup 2 > $O/probe_nowino.log 2>&1
grep -h '"variant"' $O/probe_variants.log $O/probe_nowino.log | cut -c1-260
