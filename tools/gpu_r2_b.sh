#!/bin/bash
# round-2 GPU pass B: capture probe of the full step after the select() fix, the new GPU tests, bench (eager + graph),
# e2e variants without MIOpen benchmark mode
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > $O/pytest_e2e.log 2>&1; tail -15 $O/pytest_e2e.log
for p in full_step; do timeout 200 python tools/capture_probe.py $p > $O/cap_$p.log 2>&1; grep -h "CAPTURE_" $O/cap_$p.log | head -2; done
if ! grep -q CAPTURE_OK $O/cap_full_step.log; then
  for p in proposals fwd_only fwd_bwd; do timeout 200 python tools/capture_probe.py $p > $O/cap_$p.log 2>&1; grep -h "CAPTURE_" $O/cap_$p.log | head -2; tail -12 $O/cap_$p.log; done
fi
(timeout 600 python bench.py 2> $O/bench.err | grep '^{' | tail -1 > $O/bench_line.json); tail -3 $O/bench.err; cut -c1-900 $O/bench_line.json; echo
(timeout 300 python bench.py --launch graph --no-extras --no-cpu-baseline 2> $O/bench_graph.err | grep '^{' | tail -1 > $O/bench_graph.json); tail -3 $O/bench_graph.err; cut -c1-400 $O/bench_graph.json; echo
timeout 400 python tools/e2e_probe.py --variants bf16,bf16_cl,fp32_cl --modes train,infer --steps 5 --warmup 2 > $O/probe_variants.log 2>&1
grep -h '"variant"' $O/probe_variants.log | cut -c1-200
