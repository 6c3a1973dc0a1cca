#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_c; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q > $O/pytest_e2e.log 2>&1; tail -25 $O/pytest_e2e.log
(timeout 600 python bench.py 2> $O/bench.err | grep '^{' | tail -1 > $O/bench_line.json); tail -3 $O/bench.err; cut -c1-1200 $O/bench_line.json; echo
(timeout 300 python bench.py --launch graph --no-extras --no-cpu-baseline 2> $O/bench_graph.err | grep '^{' | tail -1 > $O/bench_graph.json); tail -3 $O/bench_graph.err; cut -c1-300 $O/bench_graph.json; echo
(MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 300 python bench.py --no-extras --no-cpu-baseline 2> $O/bench_nowino.err | grep '^{' | tail -1 > $O/bench_nowino.json); cut -c1-300 $O/bench_nowino.json; echo
