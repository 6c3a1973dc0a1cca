#!/bin/bash
# kernel durations of the backward cases of tools/bwd_time.py, one rocprofv3 run per (case, slice length)
for c in ${CASES:-box cfg2}; do for s in ${SLICES:-0 32}; do
  echo "=== case $c slice $s"
  CASES=$c SLICES=$s bash tools/gpu_prof_script.sh bwd_${c}_$s python tools/bwd_time.py 30 | grep -v "fwd\|prepare"
done; done
