#!/bin/bash
# Builds the A/B arms of tools/roi_align_ab.py / tools/bwd_ab.py into .ab_r5/ (git-ignored, travels with gpurun):
#   libmi_head.so    the library of the last commit (git archive HEAD)
#   libmi_tuning.so  the working tree with -DMI_TUNING=1 (ablation switches, timeline stamps)
# and the release library of the working tree in place.   usage: bash tools/build_variants.sh [head] [tuning]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $R/.ab_r5
python -c "import sys; sys.path.insert(0,'$R'); from detectron_pytorch_amd import build; build.build(force=False)"
for what in "$@"; do
  T=$(mktemp -d)
  if [ $what = head ]; then (cd $R && git archive HEAD detectron_pytorch_amd include) | tar -x -C $T; else cp -r $R/detectron_pytorch_amd $R/include $T/; fi
  (cd $T && MI_TUNING_BUILD=$([ $what = tuning ] && echo 1) python -c "import sys; sys.path.insert(0,'.'); from detectron_pytorch_amd import build; build.build(force=True)")
  cp $T/detectron_pytorch_amd/libmi_detectron_ops.so $R/.ab_r5/libmi_$what.so; rm -rf $T
done
ls -la $R/.ab_r5
