#!/bin/bash
# Builds the A/B arms of tools/roi_align_ab.py into $AB (default .ab_r6/; git-ignored, travels with gpurun):
#   libmi_head.so    the library of the last commit (git archive HEAD); libmi_<rev>.so for any other word: that revision
#   libmi_tuning.so  the working tree with -DMI_TUNING=1 (ablation switches, timeline stamps)
# and the release library of the working tree in place.   usage: bash tools/build_variants.sh [head] [tuning]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); AB=${AB:-$R/.ab_r6}; mkdir -p $AB
python -c "import sys; sys.path.insert(0,'$R'); from detectron_pytorch_amd import build; build.build(force=False)"
for what in "$@"; do
  T=$(mktemp -d)
  if [ $what = tuning ]; then cp -r $R/detectron_pytorch_amd $R/include $T/; else rev=$what; [ $what = head ] && rev=HEAD; (cd $R && git archive $rev detectron_pytorch_amd include) | tar -x -C $T; fi
  (cd $T && MI_TUNING_BUILD=$([ $what = tuning ] && echo 1) python -c "import sys; sys.path.insert(0,'.'); from detectron_pytorch_amd import build; build.build(force=True)")
  cp $T/detectron_pytorch_amd/libmi_detectron_ops.so $AB/libmi_$what.so; rm -rf $T
done
ls -la $AB
