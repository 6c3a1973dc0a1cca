#!/bin/bash
# Builds the working tree with extra -D defines into $AB/libmi_<name>.so (an A/B arm for MI_LIB_OVERRIDE).
#   usage: bash tools/build_defines.sh <name> "<DEFINE=VALUE ...>" [tuning]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); AB=${AB:-$R/.ab_r6}; mkdir -p $AB
T=$(mktemp -d); cp -r $R/detectron_pytorch_amd $R/include $T/
(cd $T && MI_EXTRA_DEFINES="$2" MI_TUNING_BUILD=$([ "$3" = tuning ] && echo 1) python -c "import sys; sys.path.insert(0,'.'); from detectron_pytorch_amd import build; build.build(force=True, verbose=False)")
cp $T/detectron_pytorch_amd/libmi_detectron_ops.so $AB/libmi_$1.so; rm -rf $T; ls -la $AB/libmi_$1.so
