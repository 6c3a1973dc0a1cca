#!/bin/bash
# NHWC forward tuning iteration: NHWC parity tests (all, no -x; plain and with PYTEST_ENV), then bench --only-roofline
# per "ENV=VAL ..." argument.  usage: PYTEST_ENV="A=1 B=2" bash tools/gpu_nhwc_iter.sh TAG "cfg" ...
TAG=${1:-n}; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "nhwc or channels_last" --timeout 90 > $R/gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> $R/gpurun_out/${TAG}_pytest.log
if [ -n "$PYTEST_ENV" ]; then
  env $PYTEST_ENV timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "nhwc or channels_last" --timeout 90 >> $R/gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest[$PYTEST_ENV] exit $?" >> $R/gpurun_out/${TAG}_pytest.log
fi
: > $R/gpurun_out/${TAG}_exp.log
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --only-roofline 2>&1 | grep -v amdgpu.ids >> $R/gpurun_out/${TAG}_exp.log
done
grep -E "passed|failed|Error|assert|exit" $R/gpurun_out/${TAG}_pytest.log | tail -30
python - <<PY
import json
for line in open("$R/gpurun_out/${TAG}_exp.log"):
    try:
        d = json.loads(line)
    except Exception:
        print(line[:200]); continue
    print("%-90s %7.2f us  frac %.3f" % (d["env"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
