cd /root/repo
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -q -x -k "soft" 2>&1 | tail -3
timeout 300 python - <<'PY'
import json, torch, sys
sys.path.insert(0,'.')
from tools import hot_path_bench as hp
d=hp.nms_latency(torch.device('cuda',0), 50)
print(json.dumps({k:v for k,v in d.items() if 'soft' in k or 'postprocess' in k}))
PY
