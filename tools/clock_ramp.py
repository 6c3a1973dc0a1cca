"""Tuning aid: per-call time of the config-2 RoIAlign forward in consecutive batches of 200 calls after 0.5 s of idle --
the chip needs ~30 ms of sustained work before its clocks have ramped (records-free kernel, record-driven pair, again)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from detectron_pytorch_amd import _lib, synthetic as syn
dev = torch.device("cuda", 0)
lib = _lib.lib(); stream = _lib.current_stream_handle(dev)
h, w, scale = syn.FPN_LEVELS[2]; c, r = 256, 512
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, 7, 7), device=dev)
ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(r), dtype=torch.uint8, device=dev)
def call():
    assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, 7, 7, scale, 2, 0, 0, ws.data_ptr(), ws.numel(), stream) == 0
for slab in ("1", "0", "1"):
    os.environ["MI_ROI_ALIGN_SLAB"] = slab; lib.mi_dbg_reload_tuning()
    torch.cuda.synchronize(); time.sleep(0.5)
    res = []
    for b in range(16):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): call()
        e.record(); e.synchronize()
        res.append(round(a.elapsed_time(e) * 5, 2))
    print("slab", slab, res)
