#!/usr/bin/env python3
"""Profiling aid: launch ONE hot-path kernel a few times (config-2 shape) so that a rocprofv3 --pmc pass
sees only it.  usage: python tools/run_one_kernel.py {roi_align_fwd|roi_align_bwd|roi_pool_bwd|roi_crop_bwd|nms} [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "roi_align_fwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
if os.environ.get("MI_LIB_OVERRIDE"):      # tuning builds of the library (tools/)
    _lib.LIB_PATH = os.path.abspath(os.environ["MI_LIB_OVERRIDE"])
lib = _lib.lib()
stream = _lib.current_stream_handle(dev)
h, w, scale = syn.FPN_LEVELS[2]
c, r, res, sr = syn.FPN_DIM, 512, 7, 2
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
rois_np = syn.rois_canonical(r, 1, seed=0)
if os.environ.get("MI_BENCH_SORT_ROIS"):
    import numpy as np
    key = (rois_np[:, 2] + rois_np[:, 4]) // (2 * 64) * 4096 + (rois_np[:, 1] + rois_np[:, 3]) / 2
    rois_np = np.ascontiguousarray(rois_np[np.argsort(key, kind="stable")])
layout = 0
if os.environ.get("MI_BENCH_NHWC"):  # channels-last storage of the same logical tensor
    feat = feat.permute(0, 2, 3, 1).contiguous()
    layout = 1
rois = torch.from_numpy(rois_np).to(dev)
out = torch.empty((r, c, res, res), device=dev)
gtop = torch.randn(r, c, res, res, device=dev)
gin = torch.zeros(1, c, h, w, device=dev)
dets = torch.from_numpy(syn.boxes_clustered(2000, seed=0)).to(dev)
keep = torch.empty(2000, dtype=torch.int64, device=dev)
num = torch.empty(1, dtype=torch.int32, device=dev)
wsb = lib.mi_nms_workspace_bytes(2000)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
fws_bytes = lib.mi_roi_align_forward_workspace_bytes(r)
if which == "roi_align_bwd" and not os.environ.get("MI_BENCH_BWD_UNPLANNED"):  # room for the backward plan
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes

    fws_bytes = max(fws_bytes, _backward_workspace_bytes([(h, w)], 1, r))
fws = torch.empty(fws_bytes, dtype=torch.uint8, device=dev)
if which in ("roi_pool_bwd", "roi_crop_bwd"):
    import numpy as np
    argmax = torch.empty((r, c, res, res), dtype=torch.int32, device=dev)
    assert lib.mi_roi_pool_forward(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), argmax.data_ptr(), 1, c, h, w, r, res, res, scale, stream) == 0
    lin = np.linspace(-1, 1, res)
    cx = (rois_np[:, 1] + rois_np[:, 3]) * 0.5 * scale / (w - 1) * 2 - 1
    cy = (rois_np[:, 2] + rois_np[:, 4]) * 0.5 * scale / (h - 1) * 2 - 1
    sx = (rois_np[:, 3] - rois_np[:, 1]) * 0.5 * scale / (w - 1) * 2
    sy = (rois_np[:, 4] - rois_np[:, 2]) * 0.5 * scale / (h - 1) * 2
    gy = cy[:, None, None] + sy[:, None, None] * lin[None, :, None] + 0 * lin[None, None, :]
    gx = cx[:, None, None] + sx[:, None, None] * lin[None, None, :] + 0 * lin[None, :, None]
    grid = torch.from_numpy(np.stack([gy, gx], axis=3).astype(np.float32)).to(dev)
    crop_ws = torch.empty(lib.mi_roi_crop_backward_workspace_bytes(r), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
warm = 20 if os.environ.get("MI_BENCH_TIME") else 0       # MI_BENCH_TIME: 20 untimed calls, then `iters` timed ones
for it in range(warm + iters):
    if it == warm:
        t0.record()
    if which == "roi_align_fwd":
        rc = lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale,
                                         sr, 0, layout, fws.data_ptr(), fws.numel(), stream)
    elif which == "roi_align_bwd":
        rc = lib.mi_roi_align_backward_ws(gtop.data_ptr(), rois.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res, res,
                                          scale, sr, 0, 0, fws.data_ptr(), fws.numel(), 2, stream)
    elif which == "roi_pool_bwd":
        rc = lib.mi_roi_pool_backward(gtop.data_ptr(), rois.data_ptr(), argmax.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res, res, scale, stream)
    elif which == "roi_crop_bwd":
        rc = lib.mi_roi_crop_backward_ws(feat.data_ptr(), grid.data_ptr(), gtop.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res, res,
                                         crop_ws.data_ptr(), crop_ws.numel(), stream)
    else:
        rc = lib.mi_nms(dets.data_ptr(), 2000, 0.7, 0, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), wsb, stream)
    assert rc == 0
t1.record()
torch.cuda.synchronize()
print("done", which, iters, ("%.2f us per call" % (t0.elapsed_time(t1) * 1e3 / iters)) if warm else "")
