#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase timeline of roi_align_fwd_tiles (config-2 shape).  Lane 0 of every workgroup stamps
s_memtime (100 MHz) at 7 points (mi_dbg_roi_align_timeline); slot 7 holds the number of hits of the tile."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
stream = _lib.current_stream_handle(dev)
h, w, scale = syn.FPN_LEVELS[2]
c, r, res, sr = syn.FPN_DIM, 512, 7, 2
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
nwg = 16384
tl = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)


import ctypes  # noqa: E402

lvt = _lib.FpnLevels()
lvt.num_levels, lvt.height[0], lvt.width[0] = 1, h, w
wsb = lib.mi_roi_align_forward_tiles_workspace_bytes(ctypes.byref(lvt), 1, res, res, sr)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
use_ws = bool(os.environ.get("WS"))
print("path:", "tile descriptors (two launches)" if use_ws else "one launch, no scratch")


def launch():
    if use_ws:
        assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale,
                                           sr, 0, 0, ws.data_ptr(), wsb, stream) == 0
    else:
        assert lib.mi_roi_align_forward(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                        0, 0, stream) == 0


for _ in range(5):
    launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(tl.data_ptr())
launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(None)
t = tl.cpu().numpy()
t = t[t[:, 0] != 0]
print("workgroups stamped", len(t), " hits per tile: mean %.1f max %d" % (t[:, 7].mean(), t[:, 7].max()))
hits = t[:, 7].copy()
busy = t[:, 4] != 0
t0 = t[:, 0].min()
for k in (1, 2, 3, 5):  # stamps a path does not take: carry the previous one
    z = t[:, k] == 0
    t[z, k] = t[z, k - 1]
names = ["scan (all waves)", "spin + tables", "tail of build", "barrier (image landed)", "units (batch 0)", "rest (slow, more batches)"]
if os.environ.get("STREAM"):  # persistent kernel: stamps around the third item of every workgroup
    names = ["decode + issue of next item", "compute", "wait for next item's DMA", "barrier", "-", "-"]
tb = t[busy]
tb = t
for k in range(6):
    d = (tb[:, k + 1] - tb[:, k]) * 0.01
    print("%-28s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
tot = (t[:, 6] - t[:, 0]) * 0.01
print("%-28s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % ("workgroup life", tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
end = (t[:, 6].max() - t0) * 0.01
print("kernel span %.2f us; start times p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (
    end, *(np.percentile(t[:, 0] - t0, [10, 50, 90, 100]) * 0.01)))
print("avg workgroups alive: %.1f of %d" % (tot.sum() / end, len(t)))
