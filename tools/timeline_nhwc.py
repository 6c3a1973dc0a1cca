#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase timeline of roi_align_fwd_nhwc (config-2 shape, channels-last features).
Wave 0 of every workgroup stamps s_memtime (100 MHz) at 6 points (mi_dbg_roi_align_timeline)."""
import ctypes
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
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev).permute(0, 2, 3, 1).contiguous()
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(r), dtype=torch.uint8, device=dev)
nwg = r * 4
tl = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)


def launch():
    assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                       0, 1, ws.data_ptr(), ws.numel(), stream) == 0


for _ in range(5):
    launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(tl.data_ptr())
launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(None)
t = tl.cpu().numpy()
t = t[t[:, 0] != 0]
print("workgroups stamped", len(t))
t = t[:, :6] - t[:, 0].min()
names = ["record fetch", "taps (wave 0)", "tile write", "barrier (other waves)", "copy-out"]
for k in range(5):
    d = (t[:, k + 1] - t[:, k]) * 0.01
    print("%-22s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
tot = (t[:, 5] - t[:, 0]) * 0.01
print("%-22s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % ("workgroup life", tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
print("kernel span %.2f us; start times p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (
    (t[:, 5].max()) * 0.01, *(np.percentile(t[:, 0], [10, 50, 90, 100]) * 0.01)))
print("avg workgroups alive: %.1f" % (tot.sum() / (t[:, 5].max() * 0.01)))
