#!/usr/bin/env python3
"""Tuning aid: per-iteration phase timeline of the persistent RoIAlign forward kernel (config-2 shape)."""
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
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(r), dtype=torch.uint8, device=dev)
nwg = 768
tl = torch.zeros((nwg, 8, 8), dtype=torch.int64, device=dev)


def launch():
    rc = lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                     0, 0, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0


for _ in range(5):
    launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(tl.data_ptr())
launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(None)
t = tl.cpu().numpy()
valid = t[:, :, 7] > 0
names = ["wait_vm", "B1", "prefetch next (loads)", "compute", "B2", "issue next dma", "store", ]
print("iterations recorded per WG: mean %.2f" % valid.sum(1).mean())
for k in range(7):
    d = (t[:, :, k + 1] - t[:, :, k])[valid]
    print("%-24s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
tot = (t[:, :, 7] - t[:, :, 0])[valid]
print("%-24s mean %8.0f  p50 %8.0f  p90 %8.0f" % ("iteration (0->7)", tot.mean(), np.median(tot), np.percentile(tot, 90)))
gap = (t[:, 1:, 0] - t[:, :-1, 7])[valid[:, 1:] & valid[:, :-1]]
print("%-24s mean %8.0f  p50 %8.0f  p90 %8.0f" % ("gap 7->next 0 (ticket)", gap.mean(), np.median(gap), np.percentile(gap, 90)))
life = t[:, :, 7].max(1) - np.where(valid, t[:, :, 0], 1 << 62).min(1)
print("WG lifetime (first stamp -> last): mean %.0f max %.0f" % (life[valid.any(1)].mean(), life[valid.any(1)].max()))
