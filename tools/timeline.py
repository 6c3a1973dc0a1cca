#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase timeline of the RoIAlign forward tile kernel (config-2 shape).
The kernel stamps clock64() -- shader-clock ticks (~2.1 GHz), a counter whose base differs from compute unit to compute
unit, so only differences inside one workgroup mean anything -- at 8 points per workgroup into a device buffer
(mi_dbg_roi_align_timeline); this prints the per-phase durations in ticks.  usage: python tools/timeline.py [sorted]"""
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
lib.mi_dbg_roi_align_timeline.restype = None
lib.mi_dbg_roi_align_timeline.argtypes = [ctypes.c_void_p]
stream = _lib.current_stream_handle(dev)
h, w, scale = syn.FPN_LEVELS[2]
c, r, res, sr = syn.FPN_DIM, 512, 7, 2
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
nwg = r * (c // 32)
tl = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)


def launch():
    rc = lib.mi_roi_align_forward(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                  0, 0, stream)
    assert rc == 0


for _ in range(5):
    launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(tl.data_ptr())
launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(None)
t = tl.cpu().numpy()
meta = t[:, 7] * 0
t[:, 7] &= (1 << 44) - 1
t[:, :7] &= (1 << 44) - 1
t = t - t[:, :1]
names = ["roi+geom", "dma issue", "tables", "wait dma+barrier", "compute", "barrier", "store"]
print("workgroups", nwg, "(ticks of the shader clock, ~2100 per us)")
for k in range(7):
    d = t[:, k + 1] - t[:, k]
    print("%-18s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
tot = t[:, 7] - t[:, 0]
print("%-18s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % ("total", tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
xcc = meta & 0xf
print("xcc histogram", np.bincount(xcc.astype(int), minlength=8))
