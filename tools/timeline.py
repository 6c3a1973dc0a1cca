#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase timeline of the RoIAlign forward tile kernel (config-2 shape).
The kernel stamps s_memtime at 8 points per workgroup into a device buffer (mi_dbg_roi_align_timeline);
this prints per-phase durations and the concurrency picture.  usage: python tools/timeline.py [sorted]"""
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
t0 = t[:, 0].min()
t = t - t0
names = ["roi+geom", "dma issue", "tables", "wait dma+barrier", "compute", "barrier", "store"]
print("workgroups", nwg, "kernel span (cycles)", t[:, 7].max(), "= us @100MHz-const?", )
for k in range(7):
    d = t[:, k + 1] - t[:, k]
    print("%-18s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
tot = t[:, 7] - t[:, 0]
print("%-18s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % ("total", tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
start = np.sort(t[:, 0])
print("start times: p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(start, [10, 50, 90, 100])))
# concurrency: average number of workgroups alive
span = t[:, 7].max()
print("avg WGs alive: %.1f (of %d slots at 3/CU)" % (tot.sum() / span, 768))
xcc = meta & 0xf
print("xcc histogram", np.bincount(xcc.astype(int), minlength=8))
