#!/usr/bin/env python3
"""Tuning aid: per-item timeline of roi_align_fwd_pipe (MI_ROI_ALIGN_IMPL=pipe), config-2 shape.  Worker wave 0, the
storer and the agent of every workgroup stamp s_memtime (100 MHz) at the begin / end of their part of the first 24 items
(mi_dbg_roi_align_timeline).  usage: python tools/timeline_pipe.py [nhwc]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MI_ROI_ALIGN_IMPL"] = "pipe"
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402

nhwc = "nhwc" in sys.argv[1:]
dev = torch.device("cuda", 0)
lib = _lib.lib()
stream = _lib.current_stream_handle(dev)
h, w, scale = syn.FPN_LEVELS[2]
c, r, res, sr = syn.FPN_DIM, 512, 7, 2
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
if nhwc:
    feat = feat.permute(0, 2, 3, 1).contiguous()
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(r), dtype=torch.uint8, device=dev)
NWG, NIT = 256, 24
tl = torch.zeros((NWG, NIT, 48), dtype=torch.int64, device=dev)


def launch():
    assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                       0, 1 if nhwc else 0, ws.data_ptr(), ws.numel(), stream) == 0


for _ in range(5):
    launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(tl.data_ptr())
launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(None)
t = tl.cpu().numpy().astype(np.float64)
t[t == 0] = np.nan
GHZ = float(os.environ.get("SCLK_GHZ", "2.1"))  # shader clock; the stamps are s_memtime ticks, per-XCD time bases
t = (t - np.nanmin(t, axis=(1, 2), keepdims=True)) / (GHZ * 1e3)  # us since the workgroup's first stamp
print("layout", "NHWC" if nhwc else "NCHW", "(us at %.1f GHz; times relative to each workgroup's first stamp)" % GHZ)


def show(label, d):
    print("%-48s mean %5.2f p50 %5.2f p90 %5.2f max %5.2f us" % (label, np.nanmean(d), np.nanmedian(d), np.nanpercentile(d, 90), np.nanmax(d)))


mid = t[:, 4:16]  # steady-state iterations
arr, dep = mid[:, :, 0:16], mid[:, :, 16:32]
start = np.nanmin(np.concatenate([t[:, 3:15, 16:32]], axis=2), axis=2)  # first wave out of the previous barrier
show("iteration period (first departure k - first departure k-1)", np.diff(np.nanmin(t[:, 3:17, 16:32], axis=2), axis=1))
rel = arr - start[:, :, None]
import warnings
warnings.filterwarnings("ignore")
for wv in range(13):
    show("wave %2d reaches the barrier after" % wv, rel[:, :, wv])
show("last arrival -> first departure", np.nanmin(dep, axis=2) - np.nanmax(arr, axis=2))
show("first -> last departure", np.nanmax(dep, axis=2) - np.nanmin(dep, axis=2))
show("loader 0: issue windows of item k+2", mid[:, :, 33] - mid[:, :, 32])
show("bin wave 4: bins of item k", mid[:, :, 41] - mid[:, :, 40])
show("loader 0: top -> first piece issued", mid[:, :, 44] - mid[:, :, 32])
show("loader 0 top after the barrier", mid[:, :, 32] - start)
show("storer top after the barrier", mid[:, :, 35] - start)
show("storer: tile read into registers", mid[:, :, 42] - mid[:, :, 35])
show("storer: tile k-1 -> global", mid[:, :, 36] - mid[:, :, 35])
show("agent top after the barrier", mid[:, :, 37] - start)
show("agent: header read + step", mid[:, :, 43] - mid[:, :, 37])
show("bin wave 4 top after the barrier", mid[:, :, 40] - start)
show("agent: step + fetch", mid[:, :, 38] - mid[:, :, 37])
show("agent: counted wait", mid[:, :, 39] - mid[:, :, 38])
items = np.sum(~np.isnan(t[:, :, 0]), axis=1)
print("iterations stamped per workgroup (max %d): min %d mean %.1f max %d" % (NIT, items.min(), items.mean(), items.max()))
show("last stamp of the workgroup", np.nanmax(t, axis=(1, 2)))
wgi = 5
print("workgroup %d, iteration 6: arrivals per wave:" % wgi, " ".join("%.2f" % v for v in t[wgi, 6, 0:16] - np.nanmin(t[wgi, 5, 16:32])))
