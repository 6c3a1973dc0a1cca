#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase accounting of roi_align_fwd_persist (config-2 shape, NCHW).  Needs a TUNING build of the
library (MI_LIB_OVERRIDE=<that .so>): wave 0 of every resident workgroup sums clock64() differences over its units
(unit = one stage of one item): [0] wait for the landing + barrier B1, [1] bins (+ edge patch), [2] barrier B2 (the other
waves' bins), [3] issue of the next unit's window / table pieces, [4] issue of this unit's stores; [5] its whole life,
[6] items << 32 | units, [7] XCC_ID / HW_ID.  MI_SHADER_MHZ (default 2100) converts ticks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402

if os.environ.get("MI_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MI_LIB_OVERRIDE"])
dev = torch.device("cuda", 0)
lib = _lib.lib()
stream = _lib.current_stream_handle(dev)
h, w, scale = syn.FPN_LEVELS[2]
c, r, res, sr = syn.FPN_DIM, 512, 7, 2
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(r), dtype=torch.uint8, device=dev)
tl = torch.zeros((r * (c // 32), 8), dtype=torch.int64, device=dev)


def launch():
    assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                       0, 0, ws.data_ptr(), ws.numel(), stream) == 0


for _ in range(5):
    launch()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(100):
    launch()
t1.record()
torch.cuda.synchronize()
print("this build: %.2f us per call (a tuning build is slower than the release build)" % (t0.elapsed_time(t1) * 10))
lib.mi_dbg_roi_align_timeline(tl.data_ptr())
launch()
torch.cuda.synchronize()
lib.mi_dbg_roi_align_timeline(None)
MHZ = float(os.environ.get("MI_SHADER_MHZ", "2100"))
raw = tl.cpu().numpy()
raw = raw[raw[:, 5] != 0]
assert len(raw), "no stamps: not a tuning build of the library, or the per-item kernel ran (MI_ROI_ALIGN_FWD_PERSIST=0)"
items, units = raw[:, 6] >> 32, raw[:, 6] & 0xffffffff
xcc = (raw[:, 7] >> 32) & 0xf
cu_key = (xcc << 16) | (raw[:, 7] & 0x7f00)
t = raw[:, :6].astype(np.float64) / MHZ
print("resident workgroups %d on %d compute units; items per workgroup min %d mean %.2f max %d; units (stages) %d for %d items" % (
    len(raw), len(np.unique(cu_key)), items.min(), items.mean(), items.max(), units.sum(), items.sum()))
names = ["wait landing + barrier B1", "bins (wave 0, + edge patch)", "barrier B2 (other waves)", "issue next unit's pieces",
         "issue this unit's stores"]
print("-- per unit (a workgroup's phase sum / its units), us --")
for k in range(5):
    d = t[:, k] / units
    print("%-30s mean %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
print("-- per workgroup, us --")
for k in range(5):
    print("%-30s mean %6.2f  max %6.2f" % (names[k], t[:, k].mean(), t[:, k].max()))
life = t[:, 5]
print("%-30s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % ("workgroup life", life.mean(), np.percentile(life, 10),
                                                                         np.median(life), np.percentile(life, 90), life.max()))
print("accounted %.1f %% of the lives (the rest: header loads, first issue, loop bookkeeping)" % (100 * t[:, :5].sum() / life.sum()))
