#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase timeline of roi_align_fwd_records (config-2 shape, NCHW).  Needs a TUNING build of the
library (MI_TUNING_BUILD=1 python -m detectron_pytorch_amd.build, or MI_LIB_OVERRIDE=<that .so>): wave 0 of every workgroup
stamps clock64() -- shader-clock ticks, a counter whose base differs between compute units, so only differences inside a
workgroup are used -- at: entry, record header arrived, DMA pieces issued, landed + barrier, bins done, barrier, stores
issued; and where it ran (XCC_ID / HW_ID).  MI_SHADER_MHZ (default 2100) converts ticks."""
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
SLAB = bool(int(os.environ.get("MI_ROI_ALIGN_SLAB", "0")))  # the records-free kernel: one wave per (RoI, 8 channels)
nwg = r * (c // 8) if SLAB else r * (c // 32)
tl = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)


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
assert raw[:, 0].all(), "no stamps: this is not a tuning build of the library"
nst = raw[:, 7] >> 48
xcc = (raw[:, 7] >> 32) & 0xf
hw = raw[:, 7] & 0xffffffff
cu_key = (xcc << 16) | (hw & 0x7f00)
t = (raw[:, :7] - raw[:, :1]).astype(np.float64) / MHZ
print("workgroups %d on %d compute units; stages per RoI: %s" % (len(raw), len(np.unique(cu_key)), dict(zip(*np.unique(nst, return_counts=True)))))
names = ["entry -> record header", "header -> DMA pieces issued", "issued -> landed + barrier", "bins (wave 0)",
         "barrier (other waves' bins)", "tile -> stores issued"]
one = nst == 1
for label, m in (("single-stage RoIs", one), ("all", np.ones(len(raw), bool))):
    print("-- %s (%d workgroups) --" % (label, m.sum()))
    for k in range(6):
        d = t[m, k + 1] - t[m, k]
        print("%-30s mean %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f us" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
    life = t[m, 6]
    print("%-30s mean %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f us" % ("workgroup life (to stores issued)", life.mean(), np.median(life), np.percentile(life, 90), life.max()))
per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1], weights=t[:, 6])
RES = int(os.environ.get("RESIDENT", "15" if SLAB else "3"))
print("sum of the lives of a compute unit's workgroups: mean %.1f  max %.1f us (%d resident at a time -> / %d = %.1f us of kernel)" % (
    per_cu.mean(), per_cu.max(), RES, RES, per_cu.mean() / RES))
