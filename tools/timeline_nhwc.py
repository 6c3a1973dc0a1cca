#!/usr/bin/env python3
"""Tuning aid: per-workgroup phase timeline of roi_align_fwd_nhwc (config-2 shape, channels-last features).
Wave 0 of every workgroup stamps clock64() -- the shader clock, one time base per XCD -- at 6 points, and where it ran
(XCC_ID / HW_ID) and which RoI it pooled (mi_dbg_roi_align_timeline).  Times are normalised per XCD; MHZ is nominal."""
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
MHZ = float(os.environ.get("MI_SHADER_MHZ", "2100"))
raw = tl.cpu().numpy()
raw = raw[raw[:, 0] != 0]
print("workgroups stamped", len(raw))
xcc = (raw[:, 6] >> 32) & 0xf
hw = raw[:, 6] & 0xffffffff
cu_key = (xcc << 16) | (hw & 0x7f00)         # HW_ID: cu_id [11:8], sh_id [12], se_id [14:13] (bit 16: which of the unit's two slots)
roi = raw[:, 7]
t = raw[:, :6].astype(np.float64)
t = (t - t[:, :1]) / MHZ                     # the counters of different compute units do not share a base: differences only
names = ["record fetch", "taps (wave 0)", "tile write", "barrier (other waves)", "copy-out"]
for k in range(5):
    d = t[:, k + 1] - t[:, k]
    print("%-22s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max()))
life = t[:, 5] - t[:, 0]
print("%-22s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % ("workgroup life", life.mean(), np.median(life), np.percentile(life, 90), life.max()))

# cost of a RoI = tap loads of its seven waves: per bin row, (distinct feature rows its y samples tap) x 4 * PW * sr / 2
rois_h = rois.cpu().numpy()
y1 = rois_h[:, 2] * np.float32(scale)
rh = np.maximum(rois_h[:, 4] * np.float32(scale) - y1, 1.0)
bh = rh / res
cost = np.zeros(r)
for ph in range(res):
    rows = [set() for _ in range(r)]
    for iy in range(sr):
        y = np.clip(y1 + ph * bh + (iy + 0.5) * bh / sr, 0, h - 1)
        lo = np.floor(y).astype(int)
        for i in range(r):
            rows[i].update((lo[i], min(lo[i] + 1, h)))
    cost += np.array([len(s) for s in rows]) * 2 * sr * res
c = cost[roi]
taps = t[:, 2] - t[:, 1]
print("tap loads per RoI: mean %.0f min %.0f max %.0f" % (cost.mean(), cost.min(), cost.max()))
print("corr(taps time of wave 0, own cost) = %.2f" % np.corrcoef(taps, c)[0, 1])
keys, inv, cnt = np.unique(cu_key, return_inverse=True, return_counts=True)
print("compute units used %d; workgroups per unit: %s" % (len(keys), dict(zip(*np.unique(cnt, return_counts=True)))))
cu_cost = np.bincount(inv, weights=c)
print("corr(life, cost of all workgroups on the same unit) = %.2f" % np.corrcoef(life, cu_cost[inv])[0, 1])
print("corr(taps time, cost of the unit) = %.2f" % np.corrcoef(taps, cu_cost[inv])[0, 1])
print("unit cost: mean %.0f p90 %.0f max %.0f" % (cu_cost.mean(), np.percentile(cu_cost, 90), cu_cost.max()))
d = []
unit_life = np.zeros(len(keys))
np.maximum.at(unit_life, inv, life)
for k in range(len(keys)):
    members = np.nonzero(inv == k)[0]          # row index == blockIdx (every workgroup is stamped)
    if len(members) == 2:
        d.append(abs(int(members[1] // 8) - int(members[0] // 8)))
print("blockIdx / 8 distance of the two workgroups of a unit: %s" % dict(zip(*np.unique(d, return_counts=True))))
wwin = np.floor(rois_h[:, 3] * scale) - np.floor(rois_h[:, 1] * scale) + 2
px = np.bincount(inv, weights=((cost / (2 * sr * res)) * wwin)[roi])
print("life of a unit (its longer workgroup): mean %.2f p90 %.2f max %.2f us; corr with its tap loads %.2f, with rows x window "
      "width (distinct pixels) %.2f" % (unit_life.mean(), np.percentile(unit_life, 90), unit_life.max(),
                                       np.corrcoef(unit_life, cu_cost)[0, 1], np.corrcoef(unit_life, px)[0, 1]))
for x in np.unique(xcc):
    m = xcc == x
    print("XCD %d: workgroups %3d  cost %7.0f  life mean %5.2f max %5.2f us" % (x, m.sum(), c[m].sum(), life[m].mean(), life[m].max()))
