#!/usr/bin/env python3
"""Start / end of every workgroup of roi_crop_bwd_tiles at the config-2 shape (a -DMI_TILE_TIMELINE build of the library:
bash tools/build_defines.sh timeline MI_TILE_TIMELINE=1; MI_LIB_OVERRIDE=.ab_r6/libmi_timeline.so python tools/tile_timeline.py)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import synthetic as syn  # noqa: E402

h = ctypes.CDLL(os.path.abspath(os.environ["MI_LIB_OVERRIDE"]))
h.mi_roi_crop_backward_workspace_bytes.restype = ctypes.c_size_t
dev = torch.device("cuda", 0)
hh, ww, scale = syn.FPN_LEVELS[2]
c, r, res = 256, 512, 7
rois_np = syn.rois_canonical(r, 1, seed=0)
lin = np.linspace(-1, 1, res)
cx = (rois_np[:, 1] + rois_np[:, 3]) * 0.5 * scale / (ww - 1) * 2 - 1
cy = (rois_np[:, 2] + rois_np[:, 4]) * 0.5 * scale / (hh - 1) * 2 - 1
sx = (rois_np[:, 3] - rois_np[:, 1]) * 0.5 * scale / (ww - 1) * 2
sy = (rois_np[:, 4] - rois_np[:, 2]) * 0.5 * scale / (hh - 1) * 2
gy = cy[:, None, None] + sy[:, None, None] * lin[None, :, None] + 0 * lin[None, None, :]
gx = cx[:, None, None] + sx[:, None, None] * lin[None, None, :] + 0 * lin[None, :, None]
grid = torch.from_numpy(np.stack([gy, gx], axis=3).astype(np.float32)).to(dev)
gtop = torch.randn(r, c, res, res, device=dev)
gin = torch.empty(1, c, hh, ww, device=dev)
ws = torch.empty(h.mi_roi_crop_backward_workspace_bytes(r), dtype=torch.uint8, device=dev)
V = ctypes.c_void_p
args = [V(0), V(grid.data_ptr()), V(gtop.data_ptr()), V(gin.data_ptr()), 1, c, hh, ww, r, res, res, V(ws.data_ptr()), ctypes.c_size_t(ws.numel()), V(0)]
for _ in range(5):
    assert h.mi_roi_crop_backward_ws(*args) == 0
torch.cuda.synchronize()
n = 25 * 11 * 8
buf = (ctypes.c_ulonglong * (4 * n))()
assert h.mi_dbg_crop_timeline(buf, n) == 0
a = np.array(buf, dtype=np.uint64).reshape(n, 4)
t0 = a[:, 0].min()
start, end, ent = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2].astype(int)   # us
dur = end - start
print("workgroups %d  kernel span %.1f us  sum of lives %.0f us  mean life %.1f us  max %.1f us" % (n, end.max(), dur.sum(), dur.mean(), dur.max()))
print("life by entries walked:")
for lo in range(0, ent.max() + 1, 4):
    m = (ent >= lo) & (ent < lo + 4)
    if m.any():
        print("  %2d-%2d entries: %4d workgroups, life %.1f us (min %.1f, max %.1f)" % (lo, lo + 3, m.sum(), dur[m].mean(), dur[m].min(), dur[m].max()))
print("workgroups resident over time (us: count):", " ".join("%d:%d" % (t, ((start <= t) & (end > t)).sum()) for t in range(0, int(end.max()) + 1, 5)))
print("starts by time decile of blockIdx:", " ".join("%.0f" % start[i * n // 10:(i + 1) * n // 10].mean() for i in range(10)))
xcc = (a[:, 3] >> np.uint64(32)).astype(int)
print("per XCC: workgroups / last end:", " ".join("%d:%d/%.0f" % (x, (xcc == x).sum(), end[xcc == x].max()) for x in sorted(set(xcc))))
