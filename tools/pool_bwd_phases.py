#!/usr/bin/env python3
"""Per-phase clock sums of roi_pool_bwd_tiles at the config-2 shape (tuning build of the library: MI_LIB_OVERRIDE=.ab_r6/libmi_tuning.so).
usage: MI_LIB_OVERRIDE=... python tools/pool_bwd_phases.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from detectron_pytorch_amd import synthetic as syn  # noqa: E402

h = ctypes.CDLL(os.path.abspath(os.environ["MI_LIB_OVERRIDE"]))
dev = torch.device("cuda", 0)
hh, ww, scale = syn.FPN_LEVELS[2]
c, r, res = 256, 512, 7
feat = torch.from_numpy(syn.feature_map(1, c, hh, ww, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
arg = torch.empty((r, c, res, res), dtype=torch.int32, device=dev)
gtop = torch.randn(r, c, res, res, device=dev)
gin = torch.empty(1, c, hh, ww, device=dev)
V = ctypes.c_void_p
args_f = [V(feat.data_ptr()), V(rois.data_ptr()), V(out.data_ptr()), V(arg.data_ptr()), 1, c, hh, ww, r, res, res, ctypes.c_float(scale), V(0)]
assert h.mi_roi_pool_forward(*args_f) == 0
args_b = [V(gtop.data_ptr()), V(rois.data_ptr()), V(arg.data_ptr()), V(gin.data_ptr()), 1, c, hh, ww, r, res, res, ctypes.c_float(scale), V(0)]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    assert h.mi_roi_pool_backward(*args_b) == 0
torch.cuda.synchronize()
h.mi_dbg_pool_counters(buf)
n = 10
for _ in range(n):
    assert h.mi_roi_pool_backward(*args_b) == 0
torch.cuda.synchronize()
h.mi_dbg_pool_counters(buf)
names = ["total", "zero", "scan", "tabulate", "walk", "decode", "passes", "write", "wave-entries", "round-end wait", "third passes"]
v = [x / n for x in buf]
waves = 13 * 11 * 8 * 4
print("waves", waves)
for i, nm in enumerate(names):
    if i in (8, 10):
        print("%-16s %12.0f per launch (%.1f per wave)" % (nm, v[i], v[i] / waves))
    else:
        print("%-16s %12.0f clocks per wave (%.1f%% of total)" % (nm, v[i] / waves, 100 * v[i] / max(v[0], 1)))
print("wall clock span of the LAST launch's wave ends (100 MHz ticks):", buf[11] - buf[12])
if v[8]:
    print("decode per wave-entry %.0f clocks, passes %.0f" % (v[5] / v[8], v[6] / v[8]))
