#!/usr/bin/env python3
"""Iteration aid: HIP-event time of the config-2 RoIAlign forward call (200 launches), nothing else."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402
from tools.hot_path_bench import time_kernel  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
stream = _lib.current_stream_handle(dev)
h, w, scale = syn.FPN_LEVELS[2]
c, r, res, sr = syn.FPN_DIM, int(os.environ.get("R", 512)), int(os.environ.get("RES", 7)), 2
feat = torch.from_numpy(syn.feature_map(1, c, h, w, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)


def launch():
    assert lib.mi_roi_align_forward(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                    0, 0, stream) == 0


print("fwd (one launch, no scratch) %.2f us" % (time_kernel(launch, 200) * 1e6))
import ctypes  # noqa: E402

lv = _lib.FpnLevels()
lv.num_levels, lv.height[0], lv.width[0] = 1, h, w
wsb = max(lib.mi_roi_align_forward_tiles_workspace_bytes(ctypes.byref(lv), 1, res, res, sr), lib.mi_roi_align_forward_workspace_bytes(r))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)


def launch_ws():
    assert lib.mi_roi_align_forward_ws(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                       0, 0, ws.data_ptr(), wsb, stream) == 0


print("fwd (tile descriptors, two launches, %d KB scratch) %.2f us" % (wsb // 1024, time_kernel(launch_ws, 200) * 1e6))
