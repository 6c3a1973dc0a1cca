#!/usr/bin/env python3
"""RoIPool / RoICrop forward + backward at the config-2 shape (tools/hot_path_bench.pool_and_crop): us per C-ABI call and
GB/s of algorithmic bytes.  usage: python tools/pool_crop_time.py [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from detectron_pytorch_amd import _lib  # noqa: E402
from tools import hot_path_bench as hp  # noqa: E402

if os.environ.get("MI_LIB_OVERRIDE"):      # another build of the library (tuning switches, an older commit): bind what it has
    import ctypes

    h = ctypes.CDLL(os.path.abspath(os.environ["MI_LIB_OVERRIDE"]))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = restype, argtypes
    _lib._lib = h

dev = torch.device("cuda", 0)
print(json.dumps(hp.pool_and_crop(dev, int(sys.argv[1]) if len(sys.argv) > 1 else 100)))
if os.environ.get("MI_BENCH_C4"):  # RoIPool forward on a stride-16 map with image-sized RoIs (the C4 configs' use of it)
    import numpy as np
    from detectron_pytorch_amd import synthetic as syn

    lib, stream = _lib.lib(), _lib.current_stream_handle(dev)
    n, c, h, w, r = 1, 256, 50, 84, 512
    feat = torch.from_numpy(syn.feature_map(n, c, h, w, seed=0)).to(dev)
    rois = torch.from_numpy(syn.rois_canonical(r, n, seed=0, side=(64.0, 600.0))).to(dev)
    out = torch.empty((r, c, 7, 7), device=dev)
    arg = torch.empty((r, c, 7, 7), dtype=torch.int32, device=dev)
    sec = hp.time_kernel(lambda: lib.mi_roi_pool_forward(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), arg.data_ptr(), n, c, h, w,
                                                          r, 7, 7, 1.0 / 16, stream), 100)
    print(json.dumps({"roi_pool_fwd_c4_shape_us": round(sec * 1e6, 2), "shape": "512 RoIs of 64-600 px x 256 ch x 7x7 on 50x84, scale 1/16"}))
