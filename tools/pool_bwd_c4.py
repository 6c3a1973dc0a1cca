#!/usr/bin/env python3
"""RoIPool forward / backward on a stride-16 map with image-sized RoIs (the C4 configs' use of it): 512 RoIs of 64-600 px x C
channels x 7x7 on 50x84.  usage: [MI_LIB_OVERRIDE=...] python tools/pool_bwd_c4.py [channels]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402
from tools import hot_path_bench as hp  # noqa: E402

if os.environ.get("MI_LIB_OVERRIDE"):
    h = ctypes.CDLL(os.path.abspath(os.environ["MI_LIB_OVERRIDE"]))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = restype, argtypes
    _lib._lib = h
dev = torch.device("cuda", 0)
lib, stream = _lib.lib(), _lib.current_stream_handle(dev)
n, c, hh, ww, r = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 50, 84, 512
feat = torch.from_numpy(syn.feature_map(n, c, hh, ww, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, n, seed=0, side=(64.0, 600.0))).to(dev)
out = torch.empty((r, c, 7, 7), device=dev)
arg = torch.empty((r, c, 7, 7), dtype=torch.int32, device=dev)
gtop = torch.randn(r, c, 7, 7, device=dev)
gin = torch.zeros(n, c, hh, ww, device=dev)
fwd = hp.time_kernel(lambda: lib.mi_roi_pool_forward(feat.data_ptr(), rois.data_ptr(), out.data_ptr(), arg.data_ptr(), n, c, hh, ww, r, 7, 7, 1.0 / 16, stream), 30)
fill = bool(os.environ.get("MI_FILL"))  # the atomic form of earlier libraries needs the caller's zero fill


def bwd():
    if fill:
        gin.zero_()
    lib.mi_roi_pool_backward(gtop.data_ptr(), rois.data_ptr(), arg.data_ptr(), gin.data_ptr(), n, c, hh, ww, r, 7, 7, 1.0 / 16, stream)


print(json.dumps({"shape": "512 RoIs of 64-600 px x %d ch x 7x7 on 50x84, scale 1/16" % c, "roi_pool_fwd_us": round(fwd * 1e6, 1),
                  "roi_pool_bwd_us": round(hp.time_kernel(bwd, 20) * 1e6, 1), "zero_fill_in_the_call": fill}))
