#!/usr/bin/env python3
"""The generic (reference-order, bit-exact) RoIAlign kernels at the config-2 shape, NCHW and channels-last: us per C-ABI call.
usage: MI_ROI_ALIGN_IMPL=direct [MI_LIB_OVERRIDE=...] python tools/direct_time.py"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402
from tools import hot_path_bench as hp  # noqa: E402

assert os.environ.get("MI_ROI_ALIGN_IMPL") == "direct", "set MI_ROI_ALIGN_IMPL=direct (read once by the library)"
if os.environ.get("MI_LIB_OVERRIDE"):
    h = ctypes.CDLL(os.path.abspath(os.environ["MI_LIB_OVERRIDE"]))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = restype, argtypes
    _lib._lib = h
dev = torch.device("cuda", 0)
lib, stream = _lib.lib(), _lib.current_stream_handle(dev)
hh, ww, scale = syn.FPN_LEVELS[2]
c, r, res = 256, 512, 7
feat = torch.from_numpy(syn.feature_map(1, c, hh, ww, seed=0)).to(dev)
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
out = torch.empty((r, c, res, res), device=dev)
gtop = torch.randn(r, c, res, res, device=dev)
gin = torch.zeros(1, c, hh, ww, device=dev)
line = {}
for name, layout, f in (("nchw", _lib.LAYOUT_NCHW, feat), ("channels_last", _lib.LAYOUT_NHWC, feat.permute(0, 2, 3, 1).contiguous())):
    def fwd():
        assert lib.mi_roi_align_forward(f.data_ptr(), rois.data_ptr(), out.data_ptr(), 1, c, hh, ww, r, res, res, scale, 2,
                                        _lib.ROI_ALIGN_CAFFE2, layout, stream) == 0

    def bwd():
        gin.zero_()
        assert lib.mi_roi_align_backward(gtop.data_ptr(), rois.data_ptr(), gin.data_ptr(), 1, c, hh, ww, r, res, res, scale, 2,
                                         _lib.ROI_ALIGN_CAFFE2, layout, stream) == 0
    line[name] = {"fwd_us": round(hp.time_kernel(fwd, 50) * 1e6, 1), "bwd_us_incl_fill": round(hp.time_kernel(bwd, 10) * 1e6, 1)}
print(json.dumps({"roi_align_direct_config2": line}))
