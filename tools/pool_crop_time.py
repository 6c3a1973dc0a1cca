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

print(json.dumps(hp.pool_and_crop(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 100)))
