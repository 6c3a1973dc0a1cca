#!/usr/bin/env python3
"""Sweep of the float4 copy kernel behind roofline.copy_ceiling (MI_COPY_VARIANT = blocks_per_cu * 100 + unroll * 10 + nt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectron_pytorch_amd import _lib
from tools.hot_path_bench import time_kernel
dev = torch.device("cuda", 0); lib = _lib.lib(); stream = _lib.current_stream_handle(dev)
a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev); b = torch.empty_like(a)
n = a.numel() * 4
print("torch copy_ %.0f GB/s" % (2 * n / time_kernel(lambda: b.copy_(a), 20) / 1e9))
for per_cu in (2, 4, 8, 16, 32, 64):
    for unroll in (1, 4, 8):
        for nt in (0, 1):
            os.environ["MI_COPY_VARIANT"] = str(per_cu * 100 + unroll * 10 + nt)
            lib.mi_dbg_reload_tuning()  # the library reads its tuning variables once
            sec = time_kernel(lambda: lib.mi_dbg_copy_float4(a.data_ptr(), b.data_ptr(), n, stream), 20)
            print("blocks/CU %2d unroll %d nt %d: %.0f GB/s" % (per_cu, unroll, nt, 2 * n / sec / 1e9), flush=True)
