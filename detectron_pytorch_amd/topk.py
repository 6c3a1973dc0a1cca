"""Sorted top-k on the device, many problems per call -- host side of mi_topk_batched.

The selections around the NMS kernels (np.argsort / np.argpartition in the reference: lib/modeling/generate_proposals.py:
131-142, collect_and_distribute_fpn_rpn_proposals.py:85-86, core/test.py:781-784).  Same contract as
`torch.topk(..., largest=True, sorted=True)` on 1-D float32 rows, with a defined order of ties (lower index first) and NaN
ranked last; one launch for all rows instead of a dozen sort / merge launches per row.
"""
import ctypes

import torch

from . import _lib

MAX_K = 4096
MAX_N = 1 << 24


def supported(n, k):
    return 0 <= k <= n <= MAX_N and k <= MAX_K


def topk_flat(rows, ks):
    """rows: 1-D contiguous float32 device tensors; ks: one k per row (k <= len(row), k <= 4096).  Returns (values
    [sum k], indices int64 [sum k], offsets): row i's result is [offsets[i], offsets[i + 1]) of both -- asynchronous on
    the current stream, one launch per 16 rows."""
    dev = rows[0].device
    for r, k in zip(rows, ks):
        _lib.require_cuda(r, "values")
        if r.dtype != torch.float32 or r.dim() != 1 or not r.is_contiguous() or r.device != dev:
            raise ValueError("topk_many takes contiguous 1-D float32 tensors on one device")
        if not supported(r.numel(), k):
            raise ValueError("top-%d of %d values is outside mi_topk_batched's range (k <= %d, n <= %d)"
                             % (k, r.numel(), MAX_K, MAX_N))
    p = len(rows)
    offs = [0]
    for k in ks:
        offs.append(offs[-1] + int(k))
    vals = torch.empty((offs[-1],), dtype=torch.float32, device=dev)
    idx = torch.empty((offs[-1],), dtype=torch.int64, device=dev)
    val_arr = (ctypes.c_void_p * p)(*[r.data_ptr() for r in rows])
    n_arr = (ctypes.c_int * p)(*[int(r.numel()) for r in rows])
    k_arr = (ctypes.c_int * p)(*[int(k) for k in ks])
    ov_arr = (ctypes.c_void_p * p)(*[vals.data_ptr() + 4 * offs[i] for i in range(p)])
    oi_arr = (ctypes.c_void_p * p)(*[idx.data_ptr() + 8 * offs[i] for i in range(p)])
    lib = _lib.lib()
    ws_bytes = lib.mi_topk_batched_workspace_bytes(p, n_arr, k_arr)
    workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.mi_topk_batched(p, val_arr, n_arr, k_arr, ov_arr, oi_arr, workspace.data_ptr(), ws_bytes,
                                 _lib.current_stream_handle(dev))
    _lib.check(rc, "mi_topk_batched")
    return vals, idx, offs


def topk_many(rows, ks):
    """[(values [k], indices int64 [k]), ...] for the rows of `topk_flat`."""
    if not rows:
        return []
    vals, idx, offs = topk_flat(rows, ks)
    return [(vals[offs[i]:offs[i + 1]], idx[offs[i]:offs[i + 1]]) for i in range(len(rows))]


def topk_rows(values, k):
    """Sorted top-k of every row of a contiguous [N, total] tensor: (values [N,k], indices [N,k]), torch.topk(dim=1)."""
    n = values.size(0)
    vals, idx, _ = topk_flat(list(values.unbind(0)), [k] * n)
    return vals.view(n, k), idx.view(n, k)


def topk(values, k):
    """Sorted top-k of one 1-D tensor: (values [k], indices [k])."""
    return topk_many([values.contiguous()], [k])[0]
