"""TEST INFRASTRUCTURE -- numpy bindings of the CPU oracle (oracle/oracle.c -> liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package.
The product package (detectron_pytorch_amd) never imports it.

Every function takes / returns C-contiguous numpy arrays (float32 unless stated) and mirrors one
`oracle_*` symbol; see oracle.c for the reference file:line each restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile oracle.c (gcc) if liboracle.so is missing or stale."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_roi_align_touched_pixels.restype = ctypes.c_int64
        _lib.oracle_num_threads_available.restype = ctypes.c_int
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _check_batch(rois, n):
    """The reference indexes `bottom_data + roi_batch_ind * channels * height * width` unchecked (roi_align_kernel.cu:90-91,
    :230-231): a RoI of no image reads -- and in the backward WRITES -- outside the arrays.  The HIP operators define such
    rows (the padding rows of the static training path carry image -1) as "pool zeros, no gradient"; a caller of the oracle
    has to take them out first (tests/cpu_backend.py does)."""
    if rois.shape[0] and ((rois[:, 0] < 0) | (rois[:, 0] >= n)).any():
        raise ValueError("oracle: RoI with an image index outside [0, %d): the reference's arithmetic is undefined there" % n)


def num_threads_available():
    return int(lib().oracle_num_threads_available())


def roi_align_forward(features, rois, aligned_height, aligned_width, spatial_scale, sampling_ratio, threads=1):
    features, fp = _f32(features)
    rois, rp = _f32(rois)
    n, c, h, w = features.shape
    r = rois.shape[0]
    _check_batch(rois, n)
    out = np.zeros((r, c, aligned_height, aligned_width), np.float32)
    lib().oracle_roi_align_forward(fp, rp, out.ctypes.data_as(_f32p), n, c, h, w, r, int(aligned_height),
                                   int(aligned_width), ctypes.c_float(spatial_scale), int(sampling_ratio),
                                   int(threads))
    return out


def roi_align_backward(top_grad, rois, feature_shape, spatial_scale, sampling_ratio, threads=1):
    top_grad, tp = _f32(top_grad)
    rois, rp = _f32(rois)
    n, c, h, w = feature_shape
    r, _, ah, aw = top_grad.shape
    _check_batch(rois, n)
    grad = np.zeros((n, c, h, w), np.float32)
    lib().oracle_roi_align_backward(tp, rp, grad.ctypes.data_as(_f32p), n, c, h, w, r, ah, aw,
                                    ctypes.c_float(spatial_scale), int(sampling_ratio), int(threads))
    return grad


def roi_align_touched_pixels(rois, batch, height, width, aligned_height, aligned_width, spatial_scale,
                             sampling_ratio):
    rois, rp = _f32(rois)
    return int(lib().oracle_roi_align_touched_pixels(rp, int(batch), int(height), int(width), rois.shape[0],
                                                     int(aligned_height), int(aligned_width),
                                                     ctypes.c_float(spatial_scale), int(sampling_ratio)))


def roi_align_legacy_forward(features, rois, aligned_height, aligned_width, spatial_scale, threads=1):
    features, fp = _f32(features)
    rois, rp = _f32(rois)
    n, c, h, w = features.shape
    r = rois.shape[0]
    out = np.zeros((r, c, aligned_height, aligned_width), np.float32)
    lib().oracle_roi_align_legacy_forward(fp, rp, out.ctypes.data_as(_f32p), n, c, h, w, r, int(aligned_height),
                                          int(aligned_width), ctypes.c_float(spatial_scale), int(threads))
    return out


def roi_align_legacy_backward(top_grad, rois, feature_shape, spatial_scale, threads=1):
    top_grad, tp = _f32(top_grad)
    rois, rp = _f32(rois)
    n, c, h, w = feature_shape
    r, _, ah, aw = top_grad.shape
    grad = np.zeros((n, c, h, w), np.float32)
    lib().oracle_roi_align_legacy_backward(tp, rp, grad.ctypes.data_as(_f32p), n, c, h, w, r, ah, aw,
                                           ctypes.c_float(spatial_scale), int(threads))
    return grad


def roi_pool_forward(features, rois, pooled_height, pooled_width, spatial_scale, threads=1):
    features, fp = _f32(features)
    rois, rp = _f32(rois)
    n, c, h, w = features.shape
    r = rois.shape[0]
    out = np.zeros((r, c, pooled_height, pooled_width), np.float32)
    argmax = np.zeros((r, c, pooled_height, pooled_width), np.int32)
    lib().oracle_roi_pool_forward(fp, rp, out.ctypes.data_as(_f32p), argmax.ctypes.data_as(_i32p), n, c, h, w, r,
                                  int(pooled_height), int(pooled_width), ctypes.c_float(spatial_scale),
                                  int(threads))
    return out, argmax


def roi_pool_backward(top_grad, rois, argmax, feature_shape, spatial_scale, threads=1):
    top_grad, tp = _f32(top_grad)
    rois, rp = _f32(rois)
    argmax = np.ascontiguousarray(argmax, np.int32)
    n, c, h, w = feature_shape
    r, _, ph, pw = top_grad.shape
    grad = np.zeros((n, c, h, w), np.float32)
    lib().oracle_roi_pool_backward(tp, rp, argmax.ctypes.data_as(_i32p), grad.ctypes.data_as(_f32p), n, c, h, w,
                                   r, ph, pw, ctypes.c_float(spatial_scale), int(threads))
    return grad


def roi_crop_forward(inp, grid_yx, threads=1):
    inp, ip = _f32(inp)
    grid_yx, gp = _f32(grid_yx)
    n, c, h, w = inp.shape
    r, gh, gw, two = grid_yx.shape
    assert two == 2
    out = np.zeros((r, c, gh, gw), np.float32)
    lib().oracle_roi_crop_forward(ip, gp, out.ctypes.data_as(_f32p), n, c, h, w, r, gh, gw, int(threads))
    return out


def roi_crop_backward(inp, grid_yx, grad_output, threads=1):
    inp, ip = _f32(inp)
    grid_yx, gp = _f32(grid_yx)
    grad_output, op = _f32(grad_output)
    n, c, h, w = inp.shape
    r, gh, gw, _ = grid_yx.shape
    grad = np.zeros((n, c, h, w), np.float32)
    lib().oracle_roi_crop_backward(ip, gp, op, grad.ctypes.data_as(_f32p), n, c, h, w, r, gh, gw, int(threads))
    return grad


def nms_cython(dets, thresh):
    """cython_nms.nms semantics: ascending ORIGINAL indices (int64)."""
    dets, dp = _f32(dets)
    n = dets.shape[0]
    keep = np.zeros((max(n, 1),), np.int64)
    k = lib().oracle_nms_cython(dp, n, ctypes.c_float(thresh), keep.ctypes.data_as(_i64p))
    return keep[:k].copy()


def soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method=1):
    """cython_nms.soft_nms semantics: (boxes[:N] float32 [N,5], inds[:N] int64); method 0 hard, 1 linear, 2 gaussian."""
    dets, dp = _f32(dets)
    n = dets.shape[0]
    boxes = np.zeros((max(n, 1), 5), np.float32)
    inds = np.zeros((max(n, 1),), np.int64)
    k = lib().oracle_soft_nms(dp, n, ctypes.c_float(sigma), ctypes.c_float(overlap_thresh), ctypes.c_float(score_thresh),
                              int(method), boxes.ctypes.data_as(_f32p), inds.ctypes.data_as(_i64p))
    return boxes[:k].copy(), inds[:k].copy()


def nms_gpu_semantics(dets_sorted, thresh):
    """nms_gpu semantics: positions in the (pre-sorted) input, int32."""
    dets_sorted, dp = _f32(dets_sorted)
    n = dets_sorted.shape[0]
    keep = np.zeros((max(n, 1),), np.int32)
    k = lib().oracle_nms_gpu_semantics(dp, n, ctypes.c_float(thresh), keep.ctypes.data_as(_i32p))
    return keep[:k].copy()


def bbox_overlaps(boxes, query):
    boxes, bp = _f32(boxes)
    query, qp = _f32(query)
    out = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
    lib().oracle_bbox_overlaps(bp, boxes.shape[0], qp, query.shape[0], out.ctypes.data_as(_f32p))
    return out
