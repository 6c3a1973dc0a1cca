"""TEST INFRASTRUCTURE -- numpy bindings of the REFERENCE's own code built for the host.

oracle/_ref/ holds the reference's .cu kernels compiled with g++ (oracle/build_ref.py +
oracle/cuda_on_cpu.h) and its Cython NMS / IoU modules.  The functions below call the
reference's exported launchers (same symbol names, same argument order as
roi_align_kernel.h:13-30, roi_pooling_kernel.h:8-20, roi_crop_cuda_kernel.h:6-37,
nms_cuda_kernel.h:5-6) with host pointers, allocating and zero-filling outputs exactly as
the reference's Python Functions do.

`available()` is False when oracle/_ref has not been built (no /root/reference and no prebuilt
files); callers skip in that case.
"""
import ctypes
import importlib.util
import os
import sysconfig

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_libs = {}
_mods = {}

_CUDA_LIBS = ["libref_roi_align.so", "libref_roi_align_legacy.so", "libref_roi_pool.so", "libref_roi_crop.so",
              "libref_nms.so"]


def available():
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    names = _CUDA_LIBS + ["cython_nms" + suffix, "cython_bbox" + suffix]
    return all(os.path.exists(os.path.join(_REF, n)) for n in names)


def _lib(name):
    if name not in _libs:
        _libs[name] = ctypes.CDLL(os.path.join(_REF, name))
    return _libs[name]


def _mod(name):
    if name not in _mods:
        path = os.path.join(_REF, name + sysconfig.get_config_var("EXT_SUFFIX"))
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _mods[name] = mod
    return _mods[name]


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


# ---- RoIAlign (Caffe2 semantics): ROIAlignForwardLaucher / ROIAlignBackwardLaucher ----------
def roi_align_forward(features, rois, aligned_height, aligned_width, spatial_scale, sampling_ratio):
    features, fp = _f32(features)
    rois, rp = _f32(rois)
    n, c, h, w = features.shape
    r = rois.shape[0]
    out = np.zeros((r, c, aligned_height, aligned_width), np.float32)
    _lib("libref_roi_align.so").ROIAlignForwardLaucher(
        fp, ctypes.c_float(spatial_scale), r, h, w, c, int(aligned_height), int(aligned_width),
        int(sampling_ratio), rp, out.ctypes.data_as(_f32p), None)
    return out


def roi_align_backward(top_grad, rois, feature_shape, spatial_scale, sampling_ratio):
    top_grad, tp = _f32(top_grad)
    rois, rp = _f32(rois)
    n, c, h, w = feature_shape
    r, _, ah, aw = top_grad.shape
    grad = np.zeros((n, c, h, w), np.float32)
    _lib("libref_roi_align.so").ROIAlignBackwardLaucher(
        tp, ctypes.c_float(spatial_scale), n, r, h, w, c, ah, aw, int(sampling_ratio), rp,
        grad.ctypes.data_as(_f32p), None)
    return grad


# ---- RoIAlign (legacy) -------------------------------------------------------------------
def roi_align_legacy_forward(features, rois, aligned_height, aligned_width, spatial_scale):
    features, fp = _f32(features)
    rois, rp = _f32(rois)
    n, c, h, w = features.shape
    r = rois.shape[0]
    out = np.zeros((r, c, aligned_height, aligned_width), np.float32)
    _lib("libref_roi_align_legacy.so").ROIAlignForwardLaucher(
        fp, ctypes.c_float(spatial_scale), r, h, w, c, int(aligned_height), int(aligned_width), rp,
        out.ctypes.data_as(_f32p), None)
    return out


def roi_align_legacy_backward(top_grad, rois, feature_shape, spatial_scale):
    top_grad, tp = _f32(top_grad)
    rois, rp = _f32(rois)
    n, c, h, w = feature_shape
    r, _, ah, aw = top_grad.shape
    grad = np.zeros((n, c, h, w), np.float32)
    _lib("libref_roi_align_legacy.so").ROIAlignBackwardLaucher(
        tp, ctypes.c_float(spatial_scale), n, r, h, w, c, ah, aw, rp, grad.ctypes.data_as(_f32p), None)
    return grad


# ---- RoIPool -----------------------------------------------------------------------------
def roi_pool_forward(features, rois, pooled_height, pooled_width, spatial_scale):
    features, fp = _f32(features)
    rois, rp = _f32(rois)
    n, c, h, w = features.shape
    r = rois.shape[0]
    out = np.zeros((r, c, pooled_height, pooled_width), np.float32)
    argmax = np.zeros((r, c, pooled_height, pooled_width), np.int32)
    _lib("libref_roi_pool.so").ROIPoolForwardLaucher(
        fp, ctypes.c_float(spatial_scale), r, h, w, c, int(pooled_height), int(pooled_width), rp,
        out.ctypes.data_as(_f32p), argmax.ctypes.data_as(_i32p), None)
    return out, argmax


def roi_pool_backward(top_grad, rois, argmax, feature_shape, spatial_scale):
    top_grad, tp = _f32(top_grad)
    rois, rp = _f32(rois)
    argmax = np.ascontiguousarray(argmax, np.int32)
    n, c, h, w = feature_shape
    r, _, ph, pw = top_grad.shape
    grad = np.zeros((n, c, h, w), np.float32)
    _lib("libref_roi_pool.so").ROIPoolBackwardLaucher(
        tp, ctypes.c_float(spatial_scale), n, r, h, w, c, ph, pw, rp, grad.ctypes.data_as(_f32p),
        argmax.ctypes.data_as(_i32p), None)
    return grad


# ---- RoICrop: argument order of roi_crop_cuda.c:23-44 (sizes 1,3,2,0 then strides 0,1,2,3) ------
def _strides(a):
    return [s // a.itemsize for s in a.strides]


def roi_crop_forward(inp, grid_yx):
    inp, ip = _f32(inp)
    grid_yx, gp = _f32(grid_yx)
    n, c, h, w = inp.shape
    r, gh, gw, _ = grid_yx.shape
    out = np.zeros((r, c, gh, gw), np.float32)
    i_s, g_s, o_s = _strides(inp), _strides(grid_yx), _strides(out)
    ok = _lib("libref_roi_crop.so").BilinearSamplerBHWD_updateOutput_cuda_kernel(
        c, gw, gh, r, c, h, w, n,
        ip, i_s[0], i_s[1], i_s[2], i_s[3],
        gp, g_s[0], g_s[3], g_s[1], g_s[2],
        out.ctypes.data_as(_f32p), o_s[0], o_s[1], o_s[2], o_s[3], None)
    assert ok == 1
    return out


def roi_crop_backward(inp, grid_yx, grad_output):
    inp, ip = _f32(inp)
    grid_yx, gp = _f32(grid_yx)
    grad_output, op = _f32(grad_output)
    n, c, h, w = inp.shape
    r, gh, gw, _ = grid_yx.shape
    grad_in = np.zeros_like(inp)
    grad_grid = np.zeros_like(grid_yx)
    i_s, g_s, o_s = _strides(inp), _strides(grid_yx), _strides(grad_output)
    ok = _lib("libref_roi_crop.so").BilinearSamplerBHWD_updateGradInput_cuda_kernel(
        c, gw, gh, r, c, h, w, n,
        ip, i_s[0], i_s[1], i_s[2], i_s[3],
        gp, g_s[0], g_s[3], g_s[1], g_s[2],
        grad_in.ctypes.data_as(_f32p), i_s[0], i_s[1], i_s[2], i_s[3],
        grad_grid.ctypes.data_as(_f32p), g_s[0], g_s[3], g_s[1], g_s[2],
        op, o_s[0], o_s[1], o_s[2], o_s[3], None)
    assert ok == 1
    return grad_in, grad_grid


# ---- NMS ---------------------------------------------------------------------------------
def nms_gpu(dets_sorted, thresh):
    """nms_cuda_compute (nms_cuda_kernel.cu:87-161): positions in the pre-sorted input."""
    dets_sorted, dp = _f32(dets_sorted)
    n = dets_sorted.shape[0]
    keep = np.zeros((max(n, 1),), np.int32)
    num_out = np.zeros((1,), np.int32)
    _lib("libref_nms.so").nms_cuda_compute(keep.ctypes.data_as(_i32p), num_out.ctypes.data_as(_i32p), dp, n, 5,
                                           ctypes.c_float(thresh))
    return keep[:int(num_out[0])].copy()


def cython_nms(dets, thresh):
    """utils.cython_nms.nms (cython_nms.pyx:37-87), the NMS the reference model actually runs."""
    return _mod("cython_nms").nms(np.ascontiguousarray(dets, np.float32), np.float32(thresh))


def cython_soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method=1):
    return _mod("cython_nms").soft_nms(np.ascontiguousarray(dets, np.float32), np.float32(sigma),
                                       np.float32(overlap_thresh), np.float32(score_thresh), np.uint8(method))


def cython_bbox_overlaps(boxes, query):
    return _mod("cython_bbox").bbox_overlaps(np.ascontiguousarray(boxes, np.float32),
                                             np.ascontiguousarray(query, np.float32))
