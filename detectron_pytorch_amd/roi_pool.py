"""RoIPool -- host-side mirror of lib/model/roi_pooling/functions/roi_pool.py:6-38 and
modules/roi_pool.py:5-14, backed by mi_roi_pool_*.

    RoIPoolFunction(pooled_height, pooled_width, spatial_scale)(features, rois)

Kept: output [R, C, ph, pw]; an int32 argmax tensor of the same shape holding flat indices into the
whole NCHW feature tensor (-1 for empty bins) saved for the backward (:18); no gradient to rois.
Dropped: the reference's CPU branch (:20-23) feeds a non-contiguous permute view to an
NHWC/batch-1 C routine and is unusable; CPU input raises instead.
"""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

from . import _lib


def roi_pool_forward(features, rois, pooled_height, pooled_width, spatial_scale):
    """Raw forward: returns (output, argmax int32)."""
    _lib.require_cuda(features, "features")
    if features.dtype != torch.float32 or rois.dtype != torch.float32:
        raise TypeError("RoIPool supports float32 only (as the reference)")
    if rois.dim() != 2 or rois.size(1) != 5:
        raise ValueError("rois must be [R, 5]")
    features = features.contiguous()
    rois = rois.contiguous()
    n, c, h, w = features.shape
    r = rois.size(0)
    output = torch.empty((r, c, pooled_height, pooled_width), dtype=features.dtype, device=features.device)
    argmax = torch.empty((r, c, pooled_height, pooled_width), dtype=torch.int32, device=features.device)
    with torch.cuda.device(features.device):
        rc = _lib.lib().mi_roi_pool_forward(features.data_ptr(), rois.data_ptr(), output.data_ptr(),
                                            argmax.data_ptr(), n, c, h, w, r, int(pooled_height),
                                            int(pooled_width), float(spatial_scale),
                                            _lib.current_stream_handle(features.device))
    _lib.check(rc, "mi_roi_pool_forward")
    return output, argmax


def roi_pool_backward(grad_output, rois, argmax, feature_size, pooled_height, pooled_width, spatial_scale):
    _lib.require_cuda(grad_output, "grad_output")
    grad_output = grad_output.contiguous()
    n, c, h, w = feature_size
    # every element is written by the call (roi_pooling_kernel.cu:202): no zero fill (the reference's .zero_() is redundant there too)
    grad_input = torch.empty((n, c, h, w), dtype=grad_output.dtype, device=grad_output.device)
    with torch.cuda.device(grad_output.device):
        rc = _lib.lib().mi_roi_pool_backward(grad_output.data_ptr(), rois.data_ptr(), argmax.data_ptr(),
                                             grad_input.data_ptr(), n, c, h, w, rois.size(0),
                                             int(pooled_height), int(pooled_width), float(spatial_scale),
                                             _lib.current_stream_handle(grad_output.device))
    _lib.check(rc, "mi_roi_pool_backward")
    return grad_input


class _RoIPool(Function):
    """`.apply(features, rois, pooled_height, pooled_width, spatial_scale)`"""

    @staticmethod
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale):
        rois = rois.contiguous()
        output, argmax = roi_pool_forward(features, rois, pooled_height, pooled_width, spatial_scale)
        ctx.save_for_backward(rois, argmax)
        ctx.cfg = (int(pooled_height), int(pooled_width), float(spatial_scale))
        ctx.feature_size = tuple(features.shape)
        ctx.mark_non_differentiable(argmax)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        rois, argmax = ctx.saved_tensors
        ph, pw, scale = ctx.cfg
        return roi_pool_backward(grad_output, rois, argmax, ctx.feature_size, ph, pw, scale), None, None, None, None


class RoIPoolFunction(object):
    """Drop-in for model.roi_pooling.functions.roi_pool.RoIPoolFunction; call site
    `RoIPoolFunction(resolution, resolution, sc)(bl_in, rois)` (lib/modeling/model_builder.py:279,312)."""

    def __init__(self, pooled_height, pooled_width, spatial_scale):
        self.pooled_width = pooled_width
        self.pooled_height = pooled_height
        self.spatial_scale = spatial_scale

    def __call__(self, features, rois):
        return _RoIPool.apply(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale)


class _RoIPooling(Module):
    """model/roi_pooling/modules/roi_pool.py:5-14"""

    def __init__(self, pooled_height, pooled_width, spatial_scale):
        super(_RoIPooling, self).__init__()
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return RoIPoolFunction(self.pooled_height, self.pooled_width, self.spatial_scale)(features, rois)
