"""RoICrop (bilinear grid sampler) -- host-side mirror of lib/model/roi_crop/functions/roi_crop.py:7-21
and modules/roi_crop.py, backed by mi_roi_crop_*.

    RoICropFunction()(input1 [N,C,H,W], grid_yx [R,gh,gw,2])  ->  [R,C,gh,gw]

Kept: grid last dim is (y, x) in [-1, 1]; RoI r samples image r // (R // N); output positions whose
four neighbours all fall outside the image stay 0; the backward returns an all-zero gradient for
the grid (the reference allocates it zeroed and its CUDA kernel never writes it,
roi_crop_cuda_kernel.cu:111-194).
"""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

from . import _lib


def roi_crop_forward(input1, grid_yx):
    _lib.require_cuda(input1, "input1")
    if input1.dtype != torch.float32 or grid_yx.dtype != torch.float32:
        raise TypeError("RoICrop supports float32 only (as the reference)")
    if grid_yx.dim() != 4 or grid_yx.size(3) != 2:
        raise ValueError("grid must be [R, gh, gw, 2] with (y, x) in the last dimension")
    # functions/roi_crop.py:13-14
    assert grid_yx.get_device() == input1.get_device(), "output and input1 must on the same device"
    input1 = input1.contiguous()
    grid_yx = grid_yx.contiguous()
    n, c, h, w = input1.shape
    r, gh, gw, _ = grid_yx.shape
    output = torch.zeros((r, c, gh, gw), dtype=input1.dtype, device=input1.device)  # :11 zero_() is load-bearing
    with torch.cuda.device(input1.device):
        rc = _lib.lib().mi_roi_crop_forward(input1.data_ptr(), grid_yx.data_ptr(), output.data_ptr(), n, c, h, w,
                                            r, gh, gw, _lib.current_stream_handle(input1.device))
    _lib.check(rc, "mi_roi_crop_forward")
    return output


def roi_crop_backward(input1, grid_yx, grad_output):
    grad_output = grad_output.contiguous()
    n, c, h, w = input1.shape
    r, gh, gw, _ = grid_yx.shape
    # the tile kernel writes every element: no zero fill (functions/roi_crop.py:18), no atomics
    grad_input1 = torch.empty_like(input1, memory_format=torch.contiguous_format)
    lib = _lib.lib()
    workspace = torch.empty(lib.mi_roi_crop_backward_workspace_bytes(r), dtype=torch.uint8, device=input1.device)
    with torch.cuda.device(input1.device):
        rc = lib.mi_roi_crop_backward_ws(input1.data_ptr(), grid_yx.data_ptr(), grad_output.data_ptr(),
                                         grad_input1.data_ptr(), n, c, h, w, r, gh, gw, workspace.data_ptr(), workspace.numel(),
                                         _lib.current_stream_handle(input1.device))
    _lib.check(rc, "mi_roi_crop_backward_ws")
    return grad_input1


class _RoICrop(Function):
    """`.apply(input1, grid_yx)`"""

    @staticmethod
    def forward(ctx, input1, input2):
        input1 = input1.contiguous()
        input2 = input2.contiguous()
        ctx.save_for_backward(input1, input2)
        return roi_crop_forward(input1, input2)

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        grad_input1 = roi_crop_backward(input1, input2, grad_output)
        grad_input2 = torch.zeros_like(input2)  # functions/roi_crop.py:19, never written by the kernel
        return grad_input1, grad_input2


class RoICropFunction(object):
    """Drop-in for model.roi_crop.functions.roi_crop.RoICropFunction; call site
    `RoICropFunction()(bl_in, Variable(grid_yx).detach())` (lib/modeling/model_builder.py:286,317)."""

    def __call__(self, input1, input2):
        return _RoICrop.apply(input1, input2)


class _RoICrop_Module(Module):
    """model/roi_crop/modules/roi_crop.py"""

    def forward(self, input1, input2):
        return RoICropFunction()(input1, input2)
