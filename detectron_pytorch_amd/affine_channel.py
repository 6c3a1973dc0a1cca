"""AffineChannel (+ residual) (+ ReLU) in one pass -- host side of mi_affine_channel_forward / _backward.

Reference: lib/nn/modules/affine.py:5-17 (`x * weight.view(1, C, 1, 1) + bias.view(1, C, 1, 1)`) and what follows it in
lib/modeling/ResNet.py: ReLU (:270-277, stem :206-213), "out += residual; relu" (:284-286) or nothing (shortcut, :191-199).
In PyTorch each of these is its own element-wise kernel over the whole activation; the HIP kernel reads and writes the
activation once and gives bit-identical numbers (multiply, add bias, add residual, clamp -- no FMA).

    affine_channel(x, weight, bias, residual=None, relu=False)

The fused kernel serves what the training / inference step actually feeds it: float32 GPU tensors, dense NCHW or
channels_last, frozen weight and bias (every reference configuration, ResNet.py:76-77).  Anything else -- CPU tensors,
autocast's bf16 activations, trainable affine parameters -- takes the torch expressions of the reference, which is an
equivalent formulation, not a stand-in for a missing library: a GPU fp32 call without the built library raises.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib


def _layout_of(x):
    """LAYOUT_NCHW / LAYOUT_NHWC of a dense 4-d tensor, None when it is neither."""
    if x.is_contiguous():
        return _lib.LAYOUT_NCHW
    if x.is_contiguous(memory_format=torch.channels_last) and x.size(1) % 4 == 0:
        return _lib.LAYOUT_NHWC
    return None


def fused_supported(x, weight, bias, residual=None):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dtype == torch.float32):
        return False
    if weight.requires_grad or bias.requires_grad or _layout_of(x) is None:
        return False
    return residual is None or (residual.shape == x.shape and residual.dtype == x.dtype and residual.is_cuda)


def _dense_like(t, layout):
    return t.contiguous() if layout == _lib.LAYOUT_NCHW else t.contiguous(memory_format=torch.channels_last)


def affine_forward(x, weight, bias, residual, relu, layout):
    n, c, h, w = x.shape
    y = torch.empty_like(x)          # preserve_format: same strides as x
    with torch.cuda.device(x.device):
        rc = _lib.lib().mi_affine_channel_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                  residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                                  n, c, h, w, int(relu), layout, _lib.current_stream_handle(x.device))
    _lib.check(rc, "mi_affine_channel_forward")
    return y


def affine_backward(grad_y, y, weight, want_residual, relu, layout):
    n, c, h, w = grad_y.shape
    grad_x = torch.empty_like(grad_y)
    grad_r = torch.empty_like(grad_y) if want_residual else None
    with torch.cuda.device(grad_y.device):
        rc = _lib.lib().mi_affine_channel_backward(grad_y.data_ptr(), y.data_ptr() if y is not None else None,
                                                   weight.data_ptr(), grad_x.data_ptr(),
                                                   grad_r.data_ptr() if want_residual else None, n, c, h, w, int(relu),
                                                   layout, _lib.current_stream_handle(grad_y.device))
    _lib.check(rc, "mi_affine_channel_backward")
    return grad_x, grad_r


class _AffineChannel(Function):
    """`.apply(x, weight, bias, residual_or_None, relu)`; weight / bias are constants of the graph."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, relu):
        layout = _layout_of(x)
        if residual is not None:
            residual = _dense_like(residual, layout)
        weight, bias = weight.contiguous(), bias.contiguous()
        y = affine_forward(x, weight, bias, residual, relu, layout)
        ctx.relu, ctx.layout, ctx.has_residual = bool(relu), layout, residual is not None
        ctx.save_for_backward(weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        weight, y = ctx.saved_tensors
        grad_y = _dense_like(grad_y, ctx.layout)
        grad_x, grad_r = affine_backward(grad_y, y, weight, ctx.has_residual and ctx.needs_input_grad[3], ctx.relu,
                                         ctx.layout)
        return grad_x, None, None, grad_r, None


def affine_channel(x, weight, bias, residual=None, relu=False):
    """relu?(x * weight[c] + bias[c] (+ residual)) over [N, C, H, W]."""
    if fused_supported(x, weight, bias, residual):
        return _AffineChannel.apply(x, weight, bias, residual, bool(relu))
    c = weight.numel()
    out = x * weight.view(1, c, 1, 1) + bias.view(1, c, 1, 1)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out
