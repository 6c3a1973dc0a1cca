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


# ---- convolution bias (+ residual) (+ ReLU) in the same pass --------------------------------------------------------------
# PyTorch-ROCm adds a convolution's bias with a broadcast `add_` of its own and the model's ReLU (FPN.py:394,
# mask_rcnn_heads.py:160-185) / the FPN's top-down sum (FPN.py:292-296) are further element-wise kernels over the whole
# activation.  The convolution is asked for its bias-free result and the AffineChannel kernel (weight == 1: x * 1 is exact)
# finishes it in one pass: relu?(conv(x) + b[c] (+ r)), bit-identical to the unfused expressions.  The bias IS trainable
# here: its gradient is the sum of the (ReLU-masked) output gradient over N, H, W, as the convolution's own backward forms it.
_ONES = {}
_FUSE_CONV_BIAS = __import__("os").environ.get("MI_FUSED_CONV_BIAS", "1") == "1"   # A/B switch of tools/ (read at import)


def _ones(c, device):
    key = (c, device)
    if key not in _ONES:
        _ONES[key] = torch.ones(c, dtype=torch.float32, device=device)
    return _ONES[key]


class _BiasAct(Function):
    """`.apply(x, bias, residual_or_None, relu)`"""

    @staticmethod
    def forward(ctx, x, bias, residual, relu):
        layout = _layout_of(x)
        if residual is not None:
            residual = _dense_like(residual, layout)
        ones = _ones(x.size(1), x.device)
        y = affine_forward(x, ones, bias.detach().contiguous(), residual, relu, layout)
        ctx.relu, ctx.layout, ctx.has_residual = bool(relu), layout, residual is not None
        ctx.save_for_backward(ones, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        ones, y = ctx.saved_tensors
        if ctx.relu:
            grad_y = _dense_like(grad_y, ctx.layout)
            grad_x, _ = affine_backward(grad_y, y, ones, False, True, ctx.layout)
        else:
            grad_x = grad_y
        grad_b = grad_x.sum((0, 2, 3)) if ctx.needs_input_grad[1] else None
        return grad_x, grad_b, (grad_x if ctx.has_residual and ctx.needs_input_grad[2] else None), None


def conv_bias_act(conv, x, relu=False, residual=None):
    """relu?(conv(x) (+ residual)) for an `nn.Conv2d` / `nn.ConvTranspose2d` with a bias: the module's parameters, the fused
    epilogue.  Falls back to the module's own forward (+ torch add / relu) for what the kernel does not serve (CPU, bf16
    under autocast, no bias, exotic layouts)."""
    ok = (conv.bias is not None and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
          and conv.weight.dtype == torch.float32 and _FUSE_CONV_BIAS)
    if ok:
        if isinstance(conv, torch.nn.ConvTranspose2d):
            y = F.conv_transpose2d(x, conv.weight, None, conv.stride, conv.padding, conv.output_padding, conv.groups, conv.dilation)
        else:
            y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if _layout_of(y) is not None and (residual is None or (residual.shape == y.shape and residual.dtype == y.dtype)):
            return _BiasAct.apply(y, conv.bias, residual, bool(relu))
        y = y + conv.bias.view(1, -1, 1, 1)
    else:
        y = conv(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y
