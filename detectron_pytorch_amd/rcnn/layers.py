"""Small layers and initialisers of the reference's `lib/nn` that the graph needs."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.init as init

from ..affine_channel import affine_channel


class AffineChannel2d(nn.Module):
    """Frozen-BN stand-in `x * w[c] + b[c]` (lib/nn/modules/affine.py:5-17).  Weight ~ U(0,1), bias 0, drawn in that
    order so that a seeded build consumes the generator exactly as the reference does."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.empty(num_features).uniform_())
        self.bias = nn.Parameter(torch.zeros(num_features))

    def forward(self, x, residual=None, relu=False):
        """`residual` / `relu`: what ResNet.py applies right after this layer (:270-286), folded into the same pass over
        the activation (csrc/affine_channel.hip) when the tensors are float32 on the GPU."""
        return affine_channel(x, self.weight, self.bias, residual, relu)


def xavier_fill(tensor):
    """Caffe2 XavierFill (lib/nn/init.py:11-16): U(-s, s), s = sqrt(3 / fan_in), fan_in = numel / shape[0]."""
    fan_in = tensor.numel() / tensor.shape[0]
    scale = math.sqrt(3 / fan_in)
    return init.uniform_(tensor, -scale, scale)


def msra_fill(tensor):
    """Caffe2 MSRAFill (lib/nn/init.py:19-24): N(0, sqrt(2 / fan_out)), fan_out = numel / shape[1]."""
    fan_out = tensor.numel() / tensor.shape[1]
    return init.normal_(tensor, 0, math.sqrt(2 / fan_out))


class BilinearInterpolation2d(nn.Module):
    """Fixed bilinear up-sampling as a frozen ConvTranspose2d (lib/nn/modules/upsample.py:9-53; FCN 'surgery' filter)."""

    def __init__(self, in_channels, out_channels, up_scale):
        super().__init__()
        assert in_channels == out_channels and up_scale % 2 == 0
        self.up_scale = int(up_scale)
        size = self.up_scale * 2
        factor = (size + 1) // 2
        centre = factor - 1 if size % 2 == 1 else factor - 0.5
        og = np.ogrid[:size, :size]
        filt = (1 - abs(og[0] - centre) / factor) * (1 - abs(og[1] - centre) / factor)
        kernel = np.zeros((in_channels, out_channels, size, size), dtype=np.float32)
        kernel[range(in_channels), range(out_channels), :, :] = filt
        self.upconv = nn.ConvTranspose2d(in_channels, out_channels, size, stride=self.up_scale, padding=self.up_scale // 2)
        self.upconv.weight.data.copy_(torch.from_numpy(kernel))
        self.upconv.bias.data.fill_(0)
        self.upconv.weight.requires_grad = False
        self.upconv.bias.requires_grad = False

    def forward(self, x):
        return self.upconv(x)


def smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, beta=1.0, num_rows=None):
    """lib/utils/net.py:15-32: sum(alpha_out * SmoothL1(alpha_in * (pred - target))) / N with N = rows of the prediction
    (`num_rows`: the count of real rows when the batch is padded to a static shape)."""
    diff = bbox_inside_weights * (bbox_pred - bbox_targets)
    abs_diff = diff.abs()
    near = (abs_diff < beta).detach().float()
    per_elem = near * 0.5 * diff * diff / beta + (1 - near) * (abs_diff - 0.5 * beta)
    total = (bbox_outside_weights * per_elem).view(-1).sum(0)
    return total / (bbox_pred.size(0) if num_rows is None else num_rows)
