"""Small layers and initialisers of the reference's `lib/nn` that the graph needs."""
import math

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
        self.up_scale = s = int(up_scale)
        # A transposed convolution of stride s with the separable triangle filter of support 2 s is bilinear up-sampling
        # by s: tap k of the 1-D filter weighs 1 - |k - (s - 1/2)| / s, i.e. (1, 3, 5, ..., 5, 3, 1) / (2 s); the 2-D
        # kernel is its outer product, on the diagonal of the [C, C] channel pairs (every channel is up-sampled on its own).
        tri = 1.0 - (torch.arange(2 * s, dtype=torch.float64) - (s - 0.5)).abs() / s
        self.upconv = nn.ConvTranspose2d(in_channels, out_channels, 2 * s, stride=s, padding=s // 2)
        with torch.no_grad():
            self.upconv.weight.zero_()
            diag = torch.arange(in_channels)
            self.upconv.weight[diag, diag] = torch.outer(tri, tri).to(self.upconv.weight.dtype)
            self.upconv.bias.zero_()
        for p in self.upconv.parameters():   # frozen: the checkpoint name `upconv.*` is what the weight mapping expects
            p.requires_grad = False

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
