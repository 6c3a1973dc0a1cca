"""ResNet / ResNeXt bottleneck body with frozen AffineChannel layers (reference: lib/modeling/ResNet.py:42-116 body,
:166-184 stages, :191-215 shortcut / stem, :246-293 bottleneck incl. grouped 3x3 for ResNeXt).

Construction order (stem conv, stem affine; per block: shortcut conv + affine first, then conv1/bn1/conv2/bn2/conv3/bn3)
follows the reference so that a build under `torch.manual_seed(cfg.RNG_SEED)` draws identical initial weights.
Stock `nn.Conv2d` (MIOpen on ROCm): this is the one place where MFMA does the work.
"""
from collections import OrderedDict

import torch.nn as nn

from .layers import AffineChannel2d

BLOCK_COUNTS = {"ResNet50_conv5_body": (3, 4, 6, 3), "ResNet101_conv5_body": (3, 4, 23, 3),
                "ResNet152_conv5_body": (3, 8, 36, 3)}


def freeze_params(module):
    """ResNet.py:400-403."""
    for p in module.parameters():
        p.requires_grad = False


class Bottleneck(nn.Module):
    """`bottleneck_transformation` (ResNet.py:246-293): 1x1 -> 3x3 (groups) -> 1x1, each followed by an AffineChannel;
    the stride sits on the first 1x1 when RESNETS.STRIDE_1X1 (MSRA weights) and on the 3x3 otherwise (C2 / torch)."""

    def __init__(self, inplanes, outplanes, innerplanes, stride, dilation, groups, stride_1x1, downsample):
        super().__init__()
        str1x1, str3x3 = (stride, 1) if stride_1x1 else (1, stride)
        self.stride = stride
        self.conv1 = nn.Conv2d(inplanes, innerplanes, kernel_size=1, stride=str1x1, bias=False)
        self.bn1 = AffineChannel2d(innerplanes)
        self.conv2 = nn.Conv2d(innerplanes, innerplanes, kernel_size=3, stride=str3x3, bias=False, padding=dilation,
                               dilation=dilation, groups=groups)
        self.bn2 = AffineChannel2d(innerplanes)
        self.conv3 = nn.Conv2d(innerplanes, outplanes, kernel_size=1, stride=1, bias=False)
        self.bn3 = AffineChannel2d(outplanes)
        self.downsample = downsample
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=residual, relu=True)     # out += residual; relu (ResNet.py:284-286)


class Stem(nn.Sequential):
    """conv1 -> affine -> relu -> maxpool with the reference's child names; the affine layer applies the ReLU itself."""

    def forward(self, x):
        return self.maxpool(self.bn1(self.conv1(x), relu=True))


def make_stage(inplanes, outplanes, innerplanes, nblocks, cfg, dilation=1, stride_init=2):
    """`add_stage` + `add_residual_block` (ResNet.py:151-184)."""
    blocks, stride = [], stride_init
    for _ in range(nblocks):
        downsample = None
        if stride != 1 or inplanes != outplanes:   # basic_bn_shortcut, ResNet.py:191-199 (built before the block's convs)
            downsample = nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False),
                                       AffineChannel2d(outplanes))
        blocks.append(Bottleneck(inplanes, outplanes, innerplanes, stride, dilation, cfg.RESNETS.NUM_GROUPS,
                                 cfg.RESNETS.STRIDE_1X1, downsample))
        inplanes, stride = outplanes, 1
    return nn.Sequential(*blocks), outplanes


class ResNetBody(nn.Module):
    """`ResNet_convX_body` (ResNet.py:42-116) for conv5 bodies: res1 (stem) .. res5."""

    def __init__(self, block_counts, cfg):
        super().__init__()
        if cfg.RESNETS.TRANS_FUNC != "bottleneck_transformation" or cfg.RESNETS.STEM_FUNC != "basic_bn_stem" \
                or cfg.RESNETS.SHORTCUT_FUNC != "basic_bn_shortcut":
            raise NotImplementedError("only the AffineChannel (frozen-BN) ResNet variants are built; GroupNorm bodies "
                                      "(ResNet.py:201-243,296-350) are out of scope")
        self.block_counts = block_counts
        self.convX = len(block_counts) + 1
        self.freeze_at = cfg.RESNETS.FREEZE_AT
        self.res1 = Stem(OrderedDict([                               # basic_bn_stem, ResNet.py:206-213
            ("conv1", nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)),
            ("bn1", AffineChannel2d(64)),
            ("relu", nn.ReLU(inplace=True)),
            ("maxpool", nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))
        dim_in = 64
        width = cfg.RESNETS.NUM_GROUPS * cfg.RESNETS.WIDTH_PER_GROUP
        self.res2, dim_in = make_stage(dim_in, 256, width, block_counts[0], cfg, 1, 1)
        self.res3, dim_in = make_stage(dim_in, 512, width * 2, block_counts[1], cfg, 1, 2)
        self.res4, dim_in = make_stage(dim_in, 1024, width * 4, block_counts[2], cfg, 1, 2)
        stride_init = 2 if cfg.RESNETS.RES5_DILATION == 1 else 1
        self.res5, dim_in = make_stage(dim_in, 2048, width * 8, block_counts[3], cfg, cfg.RESNETS.RES5_DILATION,
                                       stride_init)
        self.spatial_scale = 1 / 32 * cfg.RESNETS.RES5_DILATION
        self.dim_out = dim_in
        # ResNet.py:70-77: stages up to FREEZE_AT and every AffineChannel are not trained
        assert self.freeze_at in (0, 2, 3, 4, 5) and self.freeze_at <= self.convX
        for i in range(1, self.freeze_at + 1):
            freeze_params(getattr(self, "res%d" % i))
        self.apply(lambda m: freeze_params(m) if isinstance(m, AffineChannel2d) else None)

    def train(self, mode=True):
        """ResNet.py:105-110: frozen stages stay in eval mode."""
        self.training = mode
        for i in range(self.freeze_at + 1, self.convX + 1):
            getattr(self, "res%d" % i).train(mode)
        return self

    def forward(self, x):
        for i in range(self.convX):
            x = getattr(self, "res%d" % (i + 1))(x)
        return x
