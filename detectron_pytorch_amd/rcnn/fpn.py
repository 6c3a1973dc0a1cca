"""Feature pyramid, the RPN head shared by all pyramid levels, and the RPN losses (reference: lib/modeling/FPN.py:73-258
`fpn`, :261-296 `topdown_lateral_module`, :324-419 `fpn_rpn_outputs`, :422-463 `fpn_rpn_losses`).

Outputs are ordered coarsest level first ([P6,] P5, P4, P3, P2), as in the reference (FPN.py:76-78).  The proposal half of
`fpn_rpn_outputs.forward` (GenerateProposals per level + CollectAndDistribute, :406-417) runs on the device through
`generate_proposals.GenerateProposalsOp` / `fpn_proposals` of this package -- no device-to-host copy.
"""
import collections

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from ..affine_channel import conv_bias_act
from ..generate_proposals import GenerateProposalsOp, generate_anchors
from . import resnet
from .layers import smooth_l1_loss, xavier_fill

LOWEST_BACKBONE_LVL, HIGHEST_BACKBONE_LVL = 2, 5          # FPN.py:20-21

FpnLevelInfo = collections.namedtuple("FpnLevelInfo", ["blobs", "dims", "spatial_scales"])   # FPN.py:470-473

LEVEL_INFO = {   # FPN.py:476-497: backbone blobs feeding the pyramid, coarsest first
    "ResNet50_conv5_body": FpnLevelInfo(("res5_2_sum", "res4_5_sum", "res3_3_sum", "res2_2_sum"),
                                        (2048, 1024, 512, 256), (1. / 32., 1. / 16., 1. / 8., 1. / 4.)),
    "ResNet101_conv5_body": FpnLevelInfo(("res5_2_sum", "res4_22_sum", "res3_3_sum", "res2_2_sum"),
                                         (2048, 1024, 512, 256), (1. / 32., 1. / 16., 1. / 8., 1. / 4.)),
    "ResNet152_conv5_body": FpnLevelInfo(("res5_2_sum", "res4_35_sum", "res3_7_sum", "res2_2_sum"),
                                         (2048, 1024, 512, 256), (1. / 32., 1. / 16., 1. / 8., 1. / 4.)),
}


def get_min_max_levels(cfg):
    """FPN.py:299-316."""
    min_level, max_level = LOWEST_BACKBONE_LVL, HIGHEST_BACKBONE_LVL
    if cfg.FPN.MULTILEVEL_RPN and not cfg.FPN.MULTILEVEL_ROIS:
        max_level, min_level = cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.RPN_MIN_LEVEL
    if not cfg.FPN.MULTILEVEL_RPN and cfg.FPN.MULTILEVEL_ROIS:
        max_level, min_level = cfg.FPN.ROI_MAX_LEVEL, cfg.FPN.ROI_MIN_LEVEL
    if cfg.FPN.MULTILEVEL_RPN and cfg.FPN.MULTILEVEL_ROIS:
        max_level = max(cfg.FPN.RPN_MAX_LEVEL, cfg.FPN.ROI_MAX_LEVEL)
        min_level = min(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.ROI_MIN_LEVEL)
    return min_level, max_level


class TopdownLateral(nn.Module):
    """FPN.py:261-296: 1x1 lateral conv of the backbone blob + nearest 2x up-sampling of the coarser pyramid blob."""

    def __init__(self, dim_in_top, dim_in_lateral, zero_init):
        super().__init__()
        self.dim_out = dim_in_top
        self.conv_lateral = nn.Conv2d(dim_in_lateral, self.dim_out, 1, 1, 0)
        if zero_init:
            init.constant_(self.conv_lateral.weight, 0)
        else:
            xavier_fill(self.conv_lateral.weight)
        init.constant_(self.conv_lateral.bias, 0)

    def forward(self, top_blob, lateral_blob):
        return conv_bias_act(self.conv_lateral, lateral_blob, residual=F.interpolate(top_blob, scale_factor=2, mode="nearest"))


class FPN(nn.Module):
    """FPN.py:73-258 on a ResNet conv5 body (`conv_body_name`: a key of resnet.BLOCK_COUNTS)."""

    def __init__(self, conv_body_name, cfg):
        super().__init__()
        if cfg.FPN.USE_GN or cfg.FPN.EXTRA_CONV_LEVELS:
            raise NotImplementedError("GroupNorm / RetinaNet extra-level pyramids (FPN.py:99-104,143-152) are out of scope")
        info = LEVEL_INFO[conv_body_name]
        self.fpn_level_info = info
        self.dim_out = fpn_dim = cfg.FPN.DIM
        min_level, max_level = get_min_max_levels(cfg)
        self.num_backbone_stages = len(info.blobs) - (min_level - LOWEST_BACKBONE_LVL)
        self.spatial_scale = []
        # FPN.py:97-106 constructs the seed 1x1 conv twice (the first instance is discarded); a seeded build must draw
        # the same numbers, so the throw-away construction is kept
        nn.Conv2d(info.dims[0], fpn_dim, 1, 1, 0)
        self.conv_top = nn.Conv2d(info.dims[0], fpn_dim, 1, 1, 0)
        self.topdown_lateral_modules = nn.ModuleList()
        self.posthoc_modules = nn.ModuleList()
        for i in range(self.num_backbone_stages - 1):
            self.topdown_lateral_modules.append(TopdownLateral(fpn_dim, info.dims[i + 1], cfg.FPN.ZERO_INIT_LATERAL))
        for i in range(self.num_backbone_stages):
            self.posthoc_modules.append(nn.Conv2d(fpn_dim, fpn_dim, 3, 1, 1))
            self.spatial_scale.append(info.spatial_scales[i])
        if max_level == HIGHEST_BACKBONE_LVL + 1:                 # P6 = stride-2 subsampling of P5 (FPN.py:132-137)
            self.maxpool_p6 = nn.MaxPool2d(kernel_size=1, stride=2, padding=0)
            self.spatial_scale.insert(0, self.spatial_scale[0] * 0.5)
        # FPN.py:159-170: Xavier for conv_top and the post-hoc convs (the lateral modules initialised themselves)
        for conv in [self.conv_top] + list(self.posthoc_modules):
            xavier_fill(conv.weight)
            init.constant_(conv.bias, 0)
        self.conv_body = resnet.ResNetBody(resnet.BLOCK_COUNTS[conv_body_name], cfg)   # after the init, as FPN.py:155-157

    def forward(self, x):
        body = self.conv_body
        blobs = [body.res1(x)]
        for i in range(1, body.convX):
            blobs.append(getattr(body, "res%d" % (i + 1))(blobs[-1]))
        inner = [conv_bias_act(self.conv_top, blobs[-1])]
        for i in range(self.num_backbone_stages - 1):
            inner.append(self.topdown_lateral_modules[i](inner[-1], blobs[-(i + 2)]))
        out = [conv_bias_act(self.posthoc_modules[i], inner[i]) for i in range(self.num_backbone_stages)]
        if hasattr(self, "maxpool_p6"):
            out.insert(0, self.maxpool_p6(out[0]))
        return out


class FpnRpnOutputs(nn.Module):
    """FPN.py:324-419: one 3x3 conv + objectness / box-delta 1x1 convs shared by all levels, and one proposal generator
    per level (anchors of size RPN_ANCHOR_START_SIZE * 2^(lvl - k_min), stride 2^lvl)."""

    def __init__(self, dim_in, spatial_scales, cfg):
        super().__init__()
        if cfg.RPN.CLS_ACTIVATION not in ("sigmoid", "softmax"):
            raise ValueError("RPN.CLS_ACTIVATION must be 'sigmoid' or 'softmax' (config.py:661-663)")
        self.cfg = cfg
        self.dim_in = self.dim_out = dim_in
        self.spatial_scales = spatial_scales
        num_anchors = len(cfg.FPN.RPN_ASPECT_RATIOS)
        self.FPN_RPN_conv = nn.Conv2d(dim_in, self.dim_out, 3, 1, 1)
        dim_score = num_anchors * 2 if cfg.RPN.CLS_ACTIVATION == "softmax" else num_anchors   # FPN.py:335-336
        self.FPN_RPN_cls_score = nn.Conv2d(self.dim_out, dim_score, 1, 1, 0)
        self.FPN_RPN_bbox_pred = nn.Conv2d(self.dim_out, 4 * num_anchors, 1, 1, 0)
        self.k_min, self.k_max = cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL
        self.level_anchors = []
        for lvl in range(self.k_min, self.k_max + 1):
            self.level_anchors.append(generate_anchors(stride=2. ** lvl,
                                                       sizes=(cfg.FPN.RPN_ANCHOR_START_SIZE * 2. ** (lvl - self.k_min),),
                                                       aspect_ratios=cfg.FPN.RPN_ASPECT_RATIOS))
        for conv in (self.FPN_RPN_conv, self.FPN_RPN_cls_score, self.FPN_RPN_bbox_pred):   # FPN.py:356-362
            init.normal_(conv.weight, std=0.01)
            init.constant_(conv.bias, 0)

    def proposal_ops(self, training):
        """The per-level GenerateProposalsOps with the TRAIN / TEST settings (generate_proposals.py:103-111)."""
        key = "TRAIN" if training else "TEST"
        c = self.cfg[key]
        return [GenerateProposalsOp(self.level_anchors[lvl - self.k_min], self.spatial_scales[self.k_max - lvl],
                                    c.RPN_PRE_NMS_TOP_N, c.RPN_POST_NMS_TOP_N, c.RPN_NMS_THRESH, c.RPN_MIN_SIZE,
                                    as_numpy=False)
                for lvl in range(self.k_min, self.k_max + 1)]

    def forward(self, blobs_in):
        """The convolutional half of FPN.py:376-405: {'rpn_cls_logits_fpn<l>', 'rpn_bbox_pred_fpn<l>'} for every level."""
        assert len(blobs_in) == self.k_max - self.k_min + 1
        ret = {}
        for lvl in range(self.k_min, self.k_max + 1):
            hidden = conv_bias_act(self.FPN_RPN_conv, blobs_in[self.k_max - lvl], relu=True)
            ret["rpn_cls_logits_fpn%d" % lvl] = conv_bias_act(self.FPN_RPN_cls_score, hidden)
            ret["rpn_bbox_pred_fpn%d" % lvl] = conv_bias_act(self.FPN_RPN_bbox_pred, hidden)
        return ret


def rpn_cls_probs(cfg, logits):
    """FPN.py:399-404: objectness of every anchor [N, A, H, W] from the classification logits -- sigmoid of the A logits, or
    (jwyang's convention, RPN.CLS_ACTIVATION = 'softmax') the foreground half of a softmax over the 2 x A logits."""
    if cfg.RPN.CLS_ACTIVATION == "softmax":
        b, c, h, w = logits.shape
        return F.softmax(logits.view(b, 2, c // 2, h, w), dim=1)[:, 1]
    return torch.sigmoid(logits)


def fpn_rpn_losses(cfg, rpn_ret, rpn_targets):
    """FPN.py:422-463 (both activations).  `rpn_targets` holds the data layer's "wide" blobs
    ('rpn_labels_int32_wide_fpn<l>' [N,A,F,F] int32 in {-1,0,1}, 'rpn_bbox_targets_wide_fpn<l>' and the two weight blobs
    [N,4A,F,F]); they are narrowed to each level's map.  Returns ([loss_cls per level], [loss_bbox per level])."""
    losses_cls, losses_bbox = [], []
    norm = cfg.TRAIN.RPN_BATCH_SIZE_PER_IM * cfg.TRAIN.IMS_PER_BATCH
    for lvl in range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1):
        s = str(lvl)
        logits, pred = rpn_ret["rpn_cls_logits_fpn" + s], rpn_ret["rpn_bbox_pred_fpn" + s]
        h, w = logits.shape[2:]
        labels = rpn_targets["rpn_labels_int32_wide_fpn" + s][:, :, :h, :w]
        if cfg.RPN.CLS_ACTIVATION == "softmax":                                   # FPN.py:438-445
            b, c = logits.shape[:2]
            pairs = logits.view(b, 2, c // 2, h, w).permute(0, 2, 3, 4, 1).contiguous().view(-1, 2)
            loss_cls = F.cross_entropy(pairs, labels.contiguous().view(-1).long(), ignore_index=-1)   # mean over non-ignored
        else:
            weight = (labels >= 0).float()
            loss_cls = F.binary_cross_entropy_with_logits(logits, labels.float(), weight, reduction="sum") / norm
        h, w = pred.shape[2:]
        loss_bbox = smooth_l1_loss(pred, rpn_targets["rpn_bbox_targets_wide_fpn" + s][:, :, :h, :w],
                                   rpn_targets["rpn_bbox_inside_weights_wide_fpn" + s][:, :, :h, :w],
                                   rpn_targets["rpn_bbox_outside_weights_wide_fpn" + s][:, :, :h, :w], beta=1 / 9)
        losses_cls.append(loss_cls)
        losses_bbox.append(loss_bbox)
    return losses_cls, losses_bbox
