"""RoI heads and their losses (reference: lib/modeling/fast_rcnn_heads.py:12-116, mask_rcnn_heads.py:20-188,
keypoint_rcnn_heads.py:17-179).  Every head starts with `roi_xform` -- the RoIFeatureTransform of this package
(roi_xform.roi_feature_transform -> HIP RoIAlign) bound by the model, the same call the reference makes
(fast_rcnn_heads.py:104-111).

Losses take device tensors; a batch padded to a static shape marks its padding rows with label -1 (box head) or
all-(-1) mask targets / zero keypoint weights, which the reference's own formulas already ignore (see each loss).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init

from ..affine_channel import conv_bias_act
from .layers import BilinearInterpolation2d, msra_fill, smooth_l1_loss, xavier_fill


# ---- box head -------------------------------------------------------------------------------------------------------
class Roi2MlpHead(nn.Module):
    """`roi_2mlp_head` (fast_rcnn_heads.py:73-116): RoIAlign 7x7 -> fc1 -> relu -> fc2 -> relu."""

    def __init__(self, dim_in, roi_xform_func, spatial_scale, cfg):
        super().__init__()
        self.cfg = cfg
        self.dim_in = dim_in
        self.roi_xform = roi_xform_func
        self.spatial_scale = spatial_scale
        self.dim_out = hidden = cfg.FAST_RCNN.MLP_HEAD_DIM
        res = cfg.FAST_RCNN.ROI_XFORM_RESOLUTION
        self.fc1 = nn.Linear(dim_in * res ** 2, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        xavier_fill(self.fc1.weight)
        init.constant_(self.fc1.bias, 0)
        xavier_fill(self.fc2.weight)
        init.constant_(self.fc2.bias, 0)

    def forward(self, x, rpn_ret):
        c = self.cfg.FAST_RCNN
        x = self.roi_xform(x, rpn_ret, blob_rois="rois", method=c.ROI_XFORM_METHOD, resolution=c.ROI_XFORM_RESOLUTION,
                           spatial_scale=self.spatial_scale, sampling_ratio=c.ROI_XFORM_SAMPLING_RATIO)
        x = F.relu(self.fc1(x.reshape(x.size(0), -1)), inplace=True)
        return F.relu(self.fc2(x), inplace=True)


class FastRcnnOutputs(nn.Module):
    """`fast_rcnn_outputs` (fast_rcnn_heads.py:12-47): class scores (softmax at test time) and per-class box deltas."""

    def __init__(self, dim_in, cfg):
        super().__init__()
        self.cls_score = nn.Linear(dim_in, cfg.MODEL.NUM_CLASSES)
        self.bbox_pred = nn.Linear(dim_in, 4 * (2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else cfg.MODEL.NUM_CLASSES))
        init.normal_(self.cls_score.weight, std=0.01)
        init.constant_(self.cls_score.bias, 0)
        init.normal_(self.bbox_pred.weight, std=0.001)
        init.constant_(self.bbox_pred.bias, 0)

    def forward(self, x):
        if x.dim() == 4:
            x = x.squeeze(3).squeeze(2)
        cls_score = self.cls_score(x)
        if not self.training:
            cls_score = F.softmax(cls_score, dim=1)
        return cls_score, self.bbox_pred(x)


def fast_rcnn_losses(cls_score, bbox_pred, labels, bbox_targets, bbox_inside_weights, bbox_outside_weights):
    """fast_rcnn_heads.py:50-70.  Rows with label -1 are padding: cross_entropy's mean runs over the real rows
    (`ignore_index`), their box weights are zero, and the smooth-L1 normaliser is the real row count."""
    labels = labels.long()
    real = (labels >= 0)
    num_rows = real.sum().clamp_min(1).to(cls_score.dtype)
    loss_cls = F.cross_entropy(cls_score, labels, ignore_index=-1)
    loss_bbox = smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, num_rows=num_rows)
    preds = cls_score.max(dim=1)[1]
    accuracy = ((preds == labels) & real).float().sum() / num_rows
    return loss_cls, loss_bbox, accuracy


# ---- mask head ------------------------------------------------------------------------------------------------------
class MaskRcnnFcnHeadV1upXconvs(nn.Module):
    """`mask_rcnn_fcn_head_v1upXconvs` (mask_rcnn_heads.py:127-188): RoIAlign 14x14 -> X x (3x3 conv, relu) ->
    2x2 stride-2 transposed conv -> relu."""

    def __init__(self, dim_in, roi_xform_func, spatial_scale, num_convs, cfg):
        super().__init__()
        self.cfg = cfg
        self.dim_in = dim_in
        self.roi_xform = roi_xform_func
        self.spatial_scale = spatial_scale
        self.num_convs = num_convs
        dilation, dim_inner = cfg.MRCNN.DILATION, cfg.MRCNN.DIM_REDUCED
        self.dim_out = dim_inner
        layers = []
        for _ in range(num_convs):
            layers += [nn.Conv2d(dim_in, dim_inner, 3, 1, padding=dilation, dilation=dilation), nn.ReLU(inplace=True)]
            dim_in = dim_inner
        self.conv_fcn = nn.Sequential(*layers)
        self.upconv = nn.ConvTranspose2d(dim_inner, dim_inner, 2, 2, 0)
        for m in list(self.conv_fcn) + [self.upconv]:             # the order self.apply(_init_weights) visits them
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if cfg.MRCNN.CONV_INIT == "GaussianFill":
                    init.normal_(m.weight, std=0.001)
                elif cfg.MRCNN.CONV_INIT == "MSRAFill":
                    msra_fill(m.weight)
                else:
                    raise ValueError(cfg.MRCNN.CONV_INIT)
                init.constant_(m.bias, 0)

    def forward(self, x, rpn_ret):
        c = self.cfg.MRCNN
        x = self.roi_xform(x, rpn_ret, blob_rois="mask_rois", method=c.ROI_XFORM_METHOD,
                           resolution=c.ROI_XFORM_RESOLUTION, spatial_scale=self.spatial_scale,
                           sampling_ratio=c.ROI_XFORM_SAMPLING_RATIO)
        for m in self.conv_fcn:   # [conv, ReLU] pairs: the ReLU modules are served by the convolution's fused epilogue
            if isinstance(m, nn.Conv2d):
                x = conv_bias_act(m, x, relu=True)
        return conv_bias_act(self.upconv, x, relu=True)


class MaskRcnnOutputs(nn.Module):
    """`mask_rcnn_outputs` (mask_rcnn_heads.py:20-66): 1x1 conv to per-class mask logits (sigmoid at test time)."""

    def __init__(self, dim_in, cfg):
        super().__init__()
        if cfg.MRCNN.USE_FC_OUTPUT:
            raise NotImplementedError("MRCNN.USE_FC_OUTPUT (mask_rcnn_heads.py:27-29) is not used by any shipped yaml")
        self.upsample_ratio = cfg.MRCNN.UPSAMPLE_RATIO
        n_classes = cfg.MODEL.NUM_CLASSES if cfg.MRCNN.CLS_SPECIFIC_MASK else 1
        self.classify = nn.Conv2d(dim_in, n_classes, 1, 1, 0)
        if self.upsample_ratio > 1:
            self.upsample = BilinearInterpolation2d(n_classes, n_classes, self.upsample_ratio)
        if cfg.MRCNN.CLS_SPECIFIC_MASK and cfg.MRCNN.CONV_INIT == "MSRAFill":
            msra_fill(self.classify.weight)
        else:
            init.normal_(self.classify.weight, std=0.001)
        init.constant_(self.classify.bias, 0)

    def forward(self, x):
        x = conv_bias_act(self.classify, x)
        if self.upsample_ratio > 1:
            x = self.upsample(x)
        return x if self.training else torch.sigmoid(x)


def mask_rcnn_losses(masks_pred, masks_int32, weight_loss_mask=1.0):
    """mask_rcnn_heads.py:90-99: BCE over the entries whose target is not -1, averaged over their count.  Padding rows
    are all -1 and therefore contribute to neither the sum nor the count."""
    n = masks_pred.size(0)
    masks_gt = masks_int32.to(masks_pred.dtype)
    weight = (masks_gt > -1).to(masks_pred.dtype)
    loss = F.binary_cross_entropy_with_logits(masks_pred.reshape(n, -1), masks_gt, weight, reduction="sum")
    return loss / weight.sum() * weight_loss_mask


def mask_rcnn_losses_compact(masks_pred, masks, mask_class, weight_loss_mask=1.0):
    """The same loss for class-specific masks without the reference's [n, K*M*M] target blob (mask_rcnn.py:110-129 fills
    it with -1 except the block of the row's class, so only that block ever has weight): gather the class channel of the
    prediction and compare it with the compact [n, M*M] target.  Identical sum and identical count -- 1/K of the
    traffic.  Rows with class 0 (padding, or the reference's "no fg RoI" dummy row) carry no weight."""
    n = masks_pred.size(0)
    idx = mask_class.long().clamp_min(0)
    pred = masks_pred[torch.arange(n, device=masks_pred.device), idx].reshape(n, -1)
    masks_gt = masks.to(pred.dtype)
    weight = ((masks_gt > -1) & (mask_class.view(-1, 1) > 0)).to(pred.dtype)
    loss = F.binary_cross_entropy_with_logits(pred, masks_gt.clamp_min(0), weight, reduction="sum")
    return loss / weight.sum() * weight_loss_mask


# ---- keypoint head --------------------------------------------------------------------------------------------------
class RoiPoseHeadV1convX(nn.Module):
    """`roi_pose_head_v1convX` (keypoint_rcnn_heads.py:129-179): RoIAlign 14x14 -> X x (conv, relu)."""

    def __init__(self, dim_in, roi_xform_func, spatial_scale, cfg):
        super().__init__()
        self.cfg = cfg
        self.dim_in = dim_in
        self.roi_xform = roi_xform_func
        self.spatial_scale = spatial_scale
        hidden, k = cfg.KRCNN.CONV_HEAD_DIM, cfg.KRCNN.CONV_HEAD_KERNEL
        layers = []
        for _ in range(cfg.KRCNN.NUM_STACKED_CONVS):
            layers += [nn.Conv2d(dim_in, hidden, k, 1, k // 2), nn.ReLU(inplace=True)]
            dim_in = hidden
        self.conv_fcn = nn.Sequential(*layers)
        self.dim_out = hidden
        for m in self.conv_fcn:
            if isinstance(m, nn.Conv2d):
                if cfg.KRCNN.CONV_INIT == "GaussianFill":
                    init.normal_(m.weight, std=0.01)
                elif cfg.KRCNN.CONV_INIT == "MSRAFill":
                    msra_fill(m.weight)
                init.constant_(m.bias, 0)

    def forward(self, x, rpn_ret):
        c = self.cfg.KRCNN
        x = self.roi_xform(x, rpn_ret, blob_rois="keypoint_rois", method=c.ROI_XFORM_METHOD,
                           resolution=c.ROI_XFORM_RESOLUTION, spatial_scale=self.spatial_scale,
                           sampling_ratio=c.ROI_XFORM_SAMPLING_RATIO)
        for m in self.conv_fcn:
            if isinstance(m, nn.Conv2d):
                x = conv_bias_act(m, x, relu=True)
        return x


class KeypointOutputs(nn.Module):
    """`keypoint_outputs` (keypoint_rcnn_heads.py:17-88): optional deconv, heat-map classifier (conv or 4x4 stride-2
    deconv), optional fixed bilinear up-sampling."""

    def __init__(self, dim_in, cfg):
        super().__init__()
        k = cfg.KRCNN
        self.use_deconv = k.USE_DECONV
        self.upsample_heatmap = k.UP_SCALE > 1
        if k.USE_DECONV:
            self.deconv = nn.ConvTranspose2d(dim_in, k.DECONV_DIM, k.DECONV_KERNEL, 2, padding=int(k.DECONV_KERNEL / 2) - 1)
            dim_in = k.DECONV_DIM
        if k.USE_DECONV_OUTPUT:
            self.classify = nn.ConvTranspose2d(dim_in, k.NUM_KEYPOINTS, k.DECONV_KERNEL, 2,
                                               padding=int(k.DECONV_KERNEL / 2 - 1))
        else:
            self.classify = nn.Conv2d(dim_in, k.NUM_KEYPOINTS, 1, 1, padding=0)
        if self.upsample_heatmap:
            self.upsample = BilinearInterpolation2d(k.NUM_KEYPOINTS, k.NUM_KEYPOINTS, k.UP_SCALE)
        if k.USE_DECONV:
            init.normal_(self.deconv.weight, std=0.01)
            init.constant_(self.deconv.bias, 0)
        if k.CONV_INIT == "GaussianFill":
            init.normal_(self.classify.weight, std=0.001)
        elif k.CONV_INIT == "MSRAFill":
            msra_fill(self.classify.weight)
        else:
            raise ValueError(k.CONV_INIT)
        init.constant_(self.classify.bias, 0)

    def forward(self, x):
        if self.use_deconv:
            x = conv_bias_act(self.deconv, x, relu=True)
        x = conv_bias_act(self.classify, x)
        return self.upsample(x) if self.upsample_heatmap else x


def keypoint_losses(kps_pred, keypoint_locations, keypoint_weights, heatmap_size, loss_weight=1.0, normalizer=None):
    """keypoint_rcnn_heads.py:91-121: softmax over the heat-map's spatial positions, weighted by visibility."""
    loss = F.cross_entropy(kps_pred.reshape(-1, heatmap_size ** 2), keypoint_locations.long(), reduction="none")
    loss = torch.sum(loss * keypoint_weights) / torch.sum(keypoint_weights) * loss_weight
    return loss if normalizer is None else loss * normalizer
