"""Test-time detection of one image batch, on the device from the image blob to the final detections (reference:
lib/core/test.py:50-125 `im_detect_all`, :127-190 `im_detect_bbox`, :732-790 `box_results_with_nms_and_limit`).

The reference copies rois / scores / deltas to the host after the forward (test.py:155-166) and finishes in numpy
(bbox_transform, clip, 80 x cython NMS).  Here the same arithmetic -- in the types numpy gives it -- stays in HBM and
ends in `detection.box_results_with_nms_and_limit` (one batched HIP NMS over all classes).
"""
import torch

from .. import detection


def bbox_transform(boxes, deltas, weights, clip):
    """utils/boxes.py:156-196 for fp32 inputs, numpy >= 2 types: centres in float32; widths / heights through
    `np.minimum(dw, cfg.BBOX_XFORM_CLIP)`, whose np.float64 clip constant promotes that branch (and the sums it enters)
    to float64; the float32 result array rounds on assignment.  boxes [R,4], deltas [R,4K] -> [R,4K]."""
    if boxes.size(0) == 0:
        return torch.zeros((0, deltas.size(1)), dtype=deltas.dtype, device=deltas.device)
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw = torch.clamp_max((deltas[:, 2::4] / ww).double(), clip)
    dh = torch.clamp_max((deltas[:, 3::4] / wh).double(), clip)
    pred_ctr_x = (dx * widths[:, None] + ctr_x[:, None]).double()
    pred_ctr_y = (dy * heights[:, None] + ctr_y[:, None]).double()
    pred_w = torch.exp(dw) * widths[:, None].double()
    pred_h = torch.exp(dh) * heights[:, None].double()
    out = torch.empty_like(deltas)
    out[:, 0::4] = (pred_ctr_x - 0.5 * pred_w).to(deltas.dtype)
    out[:, 1::4] = (pred_ctr_y - 0.5 * pred_h).to(deltas.dtype)
    out[:, 2::4] = (pred_ctr_x + 0.5 * pred_w - 1).to(deltas.dtype)
    out[:, 3::4] = (pred_ctr_y + 0.5 * pred_h - 1).to(deltas.dtype)
    return out


def clip_tiled_boxes(boxes, height, width):
    """utils/boxes.py:138-153: clip every (x1,y1,x2,y2) group to the image."""
    out = boxes.clone()
    out[:, 0::2] = out[:, 0::2].clamp(0, width - 1)
    out[:, 1::2] = out[:, 1::2].clamp(0, height - 1)
    return out


@torch.no_grad()
def im_detect_bbox(model, data, im_info, im_shape=None, autocast_dtype=None):
    """test.py:127-190 for one image blob [1,3,H,W] already resident on the device; `im_shape` = (height, width) of the
    ORIGINAL image (defaults to the blob's extent / scale).  Returns (scores [R,K], pred_boxes [R,4K], blob_conv)."""
    cfg = model.cfg
    if autocast_dtype is not None:
        with torch.autocast("cuda", dtype=autocast_dtype):
            ret = model(data, im_info)
    else:
        ret = model(data, im_info)
    scale = float(im_info[0][2])
    boxes = ret["rois"][:, 1:5] / scale
    scores = ret["cls_score"].float().reshape(-1, ret["cls_score"].shape[-1])
    deltas = ret["bbox_pred"].float().reshape(-1, ret["bbox_pred"].shape[-1])
    if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG:
        deltas = deltas[:, -4:]
    pred = bbox_transform(boxes, deltas, cfg.MODEL.BBOX_REG_WEIGHTS, cfg.BBOX_XFORM_CLIP)
    if im_shape is None:
        im_shape = (float(im_info[0][0]) / scale, float(im_info[0][1]) / scale)
    pred = clip_tiled_boxes(pred, im_shape[0], im_shape[1])
    if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG:
        pred = pred.repeat(1, scores.shape[1])
    return scores, pred, ret["blob_conv"]


@torch.no_grad()
def im_detect_all(model, data, im_info, im_shape=None, autocast_dtype=None):
    """test.py:50-125 without test-time augmentation and without the mask / keypoint branches (BASELINE config 3 is
    Faster R-CNN): detections of one image as (scores [D], boxes [D,4], cls_boxes) -- device tensors."""
    cfg = model.cfg
    scores, boxes, blob_conv = im_detect_bbox(model, data, im_info, im_shape, autocast_dtype)
    t = cfg.TEST
    return detection.box_results_with_nms_and_limit(
        scores, boxes, score_thresh=t.SCORE_THRESH, nms_thresh=t.NMS, detections_per_im=t.DETECTIONS_PER_IM,
        soft_nms=t.SOFT_NMS.ENABLED, soft_nms_sigma=t.SOFT_NMS.SIGMA, soft_nms_method=t.SOFT_NMS.METHOD)
