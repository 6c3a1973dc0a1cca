"""Test-time detection of one image batch, on the device from the image blob to the final detections (reference:
lib/core/test.py:50-125 `im_detect_all`, :127-190 `im_detect_bbox`, :732-790 `box_results_with_nms_and_limit`).

The reference copies rois / scores / deltas to the host after the forward (test.py:155-166) and finishes in numpy
(bbox_transform, clip, 80 x cython NMS).  Here the same arithmetic -- in the types numpy gives it -- stays in HBM and
ends in `detection.box_results_with_nms_and_limit` (one batched HIP NMS over all classes).
"""
import torch

from .. import detection, fpn_proposals, hostcpu
from . import results

_checked_cpu_budget = []


def _check_cpu_budget():
    """Once per process: warn when torch's intra-op pool exceeds the container's CPU quota (eager launches then stall
    periodically, hostcpu.py)."""
    if not _checked_cpu_budget:
        _checked_cpu_budget.append(True)
        hostcpu.warn_if_oversubscribed()


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
    """utils/boxes.py:138-153: clip every (x1,y1,x2,y2) group to the image.  `height` / `width`: numbers, or 0-d device
    tensors (the static path keeps the image size on the device)."""
    out = boxes.clone()
    if torch.is_tensor(width):
        zero = torch.zeros((), dtype=boxes.dtype, device=boxes.device)
        out[:, 0::2] = torch.minimum(torch.maximum(out[:, 0::2], zero), (width - 1).to(boxes.dtype))
        out[:, 1::2] = torch.minimum(torch.maximum(out[:, 1::2], zero), (height - 1).to(boxes.dtype))
        return out
    out[:, 0::2] = out[:, 0::2].clamp(0, width - 1)
    out[:, 1::2] = out[:, 1::2].clamp(0, height - 1)
    return out


@torch.no_grad()
def im_detect_bbox(model, data, im_info, im_shape=None, autocast_dtype=None, static=False):
    """test.py:127-190 for one image blob [1,3,H,W] already resident on the device; `im_shape` = (height, width) of the
    ORIGINAL image (defaults to the blob's extent / scale).  Returns (scores [R,K], pred_boxes [R,4K], blob_conv).
    `static`: `im_info` is a device tensor and stays there (scale and image size are read by the kernels, not by the
    host), R = TEST.RPN_POST_NMS_TOP_N whatever the image; a fourth value `rois_valid` [R] is returned."""
    cfg = model.cfg
    was_static, model.static_inference = model.static_inference, bool(static)
    try:
        if autocast_dtype is not None:
            with torch.autocast("cuda", dtype=autocast_dtype):
                ret = model(data, im_info)
        else:
            ret = model(data, im_info)
    finally:
        model.static_inference = was_static
    scale = im_info[0, 2] if static else float(im_info[0][2])
    boxes = ret["rois"][:, 1:5] / scale
    scores = ret["cls_score"].float().reshape(-1, ret["cls_score"].shape[-1])
    deltas = ret["bbox_pred"].float().reshape(-1, ret["bbox_pred"].shape[-1])
    if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG:
        deltas = deltas[:, -4:]
    pred = bbox_transform(boxes, deltas, cfg.MODEL.BBOX_REG_WEIGHTS, cfg.BBOX_XFORM_CLIP)
    if im_shape is None:
        # the reference clips to the INTEGER shape of the original image (core/test.py:178); blob extent / scale is that
        # shape up to rounding (800 / 1.873536 = 427.0003), so it is rounded -- on the device in the static form
        im_shape = (torch.round(im_info[0, 0] / scale), torch.round(im_info[0, 1] / scale)) if static else \
            (float(round(float(im_info[0][0]) / scale)), float(round(float(im_info[0][1]) / scale)))
    pred = clip_tiled_boxes(pred, im_shape[0], im_shape[1])
    if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG:
        pred = pred.repeat(1, scores.shape[1])
    if static:
        return scores, pred, ret["blob_conv"], ret["rois_valid"]
    return scores, pred, ret["blob_conv"]


@torch.no_grad()
def im_detect_all(model, data, im_info, im_shape=None, autocast_dtype=None):
    """test.py:50-125 without test-time augmentation and without the mask / keypoint branches (BASELINE config 3 is
    Faster R-CNN): detections of one image as (scores [D], boxes [D,4], cls_boxes) -- device tensors.
    `im_shape` = (height, width) of the ORIGINAL image, what the reference clips the boxes to (core/test.py:178): pass it
    whenever the loader has it.  None reconstructs it as round(blob extent / scale) -- exact for the up-scaled images of the
    shipped configurations, but off by one pixel for some down-scaled ones (641 px at scale 0.5 -> blob 320 -> 640)."""
    _check_cpu_budget()
    cfg = model.cfg
    scores, boxes, blob_conv = im_detect_bbox(model, data, im_info, im_shape, autocast_dtype)
    t = cfg.TEST
    return detection.box_results_with_nms_and_limit(
        scores, boxes, score_thresh=t.SCORE_THRESH, nms_thresh=t.NMS, detections_per_im=t.DETECTIONS_PER_IM,
        soft_nms=t.SOFT_NMS.ENABLED, soft_nms_sigma=t.SOFT_NMS.SIGMA, soft_nms_method=t.SOFT_NMS.METHOD,
        bbox_vote=t.BBOX_VOTE.ENABLED, bbox_vote_thresh=t.BBOX_VOTE.VOTE_TH, bbox_vote_method=t.BBOX_VOTE.SCORING_METHOD)


def _rois_blob(boxes, im_scale, cfg, name):
    """test.py:869-920 `_get_rois_blob` + `_add_multilevel_rois_for_test` for ONE image on the device: [R, 5] RoIs in
    blob coordinates (batch index 0) and their FPN levels (the fused RoIAlign takes the level vector; the per-level
    blobs and the restore permutation of the reference are not needed)."""
    rois = torch.cat([torch.zeros((boxes.size(0), 1), dtype=torch.float32, device=boxes.device),
                      boxes.float() * im_scale], dim=1)
    lvls = fpn_proposals.map_rois_to_fpn_levels(rois[:, 1:5], cfg.FPN.ROI_MIN_LEVEL, cfg.FPN.ROI_MAX_LEVEL)
    return {name: rois, name + "_levels": lvls}


def _pad_to_detection_cap(boxes, cfg):
    """The number of detections differs from image to image, and every new batch size of the head's convolutions makes
    MIOpen set up kernels for that shape (tens of milliseconds on the host).  The heads therefore always see
    TEST.DETECTIONS_PER_IM rows: the detections, then copies of the first one (per-RoI results do not depend on the other
    rows; callers slice the padding off).  (This is NOT the periodic 60-80 ms stall of the eager mask-inference numbers:
    that one also occurs with exactly 100 detections per image, inside the RPN head's convolutions.)"""
    cap, r = int(cfg.TEST.DETECTIONS_PER_IM), boxes.size(0)
    if r == 0 or r >= cap:
        return boxes
    return torch.cat([boxes, boxes[:1].expand(cap - r, boxes.size(1))], dim=0)


@torch.no_grad()
def im_detect_mask(model, im_scale, boxes, blob_conv):
    """test.py:365-401: class-specific soft masks [R, K, M, M] (probabilities) of the detected `boxes` [R, 4] (image
    coordinates, device tensor), from the backbone features `blob_conv` of the same image."""
    cfg = model.cfg
    m = cfg.MRCNN.RESOLUTION
    k = cfg.MODEL.NUM_CLASSES if cfg.MRCNN.CLS_SPECIFIC_MASK else 1
    if boxes.size(0) == 0:
        return torch.zeros((0, k, m, m), dtype=torch.float32, device=boxes.device)
    r = boxes.size(0)
    pred = model.mask_net(blob_conv, _rois_blob(_pad_to_detection_cap(boxes, cfg), im_scale, cfg, "mask_rois"))
    return pred.float().reshape(-1, k, m, m)[:r]


@torch.no_grad()
def im_detect_keypoints(model, im_scale, boxes, blob_conv):
    """test.py:528-563: keypoint heat-map logits [R, J, H, H] of the detected `boxes`."""
    cfg = model.cfg
    h = cfg.KRCNN.HEATMAP_SIZE
    if boxes.size(0) == 0:
        return torch.zeros((0, cfg.KRCNN.NUM_KEYPOINTS, h, h), dtype=torch.float32, device=boxes.device)
    r = boxes.size(0)
    pred = model.keypoint_net(blob_conv, _rois_blob(_pad_to_detection_cap(boxes, cfg), im_scale, cfg, "keypoint_rois"))
    return pred.float().reshape(-1, cfg.KRCNN.NUM_KEYPOINTS, h, h)[:r]


@torch.no_grad()
def im_detect_all_results(model, data, im_info, im_shape=None, autocast_dtype=None):
    """test.py:50-112 for one image blob (no test-time augmentation): boxes, then masks and keypoints of the detected
    boxes, in the reference's result formats: (cls_boxes, cls_segms, cls_keyps).  cls_boxes[j] [k_j, 5] device tensors;
    cls_segms[j] a list of COCO RLE dicts (None when MODEL.MASK_ON is off); cls_keyps[j] a list of [4, K] tensors (None
    when MODEL.KEYPOINTS_ON is off)."""
    _check_cpu_budget()
    cfg = model.cfg
    scale = float(im_info[0][2])
    if im_shape is None:
        im_shape = (int(round(float(im_info[0][0]) / scale)), int(round(float(im_info[0][1]) / scale)))
    scores, boxes, blob_conv = im_detect_bbox(model, data, im_info, im_shape, autocast_dtype)
    t = cfg.TEST
    _, boxes_out, cls_boxes = detection.box_results_with_nms_and_limit(
        scores, boxes, score_thresh=t.SCORE_THRESH, nms_thresh=t.NMS, detections_per_im=t.DETECTIONS_PER_IM,
        soft_nms=t.SOFT_NMS.ENABLED, soft_nms_sigma=t.SOFT_NMS.SIGMA, soft_nms_method=t.SOFT_NMS.METHOD,
        bbox_vote=t.BBOX_VOTE.ENABLED, bbox_vote_thresh=t.BBOX_VOTE.VOTE_TH, bbox_vote_method=t.BBOX_VOTE.SCORING_METHOD)
    cls_segms = cls_keyps = None
    if cfg.MODEL.MASK_ON:
        masks = im_detect_mask(model, scale, boxes_out, blob_conv)
        cls_segms = results.segm_results(cls_boxes, masks, boxes_out, im_shape[0], im_shape[1], cfg)
    if cfg.MODEL.KEYPOINTS_ON:
        heatmaps = im_detect_keypoints(model, scale, boxes_out, blob_conv)
        cls_keyps = results.keypoint_results(cls_boxes, heatmaps, boxes_out, cfg)
    return cls_boxes, cls_segms, cls_keyps


RLE_CAPACITY, RLE_STRING_CAPACITY = 1024, 4096      # per mask, static path (a noisier mask sends the image to the eager path)


@torch.no_grad()
def im_detect_all_static(model, data, im_info, autocast_dtype=None, mask_im_shape=None):
    """`im_detect_all` as one asynchronous sequence of fixed shapes: image blob and `im_info` ([1,3] float32) are device
    tensors, nothing is read back.  Returns detection.box_results_static's dict (TEST.SOFT_NMS / TEST.BBOX_VOTE:
    detection.box_results_static_general's, same layout).
    `mask_im_shape` = (height, width) of the original image: the mask branch runs too (model with MODEL.MASK_ON) -- the
    fixed-size detection rows go through the mask head (unused rows carry image index -1 and pool zeros) and
    `mi_mask_paste_rle`; the dict gains 'rle_counts', 'rle_sizes', 'rle_strings' (results.mask_rle_static)."""
    cfg = model.cfg
    scores, boxes, blob_conv, valid = im_detect_bbox(model, data, im_info, None, autocast_dtype, static=True)
    t = cfg.TEST
    if t.SOFT_NMS.ENABLED or t.BBOX_VOTE.ENABLED:     # core/test.py:753-773, still without a host round trip
        res = detection.box_results_static_general(
            scores, boxes, t.SCORE_THRESH, t.NMS, t.DETECTIONS_PER_IM, roi_valid=valid, soft_nms=t.SOFT_NMS.ENABLED,
            soft_nms_sigma=t.SOFT_NMS.SIGMA, soft_nms_method=t.SOFT_NMS.METHOD, bbox_vote=t.BBOX_VOTE.ENABLED,
            bbox_vote_thresh=t.BBOX_VOTE.VOTE_TH, bbox_vote_method=t.BBOX_VOTE.SCORING_METHOD)
    else:
        res = detection.box_results_static(scores, boxes, t.SCORE_THRESH, t.NMS, t.DETECTIONS_PER_IM, roi_valid=valid)
    if mask_im_shape is not None:
        m = cfg.MRCNN.RESOLUTION
        det_boxes, cls = res["dets"][:, :4], res["cls"].long()
        rows = cls > 0
        rois = torch.cat([torch.where(rows, torch.zeros_like(det_boxes[:, 0]), torch.full_like(det_boxes[:, 0], -1.0)).view(-1, 1),
                          det_boxes * im_info[0, 2]], dim=1)
        lvls = fpn_proposals.map_rois_to_fpn_levels(rois[:, 1:5], cfg.FPN.ROI_MIN_LEVEL, cfg.FPN.ROI_MAX_LEVEL)
        masks = model.mask_net(blob_conv, {"mask_rois": rois, "mask_rois_levels": lvls}).float()
        k = cfg.MODEL.NUM_CLASSES if cfg.MRCNN.CLS_SPECIFIC_MASK else 1
        masks = masks.reshape(-1, k, m, m)
        sel = masks[torch.arange(masks.size(0), device=masks.device), cls] if cfg.MRCNN.CLS_SPECIFIC_MASK else masks[:, 0]
        boxes_int = results.expand_boxes(det_boxes, (m + 2.0) / m).to(torch.int32)
        res["rle_counts"], res["rle_sizes"], res["rle_strings"] = results.mask_rle_static(
            sel, boxes_int, mask_im_shape[0], mask_im_shape[1], cfg.MRCNN.THRESH_BINARIZE, RLE_CAPACITY, RLE_STRING_CAPACITY)
    return res


class DetectionGraph(object):
    """One image's detection -- backbone, RPN, proposals, RoIAlign, box head, decode, per-class NMS, top-100 -- captured
    once as a hipGraph and replayed per image: ~400 launches become one, and the host's only work per image is copying
    the blob in and the result sizes out (the reference: five D2H copies, numpy and Cython between them,
    SURVEY.md section 3.1).  One graph per blob shape (FPN pads blobs to multiples of 32: a handful of shapes per
    dataset); `im_info` is a graph INPUT, so images of different scale share the graph of their blob shape."""

    def __init__(self, model, blob_shape, device, autocast_dtype=None, mask_im_shape=None):
        """`mask_im_shape` = (height, width) of the original images this graph serves: boxes AND masks in the reference's
        result formats (`__call__` then returns (cls_boxes, cls_segms)); the image size is baked into the captured mask
        kernel, so such a graph serves one image size."""
        assert not model.training and blob_shape[0] == 1
        assert mask_im_shape is None or model.cfg.MODEL.MASK_ON
        self.model, self.autocast_dtype, self.mask_im_shape = model, autocast_dtype, mask_im_shape
        self.data = torch.zeros(blob_shape, dtype=torch.float32, device=device)
        self.im_info = torch.zeros((1, 3), dtype=torch.float32, device=device)
        self.graph, self.out = None, None

    def capture(self, data, im_info, warmup=3):
        """`data`, `im_info`: a representative image (MIOpen picks its solvers during the eager warm-up runs)."""
        self.data.copy_(data)
        self.im_info.copy_(torch.as_tensor(im_info, dtype=torch.float32).view(1, 3))
        side = torch.cuda.Stream(device=self.data.device)
        side.wait_stream(torch.cuda.current_stream(self.data.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                im_detect_all_static(self.model, self.data, self.im_info, self.autocast_dtype, self.mask_im_shape)
        torch.cuda.current_stream(self.data.device).wait_stream(side)
        torch.cuda.synchronize(self.data.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):     # the warm-up stream: one stream, a linear graph
            self.out = im_detect_all_static(self.model, self.data, self.im_info, self.autocast_dtype, self.mask_im_shape)
        self.graph = graph
        return self

    def replay(self, data, im_info):
        """Asynchronous: returns the static result dict (valid until the next replay)."""
        self.data.copy_(data, non_blocking=True)
        self.im_info.copy_(torch.as_tensor(im_info, dtype=torch.float32).view(1, 3), non_blocking=True)
        self.graph.replay()
        return self.out

    def __call__(self, data, im_info):
        """(scores [D], boxes [D,4], cls_boxes) like `im_detect_all`; one device-to-host copy (the result sizes).  With
        `mask_im_shape`: (cls_boxes, cls_segms) like `im_detect_all_results`."""
        if self.mask_im_shape is not None:
            return self._call_with_masks(data, im_info)
        out = detection._results_from_static(self.replay(data, im_info), False)
        if out is None:      # more tied scores at the detections_per_im cut than the static result holds
            return im_detect_all(self.model, data, torch.as_tensor(im_info).cpu().view(1, 3), None, self.autocast_dtype)
        # the views point into the graph's output buffer: hand out copies
        return out[0].clone(), out[1].clone(), [[]] + [c.clone() for c in out[2][1:]]

    def _call_with_masks(self, data, im_info):
        res = self.replay(data, im_info)
        out = detection._results_from_static(res, False)
        sizes = res["rle_sizes"].cpu().numpy()
        count = 0 if out is None else int(out[0].numel())
        if out is None or (count and (sizes[0, :count].max() > RLE_CAPACITY or sizes[1, :count].max() > RLE_STRING_CAPACITY)):
            # ties beyond the static result, or a mask with more runs than the static buffers hold: the eager path
            cls_boxes, cls_segms, _ = im_detect_all_results(self.model, data, torch.as_tensor(im_info).cpu().view(1, 3),
                                                            self.mask_im_shape, self.autocast_dtype)
            return cls_boxes, cls_segms
        cls_boxes = [[]] + [c.clone() for c in out[2][1:]]
        raw = res["rle_strings"][:count].cpu().numpy()
        h, w = int(self.mask_im_shape[0]), int(self.mask_im_shape[1])
        cls_segms, ind = [[] for _ in cls_boxes], 0
        for j in range(1, len(cls_boxes)):
            for _ in range(len(cls_boxes[j])):
                cls_segms[j].append({"size": [h, w], "counts": raw[ind, :sizes[1, ind]].tobytes().decode("ascii")})
                ind += 1
        return cls_boxes, cls_segms
