"""Proposal labelling for training, on the device and with static shapes: the training half of
`CollectAndDistributeFpnRpnProposalsOp`.

Reference path (all numpy on the host, inside the model's forward, one Python thread per GPU):
    collect_and_distribute_fpn_rpn_proposals.py:61-76   collect over the minibatch, then
    datasets/json_dataset.py:413-491                    add_proposals: IoU of [gt ; proposals] vs gt, class assignment
    roi_data/fast_rcnn.py:129-192                       _sample_rois: fg / bg sampling, labels
    roi_data/fast_rcnn.py:195-246                       bbox targets (bbox_transform_inv, weights 10,10,5,5) and expansion
    roi_data/mask_rcnn.py:34-107                        mask RoIs + 28x28 mask targets of the fg RoIs
    roi_data/fast_rcnn.py:249-278, utils/fpn.py:11-58   FPN level of every RoI

Here everything is a fixed sequence of tensor operations on the device the RPN outputs already live on -- no
device-to-host copy, no data-dependent shape:

  * every image owns `BATCH_SIZE_PER_IM` box rows and `round(FG_FRACTION * BATCH_SIZE_PER_IM)` mask rows; rows the
    reference would not have produced (an image with too few candidates, fewer fg RoIs than mask slots) are PADDING:
    label -1, zero box weights, all -1 mask targets.  The losses ignore them by their own formulas (heads.py), so the
    loss values and gradients are those of the reference's variable-length blobs.  Static shapes keep MIOpen on one
    solver per layer and make the step capturable.
  * `np.random.choice(inds, size, replace=False)` (fast_rcnn.py:146,160) is "the first `size` of a random permutation";
    the permutation is an INPUT here (`priority`: one float per candidate, lower = drawn first), which makes the
    sampling testable against the reference's function executed from its own source with the same permutation.
  * the IoU matrix is `mi_bbox_overlaps` (the cython_bbox replacement, nms.py); other callables can be injected
    (`iou_fn`) -- the CPU tests inject the oracle's.

Candidates are ordered as the reference's roidb entry: an image's gt boxes first, then its proposals in collect order
(json_dataset.py:461-465).  Crowd regions: TRAIN.CROWD_FILTER_THRESH is not applied on this path by the reference
either (crowd_thresh=0, collect_and...py:69-71); gt boxes passed in must be the non-crowd ones.
"""
import torch

from ..fpn_proposals import map_rois_to_fpn_levels


def bbox_transform_inv(boxes, gt_boxes, weights):
    """utils/boxes.py:199-233 in fp32: regression targets (dx, dy, dw, dh) of `boxes` towards `gt_boxes`."""
    ex_w = boxes[:, 2] - boxes[:, 0] + 1.0
    ex_h = boxes[:, 3] - boxes[:, 1] + 1.0
    ex_cx = boxes[:, 0] + 0.5 * ex_w
    ex_cy = boxes[:, 1] + 0.5 * ex_h
    gt_w = gt_boxes[:, 2] - gt_boxes[:, 0] + 1.0
    gt_h = gt_boxes[:, 3] - gt_boxes[:, 1] + 1.0
    gt_cx = gt_boxes[:, 0] + 0.5 * gt_w
    gt_cy = gt_boxes[:, 1] + 0.5 * gt_h
    wx, wy, ww, wh = weights
    return torch.stack([wx * (gt_cx - ex_cx) / ex_w, wy * (gt_cy - ex_cy) / ex_h,
                        ww * torch.log(gt_w / ex_w), wh * torch.log(gt_h / ex_h)], dim=1)


def rasterize_boxes(mask_boxes, rois, m):
    """The role of `segm_utils.polys_to_mask_wrt_box` (utils/segms.py:93-119) for axis-aligned rectangular ground-truth
    masks: the part of rectangle `mask_boxes[i]` inside `rois[i]`, as an m x m binary image (int32).  The rectangle is
    moved to the RoI's frame and scaled by m / max(roi side, 1) exactly as :104-112 does with the polygon's vertices; a
    pixel is set when its centre lies inside the scaled rectangle.  (The reference rasterises through pycocotools'
    `frPyObjects`, which is not available here: boundary pixels may differ from it -- data-layer detail, parity unpinned.)"""
    w = torch.clamp_min(rois[:, 2] - rois[:, 0], 1)
    h = torch.clamp_min(rois[:, 3] - rois[:, 1], 1)
    x1 = (mask_boxes[:, 0] - rois[:, 0]) * m / w
    x2 = (mask_boxes[:, 2] - rois[:, 0]) * m / w
    y1 = (mask_boxes[:, 1] - rois[:, 1]) * m / h
    y2 = (mask_boxes[:, 3] - rois[:, 1]) * m / h
    centres = torch.arange(m, device=rois.device, dtype=rois.dtype) + 0.5
    in_x = (centres[None, :] >= x1[:, None]) & (centres[None, :] <= x2[:, None])      # [n, m]
    in_y = (centres[None, :] >= y1[:, None]) & (centres[None, :] <= y2[:, None])
    return (in_y[:, :, None] & in_x[:, None, :]).to(torch.int32).reshape(-1, m * m)


def expand_bbox_targets(labels, targets, num_classes):
    """fast_rcnn.py:213-246: [n,4] targets -> ([n,4K] targets, [n,4K] inside weights); only the row's class gets the
    four values / ones, and only for labels > 0."""
    n = labels.numel()
    cols = torch.arange(4 * num_classes, device=labels.device).view(1, -1)
    cls = labels.clamp_min(0).view(-1, 1)
    hit = (cols // 4 == cls) & (labels.view(-1, 1) > 0)
    inside = hit.to(targets.dtype)
    full = targets.repeat(1, num_classes) * inside
    assert full.shape == (n, 4 * num_classes)
    return full, inside


def expand_to_class_specific_mask_targets(masks, mask_class_labels, num_classes):
    """mask_rcnn.py:110-129: [n, M*M] -> [n, K*M*M] int32, -1 everywhere except the block of the row's class (class 0
    rows stay all -1).  Only the parity tests need this layout; the loss works on the compact form."""
    n, mm = masks.shape
    out = torch.full((n, num_classes, mm), -1, dtype=torch.int32, device=masks.device)
    rows = torch.nonzero(mask_class_labels > 0, as_tuple=False).flatten()
    out[rows, mask_class_labels[rows].long()] = masks[rows]
    return out.reshape(n, num_classes * mm)


def keypoints_to_heatmap_labels(keypoints, rois, heatmap_size):
    """utils/keypoints.py:160-211: keypoints [n,3,K] (x, y, visibility) and their RoIs [n,4] (same coordinates) ->
    (heat-map cell index [n,K] float32, weight [n,K] float32).  A keypoint exactly on the RoI's right / bottom edge is
    moved into the last cell (:184-197); invisible keypoints and keypoints outside the map weigh 0."""
    x, y, vis = keypoints[:, 0, :].to(torch.float32), keypoints[:, 1, :].to(torch.float32), keypoints[:, 2, :] > 0
    off_x, off_y = rois[:, 0:1], rois[:, 1:2]
    scale_x = heatmap_size / (rois[:, 2:3] - rois[:, 0:1])
    scale_y = heatmap_size / (rois[:, 3:4] - rois[:, 1:2])
    on_right, on_bottom = x == rois[:, 2:3], y == rois[:, 3:4]
    hx = torch.floor((x - off_x) * scale_x)
    hy = torch.floor((y - off_y) * scale_y)
    hx = torch.where(on_right, torch.full_like(hx, heatmap_size - 1), hx)
    hy = torch.where(on_bottom, torch.full_like(hy, heatmap_size - 1), hy)
    valid = (hx >= 0) & (hy >= 0) & (hx < heatmap_size) & (hy < heatmap_size) & vis
    validf = valid.to(torch.float32)
    lin = hy * heatmap_size + hx
    return torch.where(valid, lin, torch.zeros_like(lin)), validf


def label_proposals(cfg, rois, gt_boxes, gt_classes, gt_image, im_scales, priority, num_images, iou_fn,
                    roi_valid=None, gt_mask_boxes=None, gt_keypoints=None, gt_polygons=None, rasterize_fn=None):
    """Label and sample the collected proposals of a minibatch.

    rois       [R,5] float32  (image index, x1, y1, x2, y2) in network-input coordinates, collect order
    gt_boxes   [G,4] float32  ground-truth boxes in ORIGINAL image coordinates (roidb 'boxes'), gt_classes [G] (> 0),
               gt_image [G] image index of each, ascending
    im_scales  [N] float32    im_info[:, 2]
    priority   [G+R] float32  sampling priority of every candidate, gt first (unique values; lower = drawn first)
    roi_valid  [R] bool       rows of `rois` that are real (static-shape collect); None = all
    gt_mask_boxes [G,4]       the rectangles that are the instances' masks (default: the gt boxes themselves)
    gt_polygons               segms.PackedPolygons of the G instances (roidb 'segms', original image coordinates): the mask
                              targets are rasterised from them by pycocotools' rule through `rasterize_fn(packed, roi_inst,
                              boxes, M)` (segms.polys_to_masks_wrt_boxes: one HIP launch); takes precedence over gt_mask_boxes
    gt_keypoints [G,3,K]      (x, y, visibility) of every instance's keypoints in original image coordinates
                              (MODEL.KEYPOINTS_ON)

    Returns a dict of device tensors, rows [i * B, (i + 1) * B) belonging to image i (B = BATCH_SIZE_PER_IM, F mask rows
    per image):
      rois [N*B,5], labels_int32 [N*B] (-1 = padding), bbox_targets / bbox_inside_weights / bbox_outside_weights [N*B,4K],
      num_rois [N] (real rows per image), num_fg [N],
      mask_rois [N*F,5], mask_class [N*F] (0 = padding), masks_int32 [N*F, M*M] (-1 rows = padding), roi_has_mask_int32,
      rois_levels / mask_rois_levels (int32 FPN level of every row, utils/fpn.py:11-28);
      with keypoints: keypoint_rois [N*F,5], keypoint_locations_int32 / keypoint_weights [N*F*K], keypoint_rois_levels,
      keypoint_loss_normalizer (roi_data/keypoint_rcnn.py:33-106)
    """
    t = cfg.TRAIN
    dev, f32 = rois.device, torch.float32
    n_img, num_classes = int(num_images), cfg.MODEL.NUM_CLASSES
    per_image = int(t.BATCH_SIZE_PER_IM)
    fg_per_image = int(round(t.FG_FRACTION * per_image))                     # np.round(...) fast_rcnn.py:134
    g, r = gt_boxes.size(0), rois.size(0)
    roi_img = rois[:, 0].long()
    inv_scale = (1.0 / im_scales.to(f32))                                     # json_dataset.py:420
    cand_boxes = torch.cat([gt_boxes, rois[:, 1:5] * inv_scale[roi_img].unsqueeze(1)], dim=0)
    cand_img = torch.cat([gt_image.long(), roi_img])
    cand_valid = torch.ones(g + r, dtype=torch.bool, device=dev)
    if roi_valid is not None:
        cand_valid[g:] = roi_valid
    # --- json_dataset.py:437-460: overlap of every candidate with the gt boxes of its own image
    if g > 0:
        iou = iou_fn(cand_boxes[g:].contiguous(), gt_boxes.contiguous())     # [R,G]
        iou = torch.where(roi_img.view(-1, 1) == gt_image.long().view(1, -1), iou, torch.zeros_like(iou))
        prop_max, prop_arg = iou.max(dim=1)
        # a gt box overlaps itself with 1.0 and belongs to its own class (roidb initialisation, json_dataset.py:228-231)
        max_ov = torch.cat([torch.ones(g, device=dev, dtype=f32), prop_max])
        assign = torch.cat([torch.arange(g, device=dev), prop_arg])
        has_gt = max_ov > 0
        max_cls = torch.where(has_gt, gt_classes.long()[assign], torch.zeros_like(assign))
    else:
        max_ov = torch.zeros(r, device=dev, dtype=f32)
        assign = torch.zeros(r, device=dev, dtype=torch.long)
        max_cls = torch.zeros(r, device=dev, dtype=torch.long)
    # --- fast_rcnn.py:137-160: fg / bg candidate sets, then "the first k of a permutation" per image and set
    fg = (max_ov >= t.FG_THRESH) & cand_valid
    bg = (max_ov < t.BG_THRESH_HI) & (max_ov >= t.BG_THRESH_LO) & cand_valid
    group = torch.where(fg, 0, torch.where(bg, 1, 2))
    seg = cand_img * 3 + group
    by_prio = torch.argsort(priority, stable=True)
    order = by_prio[torch.argsort(seg[by_prio], stable=True)]                 # (image, group, priority)-sorted candidates
    # (torch.bincount sizes its output on the host -- a synchronisation; a comparison against the 3N segment ids is not)
    counts = (seg.view(-1, 1) == torch.arange(3 * n_img, device=dev).view(1, -1)).sum(dim=0).view(n_img, 3)
    starts = (torch.cumsum(counts.view(-1), 0) - counts.view(-1)).view(n_img, 3)
    n_fg = torch.clamp_max(counts[:, 0], fg_per_image)
    n_bg = torch.minimum(per_image - n_fg, counts[:, 1])
    slot = torch.arange(per_image, device=dev).view(1, -1)                    # [1,B]
    is_fg = slot < n_fg.view(-1, 1)
    is_bg = (~is_fg) & (slot < (n_fg + n_bg).view(-1, 1))
    pos = torch.where(is_fg, starts[:, 0:1] + slot, starts[:, 1:2] + slot - n_fg.view(-1, 1))
    real = is_fg | is_bg
    src = order[torch.where(real, pos, torch.zeros_like(pos)).clamp_(0, max(g + r - 1, 0))]   # [N,B] candidate index
    labels = torch.where(is_fg, max_cls[src], torch.zeros_like(src))
    labels = torch.where(real, labels, torch.full_like(labels, -1)).view(-1)
    boxes = cand_boxes[src.view(-1)] * real.view(-1, 1).to(f32)               # padding rows: a 0,0,0,0 box
    # --- fast_rcnn.py:169-177,195-246: regression targets towards the assigned gt box
    if g > 0:
        targets = bbox_transform_inv(boxes, gt_boxes[assign[src.view(-1)]], cfg.MODEL.BBOX_REG_WEIGHTS)
    else:
        targets = torch.zeros((n_img * per_image, 4), device=dev, dtype=f32)
    reg_labels = labels.clamp_max(1) if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else labels
    bbox_targets, inside = expand_bbox_targets(reg_labels, torch.where(reg_labels.view(-1, 1) > 0, targets,
                                                                      torch.zeros_like(targets)),
                                               2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else num_classes)
    outside = (inside > 0).to(f32)
    # padding rows carry image index -1: the RoI operators pool zeros for a RoI outside the batch and send it no gradient
    # (include/mi_detectron_ops.h) -- and, unlike a dummy box, they do not pile up on one tile of the backward
    img_col = torch.where(real, torch.arange(n_img, device=dev).view(-1, 1), -1).to(f32).reshape(-1, 1)
    scale_col = im_scales.to(f32).view(-1, 1).expand(n_img, per_image).reshape(-1, 1)
    out = {
        "rois": torch.cat([img_col, boxes * scale_col], dim=1),              # fast_rcnn.py:183-185
        "labels_int32": labels.to(torch.int32), "bbox_targets": bbox_targets, "bbox_inside_weights": inside,
        "bbox_outside_weights": outside, "num_rois": (n_fg + n_bg), "num_fg": n_fg,
    }
    out["rois_levels"] = map_rois_to_fpn_levels(out["rois"][:, 1:5], cfg.FPN.ROI_MIN_LEVEL, cfg.FPN.ROI_MAX_LEVEL)
    if cfg.MODEL.MASK_ON:
        m = cfg.MRCNN.RESOLUTION
        # --- mask_rcnn.py:34-107: the fg rows are the first n_fg rows of every image's block
        fslot = torch.arange(fg_per_image, device=dev).view(1, -1)
        has = fslot < n_fg.view(-1, 1)                                        # [N,F]
        fsrc = src[:, :fg_per_image]
        fboxes = cand_boxes[fsrc.reshape(-1)] * has.view(-1, 1).to(f32)
        fcls = torch.where(has, max_cls[fsrc], torch.zeros_like(fsrc)).view(-1)
        if g > 0:
            if gt_polygons is not None:
                from .. import segms
                mboxes = segms.polys_to_boxes(gt_polygons)                    # mask_rcnn.py:44
            else:
                mboxes = gt_boxes if gt_mask_boxes is None else gt_mask_boxes
            fimg = torch.arange(n_img, device=dev).view(-1, 1).expand(n_img, fg_per_image).reshape(-1)
            ov = iou_fn(fboxes.contiguous(), mboxes.contiguous())            # [N*F,G] vs the boxes enclosing the polygons
            ov = torch.where(fimg.view(-1, 1) == gt_image.long().view(1, -1), ov, torch.full_like(ov, -1.0))
            poly = ov.argmax(dim=1)                                           # mask_rcnn.py:62
            if gt_polygons is not None:
                # mask_rcnn.py:66-76, every row in one launch; padding rows name no instance
                inst = torch.where(has.view(-1), poly, torch.full_like(poly, -1))
                masks = (rasterize_fn or segms.polys_to_masks_wrt_boxes)(gt_polygons, inst, fboxes.contiguous(), m)
            else:
                masks = rasterize_boxes(mboxes[poly], fboxes, m)
        else:
            masks = torch.zeros((n_img * fg_per_image, m * m), dtype=torch.int32, device=dev)
        masks = torch.where(has.view(-1, 1), masks, torch.full_like(masks, -1))
        fimg_col = torch.where(has, torch.arange(n_img, device=dev).view(-1, 1), -1).to(f32).reshape(-1, 1)
        fscale = im_scales.to(f32).view(-1, 1).expand(n_img, fg_per_image).reshape(-1, 1)
        out["mask_rois"] = torch.cat([fimg_col, fboxes * fscale], dim=1)     # mask_rcnn.py:99-101
        out["mask_class"] = fcls.to(torch.int32)
        out["masks_int32"] = masks
        out["roi_has_mask_int32"] = (labels > 0).to(torch.int32)
        out["mask_rois_levels"] = map_rois_to_fpn_levels(out["mask_rois"][:, 1:5], cfg.FPN.ROI_MIN_LEVEL,
                                                         cfg.FPN.ROI_MAX_LEVEL)
    if cfg.MODEL.KEYPOINTS_ON:
        # --- roi_data/keypoint_rcnn.py:33-89: candidates with IoU >= FG_THRESH whose assigned instance has a visible
        # keypoint inside the candidate box; at most fg_per_image of them per image -- all of them in candidate order
        # when they fit, else "the first fg_per_image of the permutation" (np.random.choice, :52-54)
        hm = cfg.KRCNN.HEATMAP_SIZE
        kps = gt_keypoints.to(f32)                                            # [G,3,K]
        nk = kps.size(2)
        ckp = kps[assign]                                                     # candidate -> its instance's keypoints
        inside = (ckp[:, 0, :] >= cand_boxes[:, 0:1]) & (ckp[:, 0, :] <= cand_boxes[:, 2:3]) \
            & (ckp[:, 1, :] >= cand_boxes[:, 1:2]) & (ckp[:, 1, :] <= cand_boxes[:, 3:4])
        is_visible = ((ckp[:, 2, :] > 0) & inside).any(dim=1)
        kfg = (max_ov >= t.FG_THRESH) & is_visible & cand_valid
        kseg = torch.where(kfg, cand_img, torch.full_like(cand_img, n_img))
        kcounts = (kseg.view(-1, 1) == torch.arange(n_img, device=dev).view(1, -1)).sum(dim=0)      # [N]
        too_many = (kcounts > fg_per_image)[cand_img.clamp_max(n_img - 1)]
        index_key = torch.arange(g + r, device=dev, dtype=f32)
        prio_rank = torch.empty(g + r, device=dev, dtype=f32)
        prio_rank[by_prio] = index_key
        key = torch.where(too_many, prio_rank, index_key)
        k1 = torch.argsort(key, stable=True)
        korder = k1[torch.argsort(kseg[k1], stable=True)]                     # (image, key)-sorted; non-candidates last
        kstarts = torch.cumsum(kcounts, 0) - kcounts
        n_kp = torch.clamp_max(kcounts, fg_per_image)
        kslot = torch.arange(fg_per_image, device=dev).view(1, -1)
        khas = kslot < n_kp.view(-1, 1)
        ksrc = korder[torch.where(khas, kstarts.view(-1, 1) + kslot, torch.zeros_like(kslot)).clamp_(0, max(g + r - 1, 0))]
        kboxes = cand_boxes[ksrc.reshape(-1)] * khas.view(-1, 1).to(f32)
        skp = ckp[ksrc.reshape(-1)]                                           # [N*F,3,K]
        heats, kweights = keypoints_to_heatmap_labels(skp, kboxes, hm)
        kweights = kweights * khas.view(-1, 1).to(f32)
        heats = heats * khas.view(-1, 1).to(f32)
        kimg_col = torch.where(khas, torch.arange(n_img, device=dev).view(-1, 1), -1).to(f32).reshape(-1, 1)
        kscale = im_scales.to(f32).view(-1, 1).expand(n_img, fg_per_image).reshape(-1, 1)
        out["keypoint_rois"] = torch.cat([kimg_col, kboxes * kscale], dim=1)
        out["keypoint_locations_int32"] = heats.reshape(-1).to(torch.int32)
        out["keypoint_weights"] = kweights.reshape(-1)
        out["num_keypoint_rois"] = n_kp
        out["keypoint_rois_levels"] = map_rois_to_fpn_levels(out["keypoint_rois"][:, 1:5], cfg.FPN.ROI_MIN_LEVEL,
                                                             cfg.FPN.ROI_MAX_LEVEL)
        # keypoint_rcnn.py:92-106 (used when KRCNN.NORMALIZE_BY_VISIBLE_KEYPOINTS is off)
        out["keypoint_loss_normalizer"] = kweights.sum() / (t.IMS_PER_BATCH * t.BATCH_SIZE_PER_IM * t.FG_FRACTION * nk)
    return out
