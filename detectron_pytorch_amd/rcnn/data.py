"""Host-side data layer for the synthetic benchmark minibatch: what the reference's
DataLoader workers hand to the model for one rank -- image blob, im_info, a minimal roidb and the "wide" RPN target
blobs -- built with numpy once, then kept resident in HBM.

RPN targets restate the reference's data layer:
    roi_data/data_utils.py:50-104   get_field_of_anchors   all anchors of a level over a FIXED field of
                                                           ceil(fpn_max_size / stride)^2 cells ("wide" blobs)
    roi_data/rpn.py:40-113          add_rpn_blobs          per image, all levels' anchors concatenated
    roi_data/rpn.py:115-262         _get_rpn_blobs         inside-image filter, IoU labelling (>= 0.7 positive, per-gt
                                                           best anchors positive, < 0.3 negative), sub-sampling to 256
                                                           per image, regression targets, inside / outside weights
The IoU matrix comes from a callable (`bbox_overlaps`), the sampling from a `numpy.random.RandomState` used in the
reference's call order (npr.choice for surplus positives, npr.randint for negatives), so that the function can be
checked against the reference's own source under the same seed (tests/test_model_cpu.py).
"""
import numpy as np

from ..generate_proposals import generate_anchors


def bbox_overlaps_np(boxes, query_boxes):
    """IoU matrix with the +1 convention (utils/cython_bbox.pyx:32-73 in numpy; float32 in, float32 out)."""
    boxes = boxes.astype(np.float32, copy=False)
    q = query_boxes.astype(np.float32, copy=False)
    q_area = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    b_area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    iw = np.minimum(boxes[:, None, 2], q[None, :, 2]) - np.maximum(boxes[:, None, 0], q[None, :, 0]) + 1
    ih = np.minimum(boxes[:, None, 3], q[None, :, 3]) - np.maximum(boxes[:, None, 1], q[None, :, 1]) + 1
    inter = np.where((iw > 0) & (ih > 0), iw * ih, 0).astype(np.float32)
    ua = (b_area[:, None] + q_area[None, :] - inter).astype(np.float32)
    return np.where(inter > 0, inter / ua, 0).astype(np.float32)


def bbox_transform_inv_np(boxes, gt_boxes, weights=(1.0, 1.0, 1.0, 1.0)):
    """utils/boxes.py:199-233."""
    ex_w = boxes[:, 2] - boxes[:, 0] + 1.0
    ex_h = boxes[:, 3] - boxes[:, 1] + 1.0
    ex_cx = boxes[:, 0] + 0.5 * ex_w
    ex_cy = boxes[:, 1] + 0.5 * ex_h
    gt_w = gt_boxes[:, 2] - gt_boxes[:, 0] + 1.0
    gt_h = gt_boxes[:, 3] - gt_boxes[:, 1] + 1.0
    gt_cx = gt_boxes[:, 0] + 0.5 * gt_w
    gt_cy = gt_boxes[:, 1] + 0.5 * gt_h
    wx, wy, ww, wh = weights
    return np.vstack((wx * (gt_cx - ex_cx) / ex_w, wy * (gt_cy - ex_cy) / ex_h, ww * np.log(gt_w / ex_w),
                      wh * np.log(gt_h / ex_h))).transpose()


class FieldOfAnchors(object):
    """data_utils.py:35-104: every anchor of one pyramid level over the fixed field (float32 [K*A,4], cell-major)."""

    def __init__(self, stride, anchor_sizes, aspect_ratios, max_size, coarsest_stride):
        cell = generate_anchors(stride=stride, sizes=anchor_sizes, aspect_ratios=aspect_ratios)
        self.num_cell_anchors = cell.shape[0]
        self.stride = stride
        fpn_max_size = coarsest_stride * np.ceil(max_size / float(coarsest_stride))
        self.field_size = int(np.ceil(fpn_max_size / float(stride)))
        shifts = np.arange(0, self.field_size) * stride
        sx, sy = np.meshgrid(shifts, shifts)
        shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
        a, k = self.num_cell_anchors, shifts.shape[0]
        field = cell.reshape((1, a, 4)) + shifts.reshape((1, k, 4)).transpose((1, 0, 2))
        self.field_of_anchors = field.reshape((k * a, 4)).astype(np.float32)


def fields_of_anchors(cfg):
    """rpn.py:42-55 for FPN multilevel RPN: one field per level k_min..k_max."""
    return [FieldOfAnchors(2. ** lvl, (cfg.FPN.RPN_ANCHOR_START_SIZE * 2. ** (lvl - cfg.FPN.RPN_MIN_LEVEL),),
                           cfg.FPN.RPN_ASPECT_RATIOS, cfg.TRAIN.MAX_SIZE, cfg.FPN.COARSEST_STRIDE)
            for lvl in range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1)]


def get_rpn_blobs(cfg, im_height, im_width, foas, all_anchors, gt_boxes, rng, bbox_overlaps=bbox_overlaps_np):
    """rpn.py:115-262 for one image.  Returns one dict per level with the four wide blobs ([1,A,F,F] int32 labels,
    [1,4A,F,F] float32 targets / inside weights / outside weights)."""
    t = cfg.TRAIN
    total = all_anchors.shape[0]
    straddle = t.RPN_STRADDLE_THRESH
    if straddle >= 0:
        inside = np.where((all_anchors[:, 0] >= -straddle) & (all_anchors[:, 1] >= -straddle)
                          & (all_anchors[:, 2] < im_width + straddle) & (all_anchors[:, 3] < im_height + straddle))[0]
        anchors = all_anchors[inside, :]
    else:
        inside = np.arange(total)
        anchors = all_anchors
    num_inside = len(inside)
    labels = np.full((num_inside,), -1, dtype=np.int32)
    a2g_arg = np.zeros((num_inside,), dtype=np.int64)
    a2g_max = np.zeros((num_inside,), dtype=np.float32)   # (the reference leaves these undefined without gt boxes)
    if len(gt_boxes) > 0:
        ov = bbox_overlaps(anchors, gt_boxes)
        a2g_arg = ov.argmax(axis=1)
        a2g_max = ov[np.arange(num_inside), a2g_arg]
        g2a_arg = ov.argmax(axis=0)
        g2a_max = ov[g2a_arg, np.arange(ov.shape[1])]
        labels[np.where(ov == g2a_max)[0]] = 1                       # every anchor tied for a gt's best overlap
        labels[a2g_max >= t.RPN_POSITIVE_OVERLAP] = 1
    num_fg = int(t.RPN_FG_FRACTION * t.RPN_BATCH_SIZE_PER_IM)
    fg = np.where(labels == 1)[0]
    if len(fg) > num_fg:
        labels[rng.choice(fg, size=(len(fg) - num_fg), replace=False)] = -1
    fg = np.where(labels == 1)[0]
    num_bg = t.RPN_BATCH_SIZE_PER_IM - np.sum(labels == 1)
    bg = np.where(a2g_max < t.RPN_NEGATIVE_OVERLAP)[0]
    if len(bg) > num_bg:
        labels[bg[rng.randint(len(bg), size=num_bg)]] = 0
    targets = np.zeros((num_inside, 4), dtype=np.float32)
    targets[fg, :] = bbox_transform_inv_np(anchors[fg, :], gt_boxes[a2g_arg[fg], :]).astype(np.float32, copy=False)
    w_in = np.zeros((num_inside, 4), dtype=np.float32)
    w_in[labels == 1, :] = (1.0, 1.0, 1.0, 1.0)
    w_out = np.zeros((num_inside, 4), dtype=np.float32)
    num_examples = np.sum(labels >= 0)
    w_out[labels == 1, :] = 1.0 / num_examples
    w_out[labels == 0, :] = 1.0 / num_examples

    def unmap(data, fill):
        if total == num_inside:
            return data
        out = np.full((total,) + data.shape[1:], fill, dtype=data.dtype)
        out[inside] = data
        return out

    labels, targets, w_in, w_out = unmap(labels, -1), unmap(targets, 0), unmap(w_in, 0), unmap(w_out, 0)
    blobs, start = [], 0
    for foa in foas:
        f, a = foa.field_size, foa.num_cell_anchors
        end = start + f * f * a
        blobs.append(dict(
            rpn_labels_int32_wide=labels[start:end].reshape((1, f, f, a)).transpose(0, 3, 1, 2),
            rpn_bbox_targets_wide=targets[start:end, :].reshape((1, f, f, a * 4)).transpose(0, 3, 1, 2),
            rpn_bbox_inside_weights_wide=w_in[start:end, :].reshape((1, f, f, a * 4)).transpose(0, 3, 1, 2),
            rpn_bbox_outside_weights_wide=w_out[start:end, :].reshape((1, f, f, a * 4)).transpose(0, 3, 1, 2)))
        start = end
    return blobs


def add_rpn_blobs(cfg, roidb, im_scales, rng, bbox_overlaps=bbox_overlaps_np):
    """rpn.py:40-113 for FPN: the wide RPN blobs of a minibatch, concatenated over images, keyed
    '<blob>_fpn<lvl>', plus 'im_info' [N,3]."""
    foas = fields_of_anchors(cfg)
    all_anchors = np.concatenate([f.field_of_anchors for f in foas])
    out = {}
    infos = []
    for i, entry in enumerate(roidb):
        scale = im_scales[i]
        h, w = np.round(entry["height"] * scale), np.round(entry["width"] * scale)
        keep = np.where((entry["gt_classes"] > 0) & (entry["is_crowd"] == 0))[0]
        gt = entry["boxes"][keep, :] * scale
        infos.append(np.array([[h, w, scale]], dtype=np.float32))
        per_level = get_rpn_blobs(cfg, h, w, foas, all_anchors, gt, rng, bbox_overlaps)
        for k, lvl in enumerate(range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1)):
            for name, v in per_level[k].items():
                out.setdefault("%s_fpn%d" % (name, lvl), []).append(v)
    out = {k: np.ascontiguousarray(np.concatenate(v)) for k, v in out.items()}
    out["im_info"] = np.concatenate(infos)
    return out


def synthetic_roidb(num_images, height=800, width=1333, boxes_per_image=8, num_classes=81, seed=0, num_keypoints=0):
    """`boxes_per_image` ground-truth boxes per image with sides U(32, 400), classes
    U{1..num_classes-1}, no crowd regions; each instance's mask is its own rectangle."""
    rng = np.random.RandomState(seed)
    roidb = []
    for _ in range(num_images):
        bw = rng.uniform(32, 400, boxes_per_image)
        bh = rng.uniform(32, 400, boxes_per_image)
        x1 = rng.uniform(0, width - 1 - bw)
        y1 = rng.uniform(0, height - 1 - bh)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], axis=1).astype(np.float32)
        entry = dict(height=height, width=width, boxes=boxes,
                     gt_classes=rng.randint(1, num_classes, boxes_per_image).astype(np.int32),
                     is_crowd=np.zeros(boxes_per_image, dtype=bool))
        if num_keypoints:
            # (x, y, visibility) per keypoint, uniformly inside the instance's box, all labelled visible (COCO v = 2)
            kx = boxes[:, 0:1] + rng.uniform(0, 1, (boxes_per_image, num_keypoints)) * bw[:, None]
            ky = boxes[:, 1:2] + rng.uniform(0, 1, (boxes_per_image, num_keypoints)) * bh[:, None]
            entry["gt_keypoints"] = np.stack([kx, ky, np.full_like(kx, 2.0)], axis=1).astype(np.int32)
        roidb.append(entry)
    return roidb


def synthetic_minibatch(cfg, num_images, seed=0, blob_height=800, blob_width=1344, image_width=1333):
    """One rank's training minibatch as numpy: 'data' [N,3,H,W] (mean-subtracted BGR scale, randn * 50), 'im_info',
    the wide RPN blobs and the roidb tensors the labelling needs.  Scale 1.0: a 1333x800 image padded to the /32 blob
    (utils/blob.py:97-100)."""
    rng = np.random.RandomState(seed + 1000)
    roidb = synthetic_roidb(num_images, blob_height, image_width, num_classes=cfg.MODEL.NUM_CLASSES, seed=seed,
                            num_keypoints=cfg.KRCNN.NUM_KEYPOINTS if cfg.MODEL.KEYPOINTS_ON else 0)
    batch = add_rpn_blobs(cfg, roidb, [1.0] * num_images, rng)
    batch["data"] = (rng.randn(num_images, 3, blob_height, blob_width) * 50).astype(np.float32)
    batch["gt_boxes"] = np.concatenate([e["boxes"] for e in roidb])
    batch["gt_classes"] = np.concatenate([e["gt_classes"] for e in roidb]).astype(np.int64)
    batch["gt_image"] = np.concatenate([np.full(len(e["boxes"]), i, dtype=np.int64) for i, e in enumerate(roidb)])
    if cfg.MODEL.KEYPOINTS_ON:
        batch["gt_keypoints"] = np.concatenate([e["gt_keypoints"] for e in roidb])
    return batch


def to_device(batch, device, channels_last=False):
    """Upload a synthetic minibatch; returns (data, im_info, roidb dict, rpn_targets dict) of device tensors."""
    import torch

    data = torch.from_numpy(batch["data"]).to(device)
    if channels_last:
        data = data.contiguous(memory_format=torch.channels_last)
    roidb = {k: torch.from_numpy(batch[k]).to(device) for k in ("gt_boxes", "gt_classes", "gt_image", "gt_keypoints")
             if k in batch}
    rpn = {k: torch.from_numpy(v).to(device) for k, v in batch.items() if k.startswith("rpn_")}
    return data, torch.from_numpy(batch["im_info"]).to(device), roidb, rpn
