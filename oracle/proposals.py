"""TEST INFRASTRUCTURE -- numpy restatement of the reference's RPN proposal generation.

`generate_anchors` follows lib/modeling/generate_anchors.py:54-123; `generate_proposals` follows
lib/modeling/generate_proposals.py:19-182 (forward + proposals_for_one_image + _filter_boxes) with the helpers it
calls: utils/boxes.py:156-196 (bbox_transform), :138-153 (clip_tiled_boxes), :320-324 (nms -> cython_nms, here
oracle.nms_cython).  cfg values are arguments.

Arithmetic types are those numpy >= 2 gives the reference's expressions (the environment it runs in here, and the one
tests/golden/proposals.npz was produced under): everything is float32 except the width / height branch of the decode,
which `np.minimum(dw, cfg.BBOX_XFORM_CLIP)` promotes to float64 because the clip constant is an np.float64 scalar
(core/config.py:936) -- written out below with explicit casts so that the restatement does not depend on the numpy
version that executes it.  Never imported by the product package.
"""
import numpy as np

import oracle

BBOX_XFORM_CLIP = np.float64(np.log(1000.0 / 16.0))  # core/config.py:936


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """generate_anchors.py:54-123 (float64 throughout)."""
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)

    def whctrs(a):
        w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, x_ctr, y_ctr):
        ws, hs = ws[:, None], hs[:, None]
        return np.hstack((x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)))

    base = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    w, h, x_ctr, y_ctr = whctrs(base)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = mk(ws, hs, x_ctr, y_ctr)
    out = []
    for a in ratio_anchors:
        w, h, x_ctr, y_ctr = whctrs(a)
        out.append(mk(w * scales, h * scales, x_ctr, y_ctr))
    return np.vstack(out)


def bbox_transform(boxes, deltas):
    """utils/boxes.py:156-196 with weights (1, 1, 1, 1); boxes [K,4] any float, deltas [K,4] float32."""
    f32, f64 = np.float32, np.float64
    boxes = boxes.astype(f32)                                   # :164
    widths = boxes[:, 2] - boxes[:, 0] + f32(1.0)
    heights = boxes[:, 3] - boxes[:, 1] + f32(1.0)
    ctr_x = boxes[:, 0] + f32(0.5) * widths
    ctr_y = boxes[:, 1] + f32(0.5) * heights
    dx, dy = deltas[:, 0] / f32(1.0), deltas[:, 1] / f32(1.0)
    dw = np.minimum(deltas[:, 2].astype(f64), BBOX_XFORM_CLIP)  # :178 -- float64 from here on (np.float64 scalar)
    dh = np.minimum(deltas[:, 3].astype(f64), BBOX_XFORM_CLIP)
    pred_ctr_x = dx * widths + ctr_x                            # float32
    pred_ctr_y = dy * heights + ctr_y
    pred_w = np.exp(dw) * widths.astype(f64)                    # float64
    pred_h = np.exp(dh) * heights.astype(f64)
    pred = np.zeros(deltas.shape, dtype=f32)
    pred[:, 0] = pred_ctr_x.astype(f64) - 0.5 * pred_w          # rounded to float32 on assignment
    pred[:, 1] = pred_ctr_y.astype(f64) - 0.5 * pred_h
    pred[:, 2] = pred_ctr_x.astype(f64) + 0.5 * pred_w - 1
    pred[:, 3] = pred_ctr_y.astype(f64) + 0.5 * pred_h - 1
    return pred


def generate_proposals(scores, bbox_deltas, im_info, anchors, spatial_scale, pre_nms_topN=12000, post_nms_topN=2000,
                       nms_thresh=0.7, min_size=0):
    """scores [N,A,H,W], bbox_deltas [N,4A,H,W], im_info [N,3] float32; anchors [A,4] float64.  Returns
    (rois [R,5] float32, roi_probs [R,1] float32) like GenerateProposalsOp.forward (:102-104)."""
    f32 = np.float32
    feat_stride = 1.0 / spatial_scale
    height, width = scores.shape[-2:]
    shift_x, shift_y = np.meshgrid(np.arange(0, width) * feat_stride, np.arange(0, height) * feat_stride)
    shifts = np.vstack((shift_x.ravel(), shift_y.ravel(), shift_x.ravel(), shift_y.ravel())).transpose()
    a, k = anchors.shape[0], shifts.shape[0]
    all_anchors = (anchors[np.newaxis, :, :] + shifts[:, np.newaxis, :]).reshape((k * a, 4))   # float64, (h, w, a) order
    rois = np.empty((0, 5), dtype=f32)
    roi_probs = np.empty((0, 1), dtype=f32)
    for im_i in range(scores.shape[0]):
        info = im_info[im_i].astype(f32)
        deltas = bbox_deltas[im_i].transpose((1, 2, 0)).reshape((-1, 4))
        sc = scores[im_i].transpose((1, 2, 0)).reshape((-1, 1))
        if pre_nms_topN <= 0 or pre_nms_topN >= len(sc):                                        # :131-139
            order = np.argsort(-sc.squeeze(), kind="stable")
        else:
            inds = np.argpartition(-sc.squeeze(), pre_nms_topN)[:pre_nms_topN]
            order = inds[np.argsort(-sc[inds].squeeze(), kind="stable")]
        deltas, anc, sc = deltas[order, :], all_anchors[order, :], sc[order]
        proposals = bbox_transform(anc, deltas)
        proposals[:, 0::2] = np.maximum(np.minimum(proposals[:, 0::2], info[1] - f32(1)), f32(0))   # clip, :146-152
        proposals[:, 1::2] = np.maximum(np.minimum(proposals[:, 1::2], info[0] - f32(1)), f32(0))
        ms = f32(min_size) * info[2]                                                             # _filter_boxes, :170-182
        ws = proposals[:, 2] - proposals[:, 0] + f32(1)
        hs = proposals[:, 3] - proposals[:, 1] + f32(1)
        x_ctr, y_ctr = proposals[:, 0] + ws / f32(2.0), proposals[:, 1] + hs / f32(2.0)
        keep = np.where((ws >= ms) & (hs >= ms) & (x_ctr < info[1]) & (y_ctr < info[0]))[0]
        proposals, sc = proposals[keep, :], sc[keep]
        if nms_thresh > 0:                                                                       # :155-161
            dets = np.hstack((proposals, sc)).astype(f32)
            keep = oracle.nms_cython(dets, nms_thresh) if dets.shape[0] else np.zeros((0,), np.int64)
            if post_nms_topN > 0:
                keep = keep[:post_nms_topN]
            proposals, sc = proposals[keep, :], sc[keep]
        batch_inds = im_i * np.ones((proposals.shape[0], 1), dtype=f32)
        rois = np.append(rois, np.hstack((batch_inds, proposals)), axis=0)
        roi_probs = np.append(roi_probs, sc, axis=0)
    return rois, roi_probs
