"""Per-class detection post-processing on the device: the caller of NMS at test time (call
site core/test.py:764; section 8f row 3).

Mirrors `box_results_with_nms_and_limit` (lib/core/test.py:732-790): threshold the class scores, NMS (or Soft-NMS) per
class, then keep the `detections_per_im` best over all classes.  The reference loops over the 80 classes on the host,
one cython_nms call each; here the classes become the segments of ONE class-major detection array, `mi_nms_batched`
(hard) or `mi_soft_nms_segmented` (soft) runs them side by side, and only two small count vectors cross to the host
(the reference's return type -- one array per class -- needs the sizes there anyway).

The reference reads its thresholds from the global cfg; here they are arguments with the reference's defaults
(TEST.SCORE_THRESH 0.05, TEST.NMS 0.5, TEST.DETECTIONS_PER_IM 100, TEST.SOFT_NMS.* : core/config.py:222-227,358-368).
Bounding-box voting (TEST.BBOX_VOTE, off by default) is not implemented.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .nms import SOFT_NMS_METHODS


def _to_device(a, device):
    if isinstance(a, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    return a.to(device=device, dtype=torch.float32)


def class_major_detections(scores, boxes, score_thresh):
    """dets [M,5] of all (class >= 1, RoI) pairs with score > score_thresh, class-major and RoI-ascending inside a class
    (the order `np.where(scores[:, j] > thresh)[0]` gives, core/test.py:749-752), plus the int32 segment offsets
    [num_classes] (offsets[j - 1] .. offsets[j] = class j) as a device tensor."""
    r, c = scores.shape
    valid = (scores[:, 1:] > score_thresh).t().contiguous()               # [C-1, R], class-major
    counts = valid.sum(dim=1, dtype=torch.int32)
    offsets = torch.zeros(c, dtype=torch.int32, device=scores.device)
    offsets[1:] = torch.cumsum(counts, 0)
    cls, roi = torch.nonzero(valid, as_tuple=True)                        # row-major: class, then RoI ascending
    box4 = boxes.view(r, c, 4)[roi, cls + 1]
    dets = torch.cat([box4, scores[roi, cls + 1].unsqueeze(1)], dim=1).contiguous()
    return dets, offsets


def box_results_with_nms_and_limit(scores, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100,
                                   soft_nms=False, soft_nms_sigma=0.5, soft_nms_method="linear", device=None):
    """core/test.py:732-790.  scores [R, C], boxes [R, 4C] (numpy or tensors; class 0 = background).  Returns
    (scores [D], boxes [D,4], cls_boxes) with cls_boxes[j] a float32 [k_j, 5] array (cls_boxes[0] == []), numpy out for
    numpy in and device tensors out for tensors in -- the same rows in the same order as the reference."""
    as_numpy = isinstance(scores, np.ndarray)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if as_numpy else scores.device
    scores_d, boxes_d = _to_device(scores, device), _to_device(boxes, device)
    _lib.require_cuda(scores_d, "scores")
    r, num_classes = scores_d.shape
    if boxes_d.shape != (r, 4 * num_classes):
        raise ValueError("boxes must be [R, 4 * num_classes]")
    lib = _lib.lib()
    dets, offsets = class_major_detections(scores_d, boxes_d, float(score_thresh))
    off = offsets.cpu().numpy()                                            # sync 1: the segment sizes
    ns = np.diff(off).astype(np.int64)
    nseg = num_classes - 1
    stream = _lib.current_stream_handle(device)
    if soft_nms:
        if soft_nms_method not in SOFT_NMS_METHODS:
            raise AssertionError("Unknown soft_nms method: {}".format(soft_nms_method))
        out_dets = torch.empty_like(dets)
        out_inds = torch.empty((dets.size(0),), dtype=torch.int64, device=device)
        num_out = torch.zeros((nseg,), dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            rc = lib.mi_soft_nms_segmented(dets.data_ptr(), offsets.data_ptr(), nseg, int(ns.max()) if nseg else 0,
                                           float(soft_nms_sigma), float(nms_thresh), 0.0001,   # core/test.py:758
                                           SOFT_NMS_METHODS[soft_nms_method], out_dets.data_ptr(), out_inds.data_ptr(),
                                           num_out.data_ptr(), stream)
        _lib.check(rc, "mi_soft_nms_segmented")
        kept = num_out.cpu().numpy()                                       # sync 2: rows per class
        per_class = [out_dets[off[j]:off[j] + kept[j]] for j in range(nseg)]
    else:
        keep_all = torch.empty((max(int(dets.size(0)), 1),), dtype=torch.int64, device=device)
        num_all = torch.zeros((nseg,), dtype=torch.int32, device=device)
        live = [j for j in range(nseg) if ns[j] > 0]
        if live:
            p = len(live)
            n_arr = (ctypes.c_int * p)(*[int(ns[j]) for j in live])
            ws_bytes = lib.mi_nms_batched_workspace_bytes(p, n_arr)
            workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=device)
            dets_arr = (ctypes.c_void_p * p)(*[dets.data_ptr() + int(off[j]) * 20 for j in live])
            keep_arr = (ctypes.c_void_p * p)(*[keep_all.data_ptr() + int(off[j]) * 8 for j in live])
            num_arr = (ctypes.c_void_p * p)(*[num_all.data_ptr() + 4 * j for j in live])
            with torch.cuda.device(device):
                rc = lib.mi_nms_batched(p, dets_arr, n_arr, float(nms_thresh), _lib.NMS_GE_ORIG_ASC, keep_arr, num_arr,
                                        workspace.data_ptr(), ws_bytes, stream)
            _lib.check(rc, "mi_nms_batched")
        kept = num_all.cpu().numpy()                                       # sync 2: rows per class
        # nms_dets = dets_j[keep, :]  (:765) -- keep holds ascending indices into the class's segment
        per_class = [dets[off[j]:off[j + 1]][keep_all[off[j]:off[j] + kept[j]]] for j in range(nseg)]
    # limit to detections_per_im over all classes (:776-785)
    if detections_per_im > 0 and int(kept.sum()) > detections_per_im:
        image_scores = torch.cat([d[:, 4] for d in per_class])
        image_thresh = torch.sort(image_scores)[0][-detections_per_im]
        per_class = [d[d[:, 4] >= image_thresh] for d in per_class]
    im_results = torch.cat(per_class, dim=0) if per_class else dets.new_zeros((0, 5))
    out_boxes, out_scores = im_results[:, :4], im_results[:, 4]
    if as_numpy:
        return (out_scores.cpu().numpy(), out_boxes.cpu().numpy(),
                [[]] + [d.cpu().numpy() for d in per_class])
    return out_scores, out_boxes, [[]] + per_class
