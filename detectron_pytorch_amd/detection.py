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
    """All (class >= 1, RoI) pairs with score > score_thresh as ONE array, class-major and RoI-ascending inside a class
    (the order `np.where(scores[:, j] > thresh)[0]` gives, core/test.py:749-752).  Returns dets [M,5], the 0-based
    class of every row [M] and the int32 segment offsets [num_classes] (rows offsets[j-1] .. offsets[j] = class j)."""
    r, c = scores.shape
    valid = (scores[:, 1:] > score_thresh).t().contiguous()               # [C-1, R], class-major
    counts = valid.sum(dim=1, dtype=torch.int32)
    offsets = torch.zeros(c, dtype=torch.int32, device=scores.device)
    offsets[1:] = torch.cumsum(counts, 0)
    cls, roi = torch.nonzero(valid, as_tuple=True)                        # row-major: class, then RoI ascending
    box4 = boxes.view(r, c, 4)[roi, cls + 1]
    dets = torch.cat([box4, scores[roi, cls + 1].unsqueeze(1)], dim=1).contiguous()
    return dets, cls, offsets


def box_results_with_nms_and_limit(scores, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100,
                                   soft_nms=False, soft_nms_sigma=0.5, soft_nms_method="linear", device=None):
    """core/test.py:732-790.  scores [R, C], boxes [R, 4C] (numpy or tensors; class 0 = background).  Returns
    (scores [D], boxes [D,4], cls_boxes) with cls_boxes[j] a float32 [k_j, 5] array (cls_boxes[0] == []), numpy out for
    numpy in and device tensors out for tensors in -- the same rows in the same order as the reference.

    Everything between the inputs and the final row selection works on the flat class-major array (a per-class Python
    loop of 80 slice / gather / mask operations costs more than the NMS itself): the kept rows of all classes are one
    boolean mask, the top-`detections_per_im` cut is one threshold, and the per-class results are views into the one
    gathered result."""
    as_numpy = isinstance(scores, np.ndarray)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if as_numpy else scores.device
    scores_d, boxes_d = _to_device(scores, device), _to_device(boxes, device)
    _lib.require_cuda(scores_d, "scores")
    r, num_classes = scores_d.shape
    if boxes_d.shape != (r, 4 * num_classes):
        raise ValueError("boxes must be [R, 4 * num_classes]")
    lib = _lib.lib()
    nseg = num_classes - 1
    dets, seg, offsets = class_major_detections(scores_d, boxes_d, float(score_thresh))
    m = int(dets.size(0))
    stream = _lib.current_stream_handle(device)
    row_off = offsets[:-1].long()[seg] if m else seg                       # first row of each row's segment
    slot = torch.arange(m, device=device) - row_off if m else seg
    if m == 0:
        rows, kept_mask = dets, torch.zeros((0,), dtype=torch.bool, device=device)
    elif soft_nms:
        if soft_nms_method not in SOFT_NMS_METHODS:
            raise AssertionError("Unknown soft_nms method: {}".format(soft_nms_method))
        rows = torch.empty_like(dets)
        out_inds = torch.empty((m,), dtype=torch.int64, device=device)
        num_out = torch.zeros((nseg,), dtype=torch.int32, device=device)
        # a segment cannot be longer than the number of RoIs: no host copy of the sizes unless R exceeds the kernel's
        # capacity (4096 rows per class), in which case the true longest segment decides -- and an oversized class is an
        # explicit error (the library's), never a silently truncated one
        max_seg = r if r <= 4096 else int((offsets[1:] - offsets[:-1]).max().item())
        with torch.cuda.device(device):
            rc = lib.mi_soft_nms_segmented(dets.data_ptr(), offsets.data_ptr(), nseg, max_seg,
                                           float(soft_nms_sigma), float(nms_thresh), 0.0001,   # core/test.py:758
                                           SOFT_NMS_METHODS[soft_nms_method], rows.data_ptr(), out_inds.data_ptr(),
                                           num_out.data_ptr(), stream)
        _lib.check(rc, "mi_soft_nms_segmented")
        kept_mask = slot < num_out[seg]                                    # each class's result is a prefix of its segment
    else:
        off = offsets.cpu().numpy()                                        # the batched entry takes host-side sizes
        ns = np.diff(off)
        keep_all = torch.zeros((m,), dtype=torch.int64, device=device)
        num_all = torch.zeros((nseg,), dtype=torch.int32, device=device)
        live = [j for j in range(nseg) if ns[j] > 0]
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
        # nms_dets = dets_j[keep, :] (:765): keep = ascending indices into the class's segment, the first num_all[j]
        # slots of the segment's part of keep_all -> one mask over the flat array
        rows = dets
        src = torch.where(slot < num_all[seg], row_off + keep_all, torch.full_like(slot, m))
        kept_mask = torch.zeros((m + 1,), dtype=torch.bool, device=device)
        kept_mask[src] = True
        kept_mask = kept_mask[:m]
    # limit to detections_per_im over all classes (:776-785): image_thresh = the D-th largest kept score
    if detections_per_im > 0 and m > detections_per_im:
        masked = torch.where(kept_mask, rows[:, 4], torch.full_like(rows[:, 4], float("-inf")))
        image_thresh = torch.topk(masked, detections_per_im).values[-1]    # -inf when fewer than D rows are kept
        kept_mask = kept_mask & (rows[:, 4] >= image_thresh)
    final = rows[kept_mask]                                                # class-major, original order inside a class
    counts = torch.bincount(seg[kept_mask], minlength=nseg).cpu().tolist() if m else [0] * nseg
    if as_numpy:
        final_np = final.cpu().numpy()
        per_class = np.split(final_np, np.cumsum(counts)[:-1]) if nseg else []
        return final_np[:, 4], final_np[:, :4], [[]] + per_class
    return final[:, 4], final[:, :4], [[]] + list(torch.split(final, counts))
