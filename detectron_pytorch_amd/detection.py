"""Per-class detection post-processing on the device: the caller of NMS at test time (call
site core/test.py:764; section 8f row 3).

Mirrors `box_results_with_nms_and_limit` (lib/core/test.py:732-790): threshold the class scores, NMS (or Soft-NMS) per
class, then keep the `detections_per_im` best over all classes.  The reference loops over the 80 classes on the host,
one cython_nms call each; here the classes are the segments of one call -- `mi_nms_segmented` (hard: reads the score and
box blobs in place, class sizes stay on the device) or `mi_soft_nms_segmented` (soft) -- and run side by side; only the
result sizes cross to the host (the reference's return type, one array per class, needs them there anyway).

The reference reads its thresholds from the global cfg; here they are arguments with the reference's defaults
(TEST.SCORE_THRESH 0.05, TEST.NMS 0.5, TEST.DETECTIONS_PER_IM 100, TEST.SOFT_NMS.* : core/config.py:222-227,358-368).
Bounding-box voting (TEST.BBOX_VOTE, core/test.py:766-773, off by default) runs as one more call over all classes
(`mi_box_voting`) between the NMS and the detections_per_im cut.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import topk as topk_mod
from .nms import SOFT_NMS_METHODS


VOTE_SCORING_METHODS = {"ID": 0, "TEMP_AVG": 1, "AVG": 2, "IOU_AVG": 3, "GENERALIZED_AVG": 4, "QUASI_SUM": 5}  # boxes.py:285-311


def box_voting(top_dets, all_dets, thresh, scoring_method="ID", beta=1.0, top_segments=None, all_offsets=None):
    """utils.boxes.box_voting (lib/utils/boxes.py:268-317) on the device: every row of top_dets [K,5] is refined by the
    rows of all_dets [M,5] with IoU >= thresh (score-weighted box average; the score per `scoring_method`).  numpy in ->
    numpy out, device tensors in -> device tensor out.  top_segments [K] / all_offsets [S+1] (int32 tensors) run the
    classes of an image side by side: a top row is voted on by the rows of its own segment only."""
    if scoring_method not in VOTE_SCORING_METHODS:
        raise NotImplementedError("Unknown scoring method {}".format(scoring_method))       # boxes.py:312-315
    as_numpy = isinstance(top_dets, np.ndarray)
    device = torch.device("cuda", torch.cuda.current_device()) if as_numpy else top_dets.device
    top, alld = _to_device(top_dets, device).contiguous(), _to_device(all_dets, device).contiguous()
    _lib.require_cuda(top, "top_dets")
    k, m = int(top.size(0)), int(alld.size(0))
    out = torch.empty_like(top)
    if k:
        if top_segments is None:
            top_segments = torch.zeros((k,), dtype=torch.int32, device=device)
            all_offsets = torch.tensor([0, m], dtype=torch.int32, device=device)
        top_segments = top_segments.to(dtype=torch.int32).contiguous()
        all_offsets = all_offsets.to(dtype=torch.int32).contiguous()
        with torch.cuda.device(device):
            rc = _lib.lib().mi_box_voting(top.data_ptr(), top_segments.data_ptr(), k, alld.data_ptr(), all_offsets.data_ptr(),
                                          int(all_offsets.numel()) - 1, float(thresh), VOTE_SCORING_METHODS[scoring_method],
                                          float(beta), out.data_ptr(), _lib.current_stream_handle(device))
        _lib.check(rc, "mi_box_voting")
    return out.cpu().numpy() if as_numpy else out


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


TIE_SLACK = 28      # rows beyond DETECTIONS_PER_IM the static result can hold (scores tied with the 100th best)


def box_results_static(scores, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100, roi_valid=None):
    """core/test.py:732-790 (hard NMS) with shapes fixed by the inputs' shapes and NO host synchronisation: what a
    hipGraph of the whole detection step needs.  Three calls: mi_nms_segmented (threshold + per-class NMS on the blobs in
    place, emitting the surviving scores), mi_topk_batched (their detections_per_im + TIE_SLACK best), mi_detection_select
    (the image_thresh cut of :776-785, the gather, the per-class counts).  scores [R, C], boxes [R, 4C] device tensors;
    `roi_valid` [R] marks the rows of a static-size RoI blob that are real proposals.  Returns a dict of device tensors:
        dets [cap, 5], cls [cap] (class index, 1-based; 0 = unused row)   cap = detections_per_im + TIE_SLACK: the
            detections class-major and RoI-ascending inside a class -- the reference's row order -- first `count` rows
        sizes int64 [1 + C] = (count, total, detections of class 1 .. C-1); count: rows delivered, total: rows the
            reference returns (`total > count` only when more than TIE_SLACK scores tie with the detections_per_im-th
            best: the caller then takes the dynamic path).  `count`, `total`, `class_counts` are views of it."""
    r, c = scores.shape
    nseg = c - 1
    if roi_valid is not None:
        scores = torch.where(roi_valid.view(r, 1), scores, torch.full_like(scores, float("-inf")))
    scores, boxes = scores.contiguous(), boxes.contiguous()
    m = nseg * r
    d = detections_per_im if detections_per_im > 0 else m
    cap = min(m, d + TIE_SLACK)
    if cap > 1024 or not topk_mod.supported(m, cap):
        return _box_results_static_torch(scores, boxes, score_thresh, nms_thresh, detections_per_im)
    from .nms import nms_segmented
    _, _, masked = nms_segmented(scores, boxes, score_thresh, nms_thresh, with_masked_scores=True)
    vals, idx = topk_mod.topk(masked.view(m), cap)
    dets = torch.empty((cap, 5), dtype=torch.float32, device=scores.device)
    cls = torch.empty((cap,), dtype=torch.int32, device=scores.device)
    sizes = torch.empty((2 + nseg,), dtype=torch.int64, device=scores.device)
    with torch.cuda.device(scores.device):
        rc = _lib.lib().mi_detection_select(scores.data_ptr(), boxes.data_ptr(), masked.data_ptr(), vals.data_ptr(),
                                            idx.data_ptr(), r, c, cap, int(detections_per_im), dets.data_ptr(),
                                            cls.data_ptr(), sizes.data_ptr(), _lib.current_stream_handle(scores.device))
    _lib.check(rc, "mi_detection_select")
    return {"dets": dets, "cls": cls, "sizes": sizes, "count": sizes[0], "total": sizes[1], "class_counts": sizes[2:]}


def box_results_static_general(scores, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100, roi_valid=None,
                               soft_nms=False, soft_nms_sigma=0.5, soft_nms_method="linear", bbox_vote=False,
                               bbox_vote_thresh=0.8, bbox_vote_method="ID"):
    """core/test.py:732-790 with Soft-NMS (:753-760) and / or bounding-box voting (:766-773), shapes fixed by the inputs'
    shapes and NO host synchronisation -- `box_results_static` for the two options that re-score or move rows.  Same result
    dict; the rows of a class come in the reference's order (hard NMS: RoI-ascending; Soft-NMS: its pick order).

    How the variable sizes stay on the device: the (class, RoI) pairs above the score threshold are compacted class-major
    into a buffer of nseg * R rows by a prefix sum and one index_copy (rows that are no candidates land in a dump row);
    the class offsets are a device vector, which is all `mi_soft_nms_segmented` and `mi_box_voting` need.  Voting runs over
    EVERY buffer row with the rows that did not survive NMS pointed at an empty segment (they leave at once); the
    detections_per_im cut is a top-k of fixed size and a second prefix-sum compaction into the result rows."""
    r, c = scores.shape
    nseg = c - 1
    device = scores.device
    if r > 4096:
        raise NotImplementedError("the static Soft-NMS / voting path holds one class in one workgroup: at most 4096 RoIs")
    if roi_valid is not None:
        scores = torch.where(roi_valid.view(r, 1), scores, torch.full_like(scores, float("-inf")))
    scores, boxes = scores.contiguous(), boxes.contiguous()
    m = nseg * r
    valid = (scores[:, 1:] > score_thresh).t().contiguous()                   # [nseg, R], class-major
    vflat = valid.view(m)
    pos = torch.cumsum(vflat, 0) - 1
    idx = torch.where(vflat, pos, torch.full_like(pos, m))                    # candidates -> their rank; the rest -> dump row
    counts = valid.sum(dim=1, dtype=torch.int32)
    offsets = torch.zeros(nseg + 2, dtype=torch.int32, device=device)         # + one empty segment behind the classes
    offsets[1:nseg + 1] = torch.cumsum(counts, 0)
    offsets[nseg + 1] = offsets[nseg]
    cls_of = torch.arange(nseg, device=device).view(-1, 1).expand(nseg, r).reshape(m)
    pairs = torch.cat([boxes.view(r, c, 4)[:, 1:, :].permute(1, 0, 2).reshape(m, 4), scores[:, 1:].t().reshape(m, 1)], dim=1)
    cand = torch.zeros((m + 1, 5), dtype=torch.float32, device=device).index_copy_(0, idx, pairs)
    seg = torch.full((m + 1,), nseg, dtype=torch.int64, device=device).index_copy_(0, idx, cls_of)
    seg[m] = nseg                                                             # (the dump row took arbitrary writes)
    seg_m = seg[:m]
    lib, stream = _lib.lib(), _lib.current_stream_handle(device)
    if soft_nms:
        if soft_nms_method not in SOFT_NMS_METHODS:
            raise AssertionError("Unknown soft_nms method: {}".format(soft_nms_method))
        rows = torch.zeros((m + 1, 5), dtype=torch.float32, device=device)
        out_inds = torch.empty((m + 1,), dtype=torch.int64, device=device)
        num_out = torch.zeros((nseg + 1,), dtype=torch.int32, device=device)  # + the empty segment: 0
        with torch.cuda.device(device):
            rc = lib.mi_soft_nms_segmented(cand.data_ptr(), offsets.data_ptr(), nseg, r, float(soft_nms_sigma),
                                           float(nms_thresh), 0.0001, SOFT_NMS_METHODS[soft_nms_method], rows.data_ptr(),
                                           out_inds.data_ptr(), num_out.data_ptr(), stream)
        _lib.check(rc, "mi_soft_nms_segmented")
        slot = torch.arange(m, device=device) - offsets[:-1].long()[seg_m]
        kept = slot < num_out.long()[seg_m]                                   # each class's result is a prefix of its segment
        rows = rows[:m]
    else:
        from .nms import nms_segmented
        _, _, masked = nms_segmented(scores, boxes, score_thresh, nms_thresh, with_masked_scores=True)
        kept_pairs = masked.view(m) > float("-inf")                           # (class, RoI) layout -> compacted layout
        kept = torch.zeros((m + 1,), dtype=torch.bool, device=device).index_copy_(0, idx, kept_pairs & vflat)[:m]
        rows = cand[:m]
    if bbox_vote:
        voted = box_voting(rows, cand[:m], bbox_vote_thresh, bbox_vote_method,
                           top_segments=torch.where(kept, seg_m, torch.full_like(seg_m, nseg)), all_offsets=offsets)
        rows = torch.where(kept.view(m, 1), voted, rows)
    # limit to detections_per_im over all classes (:776-785): image_thresh = the D-th largest kept score, ties stay
    d = detections_per_im if detections_per_im > 0 else m
    cap = min(m, d + TIE_SLACK)
    sel = kept
    if d < m:
        masked_final = torch.where(kept, rows[:, 4], torch.full_like(rows[:, 4], float("-inf")))
        sel = kept & (rows[:, 4] >= torch.topk(masked_final, d).values[-1])
    total = sel.sum()
    outpos = torch.cumsum(sel, 0) - 1
    deliver = sel & (outpos < cap)
    oidx = torch.where(deliver, outpos, torch.full_like(outpos, cap))
    dets = torch.zeros((cap + 1, 5), dtype=torch.float32, device=device).index_copy_(0, oidx, rows)[:cap]
    cls = torch.zeros((cap + 1,), dtype=torch.int32, device=device).index_copy_(0, oidx, (seg_m + 1).to(torch.int32))[:cap]
    count = torch.clamp_max(total, cap)
    live = torch.arange(cap, device=device) < count                           # (the dump row's writes never reach [:cap], but
    dets = torch.where(live.view(cap, 1), dets, torch.zeros_like(dets))       # rows past `count` must read as unused)
    cls = torch.where(live, cls, torch.zeros_like(cls))
    class_counts = torch.zeros((nseg + 1,), dtype=torch.int64, device=device)
    class_counts.index_add_(0, torch.where(deliver, seg_m, torch.full_like(seg_m, nseg)), torch.ones_like(seg_m))
    sizes = torch.cat([count.view(1), total.view(1), class_counts[:nseg]])
    return {"dets": dets, "cls": cls, "sizes": sizes, "count": sizes[0], "total": sizes[1], "class_counts": sizes[2:]}


def _box_results_static_torch(scores, boxes, score_thresh, nms_thresh, detections_per_im):
    """`box_results_static` for results beyond one workgroup's reach (no detections_per_im limit: up to R * (C - 1)
    rows): the cut and the gather as tensor expressions, same outputs."""
    r, c = scores.shape
    nseg = c - 1
    from .nms import nms_segmented
    _, _, masked = nms_segmented(scores, boxes, score_thresh, nms_thresh, with_masked_scores=True)
    m = nseg * r
    masked = masked.view(m)
    d = detections_per_im if detections_per_im > 0 else m
    cap = min(m, d + TIE_SLACK)
    vals, idx = torch.topk(masked, cap, sorted=True)
    # image_thresh = the D-th best kept score (:781-784); everything >= it stays, ties included
    thresh = vals[d - 1] if d <= cap else vals.new_full((), float("-inf"))
    sel = (vals >= thresh) & (vals > float("-inf"))
    total = ((masked >= thresh) & (masked > float("-inf"))).sum()
    flat = torch.sort(torch.where(sel, idx, torch.full_like(idx, m))).values  # ascending flat index = class-major order
    valid = flat < m
    flat_c = torch.clamp_max(flat, m - 1)
    cls0 = flat_c // r
    roi = flat_c - cls0 * r
    box4 = boxes.view(r, c, 4)[roi, cls0 + 1]
    dets = torch.cat([box4, scores[roi, cls0 + 1].unsqueeze(1)], dim=1)
    dets = torch.where(valid.view(cap, 1), dets, torch.zeros_like(dets))
    class_counts = torch.zeros((nseg + 1,), dtype=torch.int64, device=scores.device)
    class_counts.index_add_(0, torch.where(valid, cls0, torch.full_like(cls0, nseg)), torch.ones_like(cls0))
    sizes = torch.cat([valid.sum().view(1), total.view(1), class_counts[:nseg]])
    return {"dets": dets, "cls": ((cls0 + 1) * valid).to(torch.int32), "sizes": sizes, "count": sizes[0],
            "total": sizes[1], "class_counts": sizes[2:]}


def _results_from_static(res, as_numpy):
    """The reference's return type from the static result: ONE device-to-host copy (the sizes), then views."""
    nseg = res["class_counts"].numel()
    sizes = res["sizes"].cpu().tolist()
    count, total, counts = sizes[0], sizes[1], sizes[2:]
    if total != count:
        return None                                                           # more ties than TIE_SLACK holds
    final = res["dets"][:count]
    if as_numpy:
        final_np = final.cpu().numpy()
        per_class = np.split(final_np, np.cumsum(counts)[:-1]) if nseg else []
        return final_np[:, 4], final_np[:, :4], [[]] + per_class
    return final[:, 4], final[:, :4], [[]] + list(torch.split(final, counts))


def box_results_with_nms_and_limit(scores, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100,
                                   soft_nms=False, soft_nms_sigma=0.5, soft_nms_method="linear", device=None,
                                   bbox_vote=False, bbox_vote_thresh=0.8, bbox_vote_method="ID"):
    """core/test.py:732-790.  scores [R, C], boxes [R, 4C] (numpy or tensors; class 0 = background).  Returns
    (scores [D], boxes [D,4], cls_boxes) with cls_boxes[j] a float32 [k_j, 5] array (cls_boxes[0] == []), numpy out for
    numpy in and device tensors out for tensors in -- the same rows in the same order as the reference.

    Hard NMS (the default): `box_results_static` -- threshold, per-class NMS (`mi_nms_segmented`, all classes side by
    side, class sizes never on the host), the detections_per_im cut and the final gather are one asynchronous sequence of
    fixed shapes; the only device-to-host copy is the vector of result sizes the reference's return type needs.
    Soft-NMS re-scores rows, so its candidates are compacted first (one more synchronisation for their number) and
    `mi_soft_nms_segmented` runs the classes side by side.  bbox_vote (TEST.BBOX_VOTE.ENABLED / VOTE_TH / SCORING_METHOD,
    core/test.py:766-773): the survivors of every class are refined by `mi_box_voting` against the class's candidates
    before the detections_per_im cut (compacting path; the call site passes no beta, so box_voting's default 1.0 applies)."""
    as_numpy = isinstance(scores, np.ndarray)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if as_numpy else scores.device
    scores_d, boxes_d = _to_device(scores, device), _to_device(boxes, device)
    _lib.require_cuda(scores_d, "scores")
    r, num_classes = scores_d.shape
    if boxes_d.shape != (r, 4 * num_classes):
        raise ValueError("boxes must be [R, 4 * num_classes]")
    nseg = num_classes - 1
    if r == 0 or nseg == 0:
        empty = torch.zeros((0, 5), dtype=torch.float32, device=device)
        if as_numpy:
            e = empty.cpu().numpy()
            return e[:, 4], e[:, :4], [[]] + [e] * nseg
        return empty[:, 4], empty[:, :4], [[]] + [empty] * nseg
    if not soft_nms and not bbox_vote and r <= 4096:
        out = _results_from_static(box_results_static(scores_d, boxes_d, score_thresh, nms_thresh, detections_per_im),
                                   as_numpy)
        if out is not None:
            return out
    lib = _lib.lib()
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
        # hard NMS beyond the static path's reach (more than 4096 RoIs, or a pile of tied scores at the cut): the
        # compacted class-major array through mi_nms_batched, which takes host-side sizes
        off = offsets.cpu().numpy()
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
    if bbox_vote and m:                                                    # :766-773
        voted = box_voting(rows[kept_mask], dets, bbox_vote_thresh, bbox_vote_method, top_segments=seg[kept_mask],
                           all_offsets=offsets)
        rows = rows.clone()
        rows[kept_mask] = voted
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
