"""Collect the per-level RPN proposals and hand them to the RoI heads without leaving the device: the inference path of
`CollectAndDistributeFpnRpnProposalsOp` (lib/modeling/collect_and_distribute_fpn_rpn_proposals.py:42-119; SURVEY.md
section 8f row 2, test-time half -- the training half samples with np.random and is not reproduced).

    collect      concatenate the levels' (rois, probs), keep the post_nms_topN best         (:83-98)
    distribute   FPN level of every RoI (utils/fpn.py:11-28), per-level blobs, restore permutation   (:101-119)

With the fused RoIAlign (`roi_align.roi_align_fpn`) the per-level blobs and the permutation are not needed at all --
it takes the collected RoIs and their level indices as they are; `distribute` still builds the reference's blob dict
for callers that want it.

Parity: `collect` is exact for untied scores.  The level of a RoI is floor(4 + log2(sqrt(area) / 224 + 1e-6)) in fp32;
sqrt, the division and the sum are correctly rounded on both sides, log2 is not (neither numpy's nor the device's), so
a RoI whose scale sits within an ulp of a power of two may land on the neighbouring level -- none does on the test
sets, and the fixture from the reference's own utils/fpn.py is matched.
"""
import ctypes

import torch

from . import _lib
from . import topk as topk_mod
from .nms import nms_device_many


def collect(rois_list, scores_list, post_nms_topN):
    """:83-98.  rois_list[i] [R_i,5], scores_list[i] [R_i,1] or [R_i] device tensors (one entry per RPN level).
    Returns the post_nms_topN best rois [R,5] in descending score order (the order of tied scores is as undefined as
    np.argsort(-scores) is there)."""
    rois = torch.cat(list(rois_list), dim=0)
    scores = torch.cat([s.reshape(-1) for s in scores_list], dim=0)
    k = min(int(post_nms_topN), scores.numel()) if post_nms_topN > 0 else scores.numel()
    inds = torch.topk(scores, k, largest=True, sorted=True).indices
    return rois[inds]


def map_rois_to_fpn_levels(rois_xyxy, k_min=2, k_max=5, s0=224.0, lvl0=4):
    """utils/fpn.py:11-28 on the device: int32 level of every box [R,4] (fp32 operations in the reference's order)."""
    w = rois_xyxy[:, 2] - rois_xyxy[:, 0] + 1
    h = rois_xyxy[:, 3] - rois_xyxy[:, 1] + 1
    areas = torch.clamp_min(w * h, 0)                       # areas[neg_idx] = 0 (:18)
    s = torch.sqrt(areas)
    lvls = torch.floor(lvl0 + torch.log2(s / s0 + 1e-6))
    return torch.clamp(lvls, k_min, k_max).to(torch.int32)


def distribute(rois, k_min=2, k_max=5):
    """:101-119 for inference: {'rois', 'rois_fpn<l>'..., 'rois_idx_restore_int32'} as device tensors, plus
    'roi_levels' (int32 level of every RoI in the order of 'rois')."""
    lvls = map_rois_to_fpn_levels(rois[:, 1:5], k_min, k_max)
    order = torch.argsort(lvls, stable=True)                 # level-major, original order inside a level
    counts = torch.bincount(lvls - k_min, minlength=k_max - k_min + 1).cpu().tolist()
    blobs = {"rois": rois, "roi_levels": lvls}
    for lvl, part in zip(range(k_min, k_max + 1), torch.split(rois[order], counts)):
        blobs["rois_fpn%d" % lvl] = part
    blobs["rois_idx_restore_int32"] = torch.argsort(order).to(torch.int32)
    return blobs


def collect_and_distribute(rois_list, scores_list, post_nms_topN, k_min=2, k_max=5):
    """CollectAndDistributeFpnRpnProposalsOp.forward in eval mode (:61-80)."""
    return distribute(collect(rois_list, scores_list, post_nms_topN), k_min, k_max)


def _fused_supported(ops, heads, post_nms_topN):
    if post_nms_topN <= 0 or post_nms_topN > topk_mod.MAX_K:
        return False
    for op, (sc, _) in zip(ops, heads):
        total = sc[0].numel()
        if sc.size(0) == 0 or not topk_mod.supported(total, op.num_candidates(total)):
            return False
    return True


def generate_and_collect(ops, heads, im_info, post_nms_topN, static=False, with_levels=False, k_min=2, k_max=5,
                         records=None):
    """GenerateProposals on every RPN level followed by `collect`, as ONE asynchronous pipeline of a dozen launches:
        mi_topk_batched            pre-NMS top-k of every (level, image) score map, all side by side
        mi_rpn_decode_proposals    anchors + deltas -> boxes, clip, size filter                       (per level)
        mi_nms_batched             all (level, image) problems side by side
        mi_rpn_collect_candidates  kept & valid & first post_nms_topN of each problem -> one flat candidate array
        mi_topk_batched            the post_nms_topN best of all levels and images (:83-98)
        mi_rpn_collect_finish      the RoI blob, its validity mask and the FPN level of every RoI (:101-119)
    The per-level RoI lists of :83-95 are never materialised and nothing is read back before the final count.
    `ops`: one generate_proposals.GenerateProposalsOp per level (same nms_thresh); `heads`: the matching (rpn_cls_prob,
    rpn_bbox_pred) pairs.  Returns rois [R,5] in descending score order, R <= post_nms_topN -- what `collect` returns
    for the reference's per-level outputs.  `static=True`: (rois [k,5], valid [k]) with k = min(post_nms_topN,
    candidates) fixed by the shapes alone and no host synchronisation at all; `with_levels=True` (static only): (rois,
    valid, levels int32 [k]) where the rows that are no proposals carry image index -1.
    `records` (static + with_levels only): a roi_align.PreparedRecords for the pooling call that will consume the blob --
    the last launch then also writes the RoIAlign records of the rows (mi_rpn_collect_finish_records) and `records.rois`
    is the returned blob; ignored when the shapes do not fit."""
    if not _fused_supported(ops, heads, post_nms_topN):
        return _generate_and_collect_torch(ops, heads, im_info, post_nms_topN, static, with_levels, k_min, k_max)
    lib = _lib.lib()
    device = heads[0][0].device
    # 1. one selection call for all (level, image) rows
    rows, ks, shapes = [], [], []
    for op, (sc, _) in zip(ops, heads):
        n = sc.size(0)
        flat = sc.detach().contiguous().view(n, -1)
        k = op.num_candidates(flat.size(1))
        rows += list(flat.unbind(0))
        ks += [k] * n
        shapes.append((n, k))
    vals, idx, offs = topk_mod.topk_flat(rows, ks)
    # 2. decode per level, 3. one batched NMS
    decoded, first = [], 0
    for op, (sc, dl), (n, k) in zip(ops, heads, shapes):
        lo, hi = offs[first], offs[first + n]
        decoded.append(op.decode(sc, dl, im_info, top=(vals[lo:hi].view(n, k), idx[lo:hi].view(n, k))))
        first += n
    thresh = ops[0].nms_thresh
    problems = [(dets[i], valid[i], i) for dets, valid in decoded for i in range(dets.size(0))]
    kept = nms_device_many([d for d, _, _ in problems], thresh, _lib.NMS_GE_ORIG_ASC) if thresh > 0 else None
    # 4. candidates of all problems, flat
    p = len(problems)
    total = sum(int(d.size(0)) for d, _, _ in problems)
    cand_scores = torch.empty((total,), dtype=torch.float32, device=device)
    cand_rois = torch.empty((total, 5), dtype=torch.float32, device=device)
    void_arr, int_arr = ctypes.c_void_p * p, ctypes.c_int * p
    stream = _lib.current_stream_handle(device)
    with torch.cuda.device(device):
        rc = lib.mi_rpn_collect_candidates(
            p, void_arr(*[d.data_ptr() for d, _, _ in problems]), void_arr(*[v.data_ptr() for _, v, _ in problems]),
            void_arr(*[kp.data_ptr() for kp, _ in kept]) if kept is not None else None,
            void_arr(*[nm.data_ptr() for _, nm in kept]) if kept is not None else None,
            int_arr(*[int(d.size(0)) for d, _, _ in problems]), int_arr(*[i for _, _, i in problems]),
            int(ops[0].post_nms_topN), cand_scores.data_ptr(), cand_rois.data_ptr(), stream)
    _lib.check(rc, "mi_rpn_collect_candidates")
    # 5. the best of all levels and images, 6. the RoI blob
    k = min(int(post_nms_topN), total)
    best, inds = topk_mod.topk(cand_scores, k)
    rois = torch.empty((k, 5), dtype=torch.float32, device=device)
    valid = torch.empty((k,), dtype=torch.bool, device=device)
    levels = torch.empty((k,), dtype=torch.int32, device=device)
    fuse_records = (records is not None and static and with_levels and records.supported and records.num_rois == k
                    and len(records.features) == int(k_max) - int(k_min) + 1)
    with torch.cuda.device(device):
        if fuse_records:
            ah, aw, sr = records.cfg
            rc = lib.mi_rpn_collect_finish_records(best.data_ptr(), inds.data_ptr(), cand_rois.data_ptr(), k, 1, int(k_min),
                                                   int(k_max), 224.0, 4.0, rois.data_ptr(), valid.data_ptr(),
                                                   levels.data_ptr(), ctypes.byref(records.table), records.batch,
                                                   records.channels, ah, aw, sr, records.layout,
                                                   records.workspace.data_ptr(), records.workspace.numel(), stream)
            records.rois = rois
        else:
            rc = lib.mi_rpn_collect_finish(best.data_ptr(), inds.data_ptr(), cand_rois.data_ptr(), k,
                                           1 if (static and with_levels) else 0, int(k_min), int(k_max), 224.0, 4.0,
                                           rois.data_ptr(), valid.data_ptr(), levels.data_ptr(), stream)
    _lib.check(rc, "mi_rpn_collect_finish")
    if static:
        return (rois, valid, levels) if with_levels else (rois, valid)
    count = int(valid.sum().item())                                # the one synchronisation
    return rois[:count]


def _generate_and_collect_torch(ops, heads, im_info, post_nms_topN, static, with_levels, k_min, k_max):
    """The same pipeline with torch.topk and tensor expressions for the selections: sizes beyond the fused kernels'
    (more than 4096 candidates per problem or RoIs per batch -- no FPN configuration of the reference gets there)."""
    decoded = [op.decode(sc, dl, im_info) for op, (sc, dl) in zip(ops, heads)]
    thresh = ops[0].nms_thresh
    if thresh > 0:
        problems = [dets[i] for dets, _ in decoded for i in range(dets.size(0))]
        kept = nms_device_many(problems, thresh, _lib.NMS_GE_ORIG_ASC)
    boxes, scores, first = [], [], 0
    for op, (dets, valid) in zip(ops, decoded):
        n, k = valid.shape
        take = op.select(dets, valid, kept[first:first + n] if thresh > 0 else None)
        first += n
        img = torch.arange(n, device=dets.device, dtype=torch.float32).view(n, 1, 1).expand(n, k, 1)
        boxes.append(torch.cat([img, dets[:, :, :4]], dim=2).reshape(n * k, 5))
        scores.append(torch.where(take, dets[:, :, 4], torch.full_like(dets[:, :, 4], float("-inf"))).reshape(n * k))
    boxes, scores = torch.cat(boxes), torch.cat(scores)
    k = min(int(post_nms_topN), scores.numel()) if post_nms_topN > 0 else scores.numel()
    best, inds = torch.topk(scores, k, largest=True, sorted=True)
    if static:
        rois, valid = boxes[inds], best > float("-inf")
        if not with_levels:
            return rois, valid
        rois = torch.cat([torch.where(valid, rois[:, 0], torch.full_like(rois[:, 0], -1.0)).view(-1, 1), rois[:, 1:5]], 1)
        return rois, valid, map_rois_to_fpn_levels(rois[:, 1:5], k_min, k_max)
    count = int((best > float("-inf")).sum().item())            # the one synchronisation
    return boxes[inds[:count]]
