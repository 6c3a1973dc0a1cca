"""TEST INFRASTRUCTURE: oracle-backed stand-ins for the HIP operators so that the WIRING of the R-CNN graph (model,
labelling, losses) can be executed on CPU tensors and compared with the reference's own model code (oracle/ref_model.py).
The operators themselves are tested against the oracle on the GPU (tests/test_ops_gpu.py, tests/test_e2e_gpu.py); nothing
here is reachable from the product package."""
import contextlib

import numpy as np
import torch

import oracle
from oracle import proposals as oracle_proposals


class _RoIAlignOracle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, rois, ah, aw, scale, sr):
        ctx.cfg = (ah, aw, scale, sr, tuple(features.shape))
        ctx.save_for_backward(rois)
        return torch.from_numpy(oracle.roi_align_forward(features.detach().numpy(), rois.detach().numpy(), ah, aw,
                                                         scale, sr, threads=8))

    @staticmethod
    def backward(ctx, grad):
        ah, aw, scale, sr, shape = ctx.cfg
        (rois,) = ctx.saved_tensors
        g = oracle.roi_align_backward(grad.contiguous().numpy(), rois.numpy(), shape, scale, sr, threads=8)
        return torch.from_numpy(g), None, None, None, None, None


def roi_align_fpn(features, scales, rois, roi_levels, ah, aw, sr, prepared=None):
    """Same contract as roi_align.roi_align_fpn: output rows in the order of `rois`."""
    out = torch.zeros((rois.size(0), features[0].size(1), ah, aw))
    pieces, index = [], []
    # rows of no image (the padding rows of the static training path: image -1) pool zeros and get no gradient -- the
    # contract of the HIP operators; the reference's arithmetic, which the oracle restates, indexes out of bounds for them
    of_an_image = (rois[:, 0] >= 0) & (rois[:, 0] < features[0].size(0))
    for k, (f, sc) in enumerate(zip(features, scales)):
        idx = torch.nonzero((roi_levels == k) & of_an_image, as_tuple=False).flatten()
        if idx.numel():
            pieces.append(_RoIAlignOracle.apply(f, rois[idx].contiguous(), ah, aw, float(sc), sr))
            index.append(idx)
    if pieces:
        out = out.index_copy(0, torch.cat(index), torch.cat(pieces))
    return out


def generate_and_collect(ops, heads, im_info, post_nms_topN, static=False):
    """fpn_proposals.generate_and_collect through the numpy restatement of GenerateProposalsOp + collect."""
    info = im_info.numpy().astype(np.float32)
    rois, probs = [], []
    for op, (sc, dl) in zip(ops, heads):
        r, p = oracle_proposals.generate_proposals(sc.numpy(), dl.numpy(), info, op._anchors, 1.0 / op._feat_stride,
                                                   op.pre_nms_topN, op.post_nms_topN, op.nms_thresh, op.min_size)
        rois.append(r)
        probs.append(p)
    rois, scores = np.concatenate(rois), np.concatenate(probs).squeeze(1)
    inds = np.argsort(-scores)[:post_nms_topN]   # numpy default sort, as collect_and...py:85 (tied scores: order undefined)
    out = torch.from_numpy(rois[inds])
    if static:
        return out, torch.ones(out.size(0), dtype=torch.bool)
    return out


def bbox_overlaps(boxes, query):
    return torch.from_numpy(oracle.bbox_overlaps(boxes.numpy(), query.numpy()))


def polys_to_masks_wrt_boxes(packed, roi_inst, boxes, m):
    """segms.polys_to_masks_wrt_boxes through the oracle's restatement of utils/segms.py + pycocotools."""
    from oracle import mask_targets as oracle_segms

    pts, ps, ins = packed.points.numpy(), packed.poly_start.numpy(), packed.inst_start.numpy()
    out = np.zeros((boxes.size(0), m * m), dtype=np.int32)
    for r, i in enumerate(roi_inst.numpy()):
        if 0 <= i < len(ins) - 1:
            polys = [pts[ps[p]:ps[p + 1]].reshape(-1) for p in range(ins[i], ins[i + 1])]
            out[r] = oracle_segms.polys_to_mask_wrt_box(polys, boxes[r].numpy(), m).reshape(-1) > 0
    return torch.from_numpy(out)


@contextlib.contextmanager
def cpu_ops(model=None):
    from detectron_pytorch_amd import fpn_proposals, roi_xform

    saved = (fpn_proposals.generate_and_collect, roi_xform.roi_align_fpn, roi_xform.roi_align_fpn_supported)
    fpn_proposals.generate_and_collect = generate_and_collect
    roi_xform.roi_align_fpn = roi_align_fpn
    roi_xform.roi_align_fpn_supported = lambda *a, **k: True
    if model is not None:
        model.iou_fn = bbox_overlaps
        model.rasterize_fn = polys_to_masks_wrt_boxes
    try:
        yield
    finally:
        fpn_proposals.generate_and_collect, roi_xform.roi_align_fpn, roi_xform.roi_align_fpn_supported = saved
