"""RPN proposal generation on the device: the direct caller of the RPN NMS.

Mirrors `GenerateProposalsOp` (lib/modeling/generate_proposals.py:12-182) and `generate_anchors`
(lib/modeling/generate_anchors.py:54-123).  The reference copies the RPN outputs to the host and runs numpy +
cython_nms per image (:58-63, five levels x images per step); here, per level and for all images at once:

    mi_topk_batched over the [A*H*W] scores  ->  mi_rpn_decode_proposals (anchor + deltas -> box, clip, filter; one launch)
    ->  mi_nms_batched (one problem per image, input already score-sorted)  ->  the first post_nms_topN kept and valid.

No device-to-host copy before the final sizes are needed (the reference's return type is a concatenated array).
cfg values (TRAIN/TEST.RPN_PRE_NMS_TOP_N, RPN_POST_NMS_TOP_N, RPN_NMS_THRESH, RPN_MIN_SIZE, BBOX_XFORM_CLIP) are
constructor arguments with the reference's defaults (core/config.py:132-149,936).

Parity: the decode follows the arithmetic types numpy >= 2 gives the reference's expressions (fp32, with the
width/height branch in fp64 -- see csrc/proposals.hip); boxes and kept sets equal the reference's on the fixtures.
The order of tied scores is undefined in the reference (np.argsort of the negated scores); here: lower index first.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import topk as topk_mod
from .nms import nms_device_many

BBOX_XFORM_CLIP = float(np.log(1000.0 / 16.0))  # core/config.py:936


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """lib/modeling/generate_anchors.py:54-123: [len(aspect_ratios) * len(sizes), 4] float64 anchors (x1, y1, x2, y2)
    centred on stride / 2, ratio-major."""
    scales = np.asarray(sizes, dtype=np.float64) / stride
    ratios = np.asarray(aspect_ratios, dtype=np.float64)
    centre = 0.5 * (stride - 1)

    def boxes(ws, hs):
        return np.stack([centre - 0.5 * (ws - 1), centre - 0.5 * (hs - 1), centre + 0.5 * (ws - 1),
                         centre + 0.5 * (hs - 1)], axis=1)

    ws = np.round(np.sqrt(float(stride) * stride / ratios))  # one box per ratio with (about) the base area
    hs = np.round(ws * ratios)
    return np.vstack([boxes(w * scales, h * scales) for w, h in zip(ws, hs)])


class GenerateProposalsOp(object):
    """`GenerateProposalsOp(anchors, spatial_scale)(rpn_cls_prob, rpn_bbox_pred, im_info)` -> (rois [R,5], probs [R,1]),
    numpy float32 like the reference (`as_numpy=False`: device tensors)."""

    def __init__(self, anchors, spatial_scale, pre_nms_topN=12000, post_nms_topN=2000, nms_thresh=0.7, min_size=0,
                 as_numpy=True):
        self._anchors = np.ascontiguousarray(anchors, dtype=np.float64)
        self._num_anchors = self._anchors.shape[0]
        self._feat_stride = 1.0 / spatial_scale
        self.pre_nms_topN, self.post_nms_topN = int(pre_nms_topN), int(post_nms_topN)
        self.nms_thresh, self.min_size, self.as_numpy = float(nms_thresh), float(min_size), as_numpy

    def __call__(self, rpn_cls_prob, rpn_bbox_pred, im_info):
        return self.forward(rpn_cls_prob, rpn_bbox_pred, im_info)

    def num_candidates(self, total):
        """Pre-NMS candidates of one image of this level: min(pre_nms_topN, A*H*W) (:131-134)."""
        return total if (self.pre_nms_topN <= 0 or self.pre_nms_topN >= total) else self.pre_nms_topN

    def decode(self, rpn_cls_prob, rpn_bbox_pred, im_info, top=None):
        """Steps 1-5 (:113-153): top-k, decode, clip, filter.  Returns dets [N,k,5] (score-descending, rejected boxes
        parked far away) and valid [N,k] int32; nothing is copied to the host.  `top`: (scores [N,k], flat indices [N,k])
        when the caller has already selected (fpn_proposals does it for all levels in one call)."""
        _lib.require_cuda(rpn_cls_prob, "rpn_cls_prob")
        scores = rpn_cls_prob.detach().contiguous()
        deltas = rpn_bbox_pred.detach().contiguous()
        device = scores.device
        if isinstance(im_info, np.ndarray):
            im_info = torch.from_numpy(im_info)
        im_info = im_info.detach().to(device=device, dtype=torch.float32).contiguous()
        n, a, h, w = scores.shape
        if a != self._num_anchors or deltas.shape != (n, 4 * a, h, w) or scores.dtype != torch.float32:
            raise ValueError("rpn_cls_prob [N,A,H,W] / rpn_bbox_pred [N,4A,H,W] float32 expected for %d anchors"
                             % self._num_anchors)
        total = a * h * w
        k = self.num_candidates(total)
        if top is None:
            if topk_mod.supported(total, k) and n > 0:
                top_scores, top_idx = topk_mod.topk_rows(scores.view(n, total), k)
            else:
                top_scores, top_idx = torch.topk(scores.view(n, total), k, dim=1, largest=True, sorted=True)  # :131-142
        else:
            top_scores, top_idx = top                                          # [n, k] each, from one batched call
        top_scores, top_idx = top_scores.contiguous(), top_idx.contiguous()
        dets = torch.empty((n, k, 5), dtype=torch.float32, device=device)
        valid = torch.empty((n, k), dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            rc = _lib.lib().mi_rpn_decode_proposals(
                deltas.data_ptr(), top_scores.data_ptr(), top_idx.data_ptr(), im_info.data_ptr(),
                self._anchors.ctypes.data_as(ctypes.c_void_p), n, a, h, w, k, float(self._feat_stride), self.min_size,
                BBOX_XFORM_CLIP, dets.data_ptr(), valid.data_ptr(), _lib.current_stream_handle(device))
        _lib.check(rc, "mi_rpn_decode_proposals")
        return dets, valid

    def select(self, dets, valid, kept):
        """Steps 6-8 (:155-161) as a mask [N,k]: kept by the NMS (`kept`: one (keep, num_keep) per image, or None when
        nms_thresh <= 0), passed the filter, among the first post_nms_topN of those.  All images at once, tensor-only
        arguments (no host scalar is uploaded: the sequence can be captured in a hipGraph)."""
        n, k = valid.shape
        if kept is None:
            return valid.bool()
        keep = torch.stack([kp[:k] for kp, _ in kept])                        # [N,k] ascending positions = descending score
        num_keep = torch.cat([nm.reshape(1) for _, nm in kept]).view(n, 1).to(torch.int64)
        pos = torch.arange(k, device=valid.device).view(1, k)
        idx = torch.where(pos < num_keep, keep, torch.full_like(keep, k))     # column k absorbs the unused tail of `keep`
        take = torch.zeros((n, k + 1), dtype=torch.bool, device=valid.device).scatter_(1, idx, True)
        take = take[:, :k] & valid.bool()
        if self.post_nms_topN > 0:
            take &= torch.cumsum(take, dim=1) <= self.post_nms_topN
        return take

    def forward(self, rpn_cls_prob, rpn_bbox_pred, im_info):
        dets, valid = self.decode(rpn_cls_prob, rpn_bbox_pred, im_info)
        n = dets.size(0)
        kept = (nms_device_many([dets[i] for i in range(n)], self.nms_thresh, _lib.NMS_GE_ORIG_ASC)
                if self.nms_thresh > 0 else None)
        take = self.select(dets, valid, kept)
        img, col = torch.nonzero(take, as_tuple=True)                        # image-major, score-descending
        boxes = dets[img, col]
        rois = torch.cat([img.to(torch.float32).unsqueeze(1), boxes[:, :4]], dim=1)
        probs = boxes[:, 4:5]
        if self.as_numpy:
            return rois.cpu().numpy(), probs.cpu().numpy()
        return rois, probs
