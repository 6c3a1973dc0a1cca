"""Overlay for lib/modeling/generate_proposals.py (FPN.py:13,350 and rpn_heads.py:7,58 construct
`GenerateProposalsOp(anchors, spatial_scale)` and call it with (rpn_cls_prob, rpn_bbox_pred, im_info)).

Same constructor, same call, same numpy return values; the mode-dependent settings are read from the reference's own
cfg at call time exactly where the reference reads them (generate_proposals.py:106-111); the work runs on the device
(detectron_pytorch_amd.generate_proposals)."""
from torch import nn

from core.config import cfg
from detectron_pytorch_amd.generate_proposals import GenerateProposalsOp as _DeviceOp


class GenerateProposalsOp(nn.Module):
    def __init__(self, anchors, spatial_scale):
        super().__init__()
        self._anchors = anchors
        self._num_anchors = self._anchors.shape[0]
        self._spatial_scale = spatial_scale
        self._feat_stride = 1. / spatial_scale

    def forward(self, rpn_cls_prob, rpn_bbox_pred, im_info):
        cfg_key = 'TRAIN' if self.training else 'TEST'
        op = _DeviceOp(self._anchors, self._spatial_scale, cfg[cfg_key].RPN_PRE_NMS_TOP_N,
                       cfg[cfg_key].RPN_POST_NMS_TOP_N, cfg[cfg_key].RPN_NMS_THRESH, cfg[cfg_key].RPN_MIN_SIZE)
        return op(rpn_cls_prob.data, rpn_bbox_pred.data, im_info)  # Note: ndarrays, as the reference returns
