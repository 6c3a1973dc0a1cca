"""Generalized R-CNN: the graph `BASELINE.json:metric` is quoted on (reference: lib/modeling/model_builder.py:71-250
`Generalized_RCNN.__init__/_forward`, :326-348 the inference entry points).

Same sub-module names (Conv_Body, RPN, Box_Head, Box_Outs, Mask_Head, Mask_Outs, Keypoint_Head, Keypoint_Outs) and the
same construction order as the reference: a reference checkpoint's `state_dict` loads unchanged, and a build under
`torch.manual_seed(cfg.RNG_SEED)` draws the reference's initial weights (tests/test_model_cpu.py).

What differs is everything between the convolutions and the heads.  The reference leaves the device there
(generate_proposals.py:58-63 D2H per level, numpy top-k / decode / cython NMS per image, numpy collect, numpy labelling
and sampling, H2D of the RoI blobs per level and per head: SURVEY.md section 3.1); here the RPN outputs never leave HBM:

    RPN convs -> fpn_proposals.generate_and_collect   (HIP decode + one batched HIP NMS for all levels and images)
              -> targets.label_proposals               (training: HIP IoU matrix + tensor ops, static shapes)
              -> roi_xform.roi_feature_transform       (one fused HIP RoIAlign over P2-P5 per head)

The training forward has no device-to-host synchronisation.
"""
import os

import torch
import torch.nn as nn

from .. import fpn_proposals, nms, roi_xform, segms
from .. import roi_align as roi_align_mod
from . import fpn as fpn_mod
from . import heads, targets

# 1 = the proposal stage's last launch writes the RoIAlign records of the blob it emits (round 5: one launch fewer than a
# pooling call that starts with its records launch).  Default 0 since the records-free forward (late round 6): the pooling call
# of an inference pass is ONE launch that needs no records -- 58.6-63.2 us for collect + pooling against 65.2-65.7 with
# producer-written records (bench.py inference_path.records_by_producer).  Kept as the A/B switch of the measurement tools.
PRODUCER_RECORDS = os.environ.get("MI_RCNN_PRODUCER_RECORDS", "0") != "0"


def _conv_body(cfg):
    name = cfg.MODEL.CONV_BODY                      # e.g. "FPN.fpn_ResNet50_conv5_body" (model_builder.py:25-43 get_func)
    if not name.startswith("FPN.fpn_") or not cfg.FPN.FPN_ON:
        raise NotImplementedError("only FPN conv bodies are built (CONV_BODY=%r); the C4 bodies are out of scope" % name)
    body = name[len("FPN.fpn_"):]
    if body.endswith("_P2only_body"):
        raise NotImplementedError("P2only pyramids are not built")
    return fpn_mod.FPN(body, cfg)


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if not (cfg.RPN.RPN_ON and cfg.FPN.FPN_ON and cfg.MODEL.FASTER_RCNN) or cfg.MODEL.RPN_ONLY:
            raise NotImplementedError("end-to-end Faster / Mask / Keypoint R-CNN with FPN only (the BASELINE configs)")
        self.Conv_Body = _conv_body(cfg)
        self.RPN = fpn_mod.FpnRpnOutputs(self.Conv_Body.dim_out, self.Conv_Body.spatial_scale, cfg)
        # model_builder.py:88-99: the RPN may use one more (coarser) level than the RoI heads
        assert cfg.FPN.RPN_MIN_LEVEL == cfg.FPN.ROI_MIN_LEVEL and cfg.FPN.RPN_MAX_LEVEL >= cfg.FPN.ROI_MAX_LEVEL
        self.num_roi_levels = cfg.FPN.ROI_MAX_LEVEL - cfg.FPN.ROI_MIN_LEVEL + 1
        self.roi_scales = self.Conv_Body.spatial_scale[-self.num_roi_levels:]
        if cfg.FAST_RCNN.ROI_BOX_HEAD.split(".")[-1] != "roi_2mlp_head":
            raise NotImplementedError("box head %r" % cfg.FAST_RCNN.ROI_BOX_HEAD)
        self.Box_Head = heads.Roi2MlpHead(self.RPN.dim_out, self.roi_feature_transform, self.roi_scales, cfg)
        self.Box_Outs = heads.FastRcnnOutputs(self.Box_Head.dim_out, cfg)
        if cfg.MODEL.MASK_ON:
            head = cfg.MRCNN.ROI_MASK_HEAD.split(".")[-1]
            convs = {"mask_rcnn_fcn_head_v1up4convs": 4, "mask_rcnn_fcn_head_v1up": 2}
            if head not in convs:
                raise NotImplementedError("mask head %r" % cfg.MRCNN.ROI_MASK_HEAD)
            self.Mask_Head = heads.MaskRcnnFcnHeadV1upXconvs(self.RPN.dim_out, self.roi_feature_transform,
                                                             self.roi_scales, convs[head], cfg)
            self.Mask_Outs = heads.MaskRcnnOutputs(self.Mask_Head.dim_out, cfg)
        if cfg.MODEL.KEYPOINTS_ON:
            if cfg.KRCNN.ROI_KEYPOINTS_HEAD.split(".")[-1] != "roi_pose_head_v1convX":
                raise NotImplementedError("keypoint head %r" % cfg.KRCNN.ROI_KEYPOINTS_HEAD)
            self.Keypoint_Head = heads.RoiPoseHeadV1convX(self.RPN.dim_out, self.roi_feature_transform,
                                                          self.roi_scales, cfg)
            self.Keypoint_Outs = heads.KeypointOutputs(self.Keypoint_Head.dim_out, cfg)
        if cfg.TRAIN.FREEZE_CONV_BODY:
            for p in self.Conv_Body.parameters():
                p.requires_grad = False
        self.iou_fn = nms.bbox_overlaps                 # mi_bbox_overlaps; tests on CPU tensors inject the oracle's
        self.rasterize_fn = segms.polys_to_masks_wrt_boxes   # mi_polys_to_masks_wrt_boxes (roidb["gt_polygons"]); likewise
        self.mark = None                                # optional callable(label) invoked at stage boundaries (bench.py)
        self.static_inference = False                   # eval forward with fixed shapes and no host sync (inference.py)

    def _mark(self, label):
        if self.mark is not None:
            self.mark(label)

    # ---- the boundary the reference's heads call (model_builder.py:252-324) ------------------------------------------
    def roi_feature_transform(self, blobs_in, rpn_ret, blob_rois="rois", method="RoIPoolF", resolution=7,
                              spatial_scale=1. / 16., sampling_ratio=0):
        return roi_xform.roi_feature_transform(blobs_in, rpn_ret, blob_rois, method, resolution, spatial_scale,
                                               sampling_ratio, self.cfg.FPN.ROI_MIN_LEVEL, self.cfg.FPN.ROI_MAX_LEVEL)

    # ---- proposals -----------------------------------------------------------------------------------------------------
    def proposals(self, rpn_ret, im_info, static, with_levels=False, records=None):
        """FPN.py:390-417 without leaving the device: objectness (sigmoid / softmax), GenerateProposals on every level, collect."""
        cfg = self.cfg
        heads_ = [(fpn_mod.rpn_cls_probs(cfg, rpn_ret["rpn_cls_logits_fpn%d" % lvl].detach().float()).contiguous(),
                   rpn_ret["rpn_bbox_pred_fpn%d" % lvl].detach().float().contiguous())
                  for lvl in range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1)]
        key = "TRAIN" if self.training else "TEST"
        post = int(cfg[key].RPN_POST_NMS_TOP_N * cfg.FPN.RPN_COLLECT_SCALE + 0.5)      # collect_and...py:74
        if with_levels:
            return fpn_proposals.generate_and_collect(self.RPN.proposal_ops(self.training), heads_, im_info, post,
                                                      static=True, with_levels=True, k_min=cfg.FPN.ROI_MIN_LEVEL,
                                                      k_max=cfg.FPN.ROI_MAX_LEVEL, records=records)
        return fpn_proposals.generate_and_collect(self.RPN.proposal_ops(self.training), heads_, im_info, post,
                                                  static=static)

    def forward(self, data, im_info, roidb=None, rpn_targets=None, priority=None):
        """Training (`self.training`): `roidb` = dict of device tensors {'gt_boxes' [G,4] (original image coordinates),
        'gt_classes' [G], 'gt_image' [G]} for the minibatch, `rpn_targets` = the data layer's wide RPN blobs
        (fpn.fpn_rpn_losses), `priority` = [G + R_collect] sampling priorities (None: drawn on the device).
        Returns {'losses': {...}, 'metrics': {...}} like model_builder.py:183-242 (scalars, not [1] tensors: there is no
        cross-GPU gather of them here).
        Inference: returns {'rois', 'cls_score', 'bbox_pred', 'blob_conv'} like :244-248."""
        with torch.set_grad_enabled(self.training):
            return self._forward(data, im_info, roidb, rpn_targets, priority)

    def _forward(self, data, im_info, roidb, rpn_targets, priority):
        self._mark("start")
        blob_conv = self.Conv_Body(data)
        self._mark("backbone")
        rpn_ret = self.RPN(blob_conv)
        self._mark("rpn_convs")
        return self.forward_from_features(blob_conv, rpn_ret, im_info, roidb, rpn_targets, priority)

    def forward_from_features(self, blob_conv, rpn_ret, im_info, roidb=None, rpn_targets=None, priority=None):
        """Everything after the convolutions of the backbone and the RPN head (model_builder.py:157-248): proposals,
        labelling, RoI heads, losses.  Separate so that this half -- the half that differs from the reference -- can be fed
        identical inputs on two devices."""
        cfg = self.cfg
        device = blob_conv[0].device
        n_img = blob_conv[0].size(0)
        im_info_d = im_info.to(device=device, dtype=torch.float32) if torch.is_tensor(im_info) else \
            torch.as_tensor(im_info, dtype=torch.float32, device=device)
        roi_blobs = blob_conv[-self.num_roi_levels:]
        # the RoI operators are fp32 (as the reference's); under autocast the pyramid is handed over in fp32
        roi_blobs = [b.float() for b in roi_blobs]
        ret = {}
        if not self.training:
            with torch.no_grad():
                if self.static_inference:
                    # always RPN_POST_NMS_TOP_N rows; the rows that are no proposals get image index -1 (the RoI operators
                    # pool zeros for them) and are reported in `rois_valid`: no host synchronisation, capturable
                    # the box head's pooling call is known before its RoIs are: the proposal stage's last launch writes
                    # the RoIAlign records of the blob it emits, and the pooling starts with its gather kernel
                    fr = cfg.FAST_RCNN
                    records = None
                    if fr.ROI_XFORM_METHOD == "RoIAlign" and cfg.FPN.MULTILEVEL_ROIS and PRODUCER_RECORDS:
                        post = int(cfg.TEST.RPN_POST_NMS_TOP_N * cfg.FPN.RPN_COLLECT_SCALE + 0.5)
                        records = roi_align_mod.PreparedRecords(roi_blobs, self.roi_scales, fr.ROI_XFORM_RESOLUTION,
                                                                fr.ROI_XFORM_RESOLUTION, fr.ROI_XFORM_SAMPLING_RATIO, post)
                    rois, valid, lvls = self.proposals(rpn_ret, im_info_d, static=True, with_levels=True, records=records)
                    blobs = {"rois": rois, "rois_levels": lvls, "rois_records": records}
                    ret["rois_valid"] = valid
                else:
                    rois = self.proposals(rpn_ret, im_info_d, static=False)
                    blobs = fpn_proposals.distribute(rois, cfg.FPN.ROI_MIN_LEVEL, cfg.FPN.ROI_MAX_LEVEL)
                    blobs["rois_levels"] = blobs["roi_levels"]
            self._mark("proposals")
            cls_score, bbox_pred = self.Box_Outs(self.Box_Head(roi_blobs, blobs))
            self._mark("box_head")
            ret.update(blob_conv=roi_blobs, rois=blobs["rois"], cls_score=cls_score, bbox_pred=bbox_pred)
            return ret
        with torch.no_grad():
            rois, valid = self.proposals(rpn_ret, im_info_d, static=True)
            if priority is None:
                priority = torch.rand(roidb["gt_boxes"].size(0) + rois.size(0), device=device)
            blobs = targets.label_proposals(cfg, rois, roidb["gt_boxes"], roidb["gt_classes"], roidb["gt_image"],
                                            im_info_d[:, 2], priority, n_img, self.iou_fn, roi_valid=valid,
                                            gt_mask_boxes=roidb.get("gt_mask_boxes"),
                                            gt_keypoints=roidb.get("gt_keypoints"),
                                            gt_polygons=roidb.get("gt_polygons"), rasterize_fn=self.rasterize_fn)
        self._mark("proposals_labelling")
        box_feat = self.Box_Head(roi_blobs, blobs)
        cls_score, bbox_pred = self.Box_Outs(box_feat)
        losses, metrics = {}, {}
        loss_rpn_cls, loss_rpn_bbox = fpn_mod.fpn_rpn_losses(cfg, rpn_ret, rpn_targets)
        for i, lvl in enumerate(range(cfg.FPN.RPN_MIN_LEVEL, cfg.FPN.RPN_MAX_LEVEL + 1)):
            losses["loss_rpn_cls_fpn%d" % lvl] = loss_rpn_cls[i]
            losses["loss_rpn_bbox_fpn%d" % lvl] = loss_rpn_bbox[i]
        loss_cls, loss_bbox, accuracy = heads.fast_rcnn_losses(
            cls_score.float(), bbox_pred.float(), blobs["labels_int32"], blobs["bbox_targets"],
            blobs["bbox_inside_weights"], blobs["bbox_outside_weights"])
        losses["loss_cls"], losses["loss_bbox"] = loss_cls, loss_bbox
        metrics["accuracy_cls"] = accuracy
        self._mark("box_head_losses")
        if cfg.MODEL.MASK_ON:
            mask_pred = self.Mask_Outs(self.Mask_Head(roi_blobs, blobs))
            losses["loss_mask"] = heads.mask_rcnn_losses_compact(mask_pred.float(), blobs["masks_int32"],
                                                                 blobs["mask_class"], cfg.MRCNN.WEIGHT_LOSS_MASK) \
                if cfg.MRCNN.CLS_SPECIFIC_MASK else \
                heads.mask_rcnn_losses(mask_pred.float(), blobs["masks_int32"], cfg.MRCNN.WEIGHT_LOSS_MASK)
            self._mark("mask_head_loss")
        if cfg.MODEL.KEYPOINTS_ON:
            kps_pred = self.Keypoint_Outs(self.Keypoint_Head(roi_blobs, blobs))
            norm = None if cfg.KRCNN.NORMALIZE_BY_VISIBLE_KEYPOINTS else blobs["keypoint_loss_normalizer"]
            losses["loss_kps"] = heads.keypoint_losses(kps_pred.float(), blobs["keypoint_locations_int32"],
                                                       blobs["keypoint_weights"], cfg.KRCNN.HEATMAP_SIZE,
                                                       cfg.KRCNN.LOSS_WEIGHT, norm)
            self._mark("keypoint_head_loss")
        ret["losses"], ret["metrics"] = losses, metrics
        ret["blobs"], ret["collected_rois"], ret["collected_valid"] = blobs, rois, valid
        return ret

    # ---- inference entry points (model_builder.py:326-348) -------------------------------------------------------------
    @torch.no_grad()
    def convbody_net(self, data):
        assert not self.training
        return [b.float() for b in self.Conv_Body(data)[-self.num_roi_levels:]]

    @torch.no_grad()
    def mask_net(self, blob_conv, rpn_blob):
        assert not self.training
        return self.Mask_Outs(self.Mask_Head(blob_conv, rpn_blob))

    @torch.no_grad()
    def keypoint_net(self, blob_conv, rpn_blob):
        assert not self.training
        return self.Keypoint_Outs(self.Keypoint_Head(blob_conv, rpn_blob))


def total_loss(ret):
    """The scalar the reference back-propagates: the plain sum of all losses (tools/train_net_step.py:425-429)."""
    return sum(ret["losses"].values())
