"""Weights in and out: Detectron (Caffe2) `.pkl` blobs, and the reference's own checkpoint dict (SURVEY.md section 8f
row 4).

Reference: every module of lib/modeling carries a hand-written `detectron_weight_mapping()` (ResNet.py:80-103,353-389;
FPN.py:172-225,364-374; fast_rcnn_heads.py:29-37,94-101; mask_rcnn_heads.py:50-62,168-181;
keypoint_rcnn_heads.py:59-79,161-168) that Generalized_RCNN merges (model_builder.py:350-365);
`utils/detectron_weight_helper.py:9-21` copies the pickled blobs through it, `utils/net.py:156-163` filters a checkpoint
through it, `tools/train_net_step.py:118-135` writes {'step', 'train_size', 'batch_size', 'model', 'optimizer'}.

Here the mapping is ONE table of naming rules applied to the parameter names (which are the reference's), instead of a
method per module; tests/test_weights_cpu.py checks the resulting dict against the one the reference's own code builds.
"""
import pickle
import re

import numpy as np
import torch

from . import fpn as fpn_mod

_BRANCH = {"1": "a", "2": "b", "3": "c"}


def _body_rule(rest):
    """Names under Conv_Body.conv_body (ResNet.py:80-103 stem, :353-389 stages)."""
    if rest == "res1.conv1.weight":
        return "conv1_w"
    m = re.fullmatch(r"res1\.bn1\.(weight|bias)", rest)
    if m:
        return "res_conv1_bn_" + ("s" if m.group(1) == "weight" else "b")
    m = re.fullmatch(r"res(\d+)\.(\d+)\.(conv|bn)([123])\.(weight|bias)", rest)
    if m:
        stage, blk, kind, i, what = m.groups()
        prefix = "res%s_%s_branch2%s" % (stage, blk, _BRANCH[i])
        if kind == "conv":
            return prefix + "_w"
        return prefix + "_bn_" + ("s" if what == "weight" else "b")
    m = re.fullmatch(r"res(\d+)\.(\d+)\.downsample\.([01])\.(weight|bias)", rest)
    if m:
        stage, blk, idx, what = m.groups()
        prefix = "res%s_%s_branch1" % (stage, blk)
        if idx == "0":
            return prefix + "_w"
        return prefix + "_bn_" + ("s" if what == "weight" else "b")
    raise KeyError(rest)


def detectron_weight_mapping(model):
    """(mapping, orphans): parameter / buffer name of `model.state_dict()` -> Detectron blob name (None: neither loaded nor
    saved, e.g. the fixed bilinear up-sampling filters), and the Detectron blobs this graph has no parameter for."""
    cfg = model.cfg
    blobs = model.Conv_Body.fpn_level_info.blobs
    k_min = cfg.FPN.RPN_MIN_LEVEL
    wb = {"weight": "_w", "bias": "_b"}
    mapping = {}
    for name in model.state_dict():
        top, rest = name.split(".", 1)
        if top == "Conv_Body":
            if rest.startswith("conv_body."):
                mapping[name] = _body_rule(rest[len("conv_body."):])
                continue
            m = re.fullmatch(r"conv_top\.(weight|bias)", rest)
            if m:
                mapping[name] = "fpn_inner_" + blobs[0] + wb[m.group(1)]
                continue
            m = re.fullmatch(r"topdown_lateral_modules\.(\d+)\.conv_lateral\.(weight|bias)", rest)
            if m:
                mapping[name] = "fpn_inner_%s_lateral%s" % (blobs[int(m.group(1)) + 1], wb[m.group(2)])
                continue
            m = re.fullmatch(r"posthoc_modules\.(\d+)\.(weight|bias)", rest)
            if m:
                mapping[name] = "fpn_%s%s" % (blobs[int(m.group(1))], wb[m.group(2)])
                continue
        elif top == "RPN":
            m = re.fullmatch(r"FPN_RPN_(conv|cls_score|bbox_pred)\.(weight|bias)", rest)
            if m:
                base = {"conv": "conv_rpn_fpn%d", "cls_score": "rpn_cls_logits_fpn%d", "bbox_pred": "rpn_bbox_pred_fpn%d"}
                mapping[name] = base[m.group(1)] % k_min + wb[m.group(2)]
                continue
        elif top == "Box_Head":
            m = re.fullmatch(r"fc([12])\.(weight|bias)", rest)
            if m:
                mapping[name] = "fc%d%s" % (5 + int(m.group(1)), wb[m.group(2)])
                continue
        elif top == "Box_Outs":
            m = re.fullmatch(r"(cls_score|bbox_pred)\.(weight|bias)", rest)
            if m:
                mapping[name] = m.group(1) + wb[m.group(2)]
                continue
        elif top == "Mask_Head":
            m = re.fullmatch(r"conv_fcn\.(\d+)\.(weight|bias)", rest)
            if m:
                mapping[name] = "_[mask]_fcn%d%s" % (int(m.group(1)) // 2 + 1, wb[m.group(2)])
                continue
            m = re.fullmatch(r"upconv\.(weight|bias)", rest)
            if m:
                mapping[name] = "conv5_mask" + wb[m.group(1)]
                continue
        elif top == "Mask_Outs":
            m = re.fullmatch(r"classify\.(weight|bias)", rest)
            if m:
                mapping[name] = "mask_fcn_logits" + wb[m.group(1)]
                continue
            if rest.startswith("upsample.upconv."):
                mapping[name] = None
                continue
        elif top == "Keypoint_Head":
            m = re.fullmatch(r"conv_fcn\.(\d+)\.(weight|bias)", rest)
            if m:
                mapping[name] = "conv_fcn%d%s" % (int(m.group(1)) // 2 + 1, wb[m.group(2)])
                continue
        elif top == "Keypoint_Outs":
            m = re.fullmatch(r"deconv\.(weight|bias)", rest)
            if m:
                mapping[name] = "kps_deconv" + wb[m.group(1)]
                continue
            m = re.fullmatch(r"classify\.(weight|bias)", rest)
            if m:
                mapping[name] = ("kps_score_lowres" if cfg.KRCNN.UP_SCALE > 1 else "kps_score") + wb[m.group(1)]
                continue
            if rest.startswith("upsample.upconv."):
                mapping[name] = None
                continue
        raise KeyError("no Detectron name rule for parameter %r" % name)
    # blobs of the Detectron file without a counterpart: the stem's conv bias, the ImageNet classifier, and the (zero)
    # biases Caffe2 keeps for convolutions that are followed by an affine layer (ResNet.py:94,370,379)
    orphans = ["conv1_b", "fc1000_w", "fc1000_b"]
    for v in mapping.values():
        if v is not None and re.fullmatch(r"res\d+_\d+_branch(1|2[abc])_w", v):
            orphans.append(v[:-2] + "_b")
    return mapping, orphans


def load_detectron_weight(model, source):
    """utils/detectron_weight_helper.py:9-21.  `source`: path of a Detectron `.pkl` (latin1 pickle, optionally wrapped in
    {'blobs': ...}) or the blob dict itself.  Every parameter with a Detectron name is overwritten."""
    if isinstance(source, (str, bytes)):
        with open(source, "rb") as fp:
            source = pickle.load(fp, encoding="latin1")
    if "blobs" in source:
        source = source["blobs"]
    mapping, _ = detectron_weight_mapping(model)
    with torch.no_grad():
        for name, tensor in model.state_dict().items():
            d_name = mapping[name]
            if isinstance(d_name, str):
                tensor.copy_(torch.from_numpy(np.asarray(source[d_name], dtype=np.float32)).view_as(tensor))


def to_detectron_blobs(model):
    """The inverse: {Detectron blob name: float32 array} of every mapped parameter (what `load_detectron_weight` reads)."""
    mapping, _ = detectron_weight_mapping(model)
    return {mapping[k]: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items() if mapping[k] is not None}


def load_ckpt(model, ckpt_model_state):
    """utils/net.py:156-163: load checkpoint['model'], skipping the entries the mapping marks None (strict=False)."""
    mapping, _ = detectron_weight_mapping(model)
    state = {k: v for k, v in ckpt_model_state.items() if mapping.get(k)}
    return model.load_state_dict(state, strict=False)


def checkpoint_dict(step, train_size, batch_size, model, optimizer):
    """tools/train_net_step.py:118-135: the dict the reference `torch.save`s as ckpt/model_step<step>.pth."""
    return {"step": step, "train_size": train_size, "batch_size": batch_size, "model": model.state_dict(),
            "optimizer": optimizer.state_dict()}


def save_ckpt(path, step, train_size, batch_size, model, optimizer):
    torch.save(checkpoint_dict(step, train_size, batch_size, model, optimizer), path)
    return path


assert fpn_mod.LEVEL_INFO  # the blob names of the pyramid come from the level table of the FPN module
