"""The R-CNN graph against the REFERENCE'S OWN model code, executed on the CPU from /root/reference
(oracle/ref_model.py; skipped where the reference is absent, i.e. on the GPU box).

What is pinned here:
  * a build under torch.manual_seed(RNG_SEED) draws bit-identical initial weights under identical parameter names, with
    the same set of trainable parameters ("reference initialisers, seed 3");
  * the synthetic data layer's RPN target blobs equal roi_data/rpn.py's under the same seed;
  * the WIRING of the training forward -- proposals -> collect -> labelling / sampling / targets -> RoI heads -> losses --
    and of the backward equals Generalized_RCNN._forward's, with the HIP operators replaced on both sides by CPU
    implementations of the same arithmetic (tests/cpu_backend.py: the oracle; oracle/ref_model.py: the reference's own
    kernels built for the host).  The operators themselves are pinned on the GPU (test_ops_gpu.py, test_e2e_gpu.py);
  * the inference forward and the box decoding of im_detect_bbox.
"""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_model  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_model.available(), reason="needs /root/reference and oracle/_ref")

from scenarios import H, W, NUM_GT, polygons_for_boxes, scenario  # noqa: E402


@pytest.fixture(scope="module")
def ref_cfg():
    warnings.filterwarnings("ignore")
    return ref_model.configure("configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml",
                               MODEL__LOAD_IMAGENET_PRETRAINED_WEIGHTS=False, MODEL__NUM_CLASSES=81)


@pytest.fixture(scope="module")
def models(ref_cfg):
    from detectron_pytorch_amd.rcnn import config, model

    ref = ref_model.build_model(seed=3)
    cfg = config.mask_rcnn_r50_fpn()
    torch.manual_seed(cfg.RNG_SEED)
    mine = model.GeneralizedRCNN(cfg)
    return ref, mine, cfg


def rect_rasterizer(polygons, box, m):
    from detectron_pytorch_amd.rcnn import targets

    p = np.array(polygons[0], dtype=np.float32)
    mb = torch.tensor([[p[0::2].min(), p[1::2].min(), p[0::2].max(), p[1::2].max()]])
    roi = torch.from_numpy(np.asarray(box, dtype=np.float32)).view(1, 4)
    return targets.rasterize_boxes(mb, roi, m).view(m, m).numpy().astype(np.float32)


def test_yaml_merge_equals_the_builtin_config(ref_cfg):
    from detectron_pytorch_amd.rcnn import config

    c = config.default_config().merge_from_file(
        os.path.join(ref_model.REFERENCE, "configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml"))
    b = config.mask_rcnn_r50_fpn()
    for sec in ("MODEL", "FPN", "FAST_RCNN", "MRCNN", "TRAIN", "TEST", "RPN"):
        for k, v in b[sec].items():
            assert c[sec][k] == v, (sec, k, c[sec][k], v)
            if sec in ref_cfg and k in ref_cfg[sec] and not isinstance(v, dict):
                rv = ref_cfg[sec][k]
                assert (tuple(rv) if isinstance(rv, (list, tuple)) else rv) == v, (sec, k, rv, v)


def test_keypoint_yaml_merge_equals_the_builtin_config():
    """config.keypoint_rcnn_r50_fpn() (what the GPU keypoint parity test builds, the reference's yaml is not on the GPU
    box) against the reference's own yaml merged on the defaults."""
    from detectron_pytorch_amd.rcnn import config

    c = config.infer(config.default_config().merge_from_file(
        os.path.join(ref_model.REFERENCE, "configs/baselines/e2e_keypoint_rcnn_R-50-FPN_1x.yaml")))
    b = config.keypoint_rcnn_r50_fpn()
    for sec in ("MODEL", "FPN", "FAST_RCNN", "KRCNN", "MRCNN", "TRAIN", "TEST", "RPN"):
        for k, v in b[sec].items():
            assert c[sec][k] == v, (sec, k, c[sec][k], v)


def test_seeded_initial_weights_equal_the_reference(models):
    ref, mine, _ = models
    a, b = ref.state_dict(), mine.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    trainable = lambda m: {k for k, p in m.named_parameters() if p.requires_grad}  # noqa: E731
    assert trainable(ref) == trainable(mine)
    assert sum(p.numel() for p in mine.parameters() if p.requires_grad) == 44125173     # ~44.1 M


def test_rpn_target_blobs_equal_the_reference_data_layer(ref_cfg, models):
    from detectron_pytorch_amd.rcnn import data as rdata

    _, _, cfg = models
    boxes, classes, _ = scenario()
    entries = [ref_model.roidb_entry(H, W, b, c, 81) for b, c in zip(boxes, classes)]
    want = ref_model.rpn_blobs(entries, [1.0, 1.0], seed=11)
    mine_entries = [dict(height=H, width=W, boxes=b, gt_classes=c, is_crowd=np.zeros(len(c), bool))
                    for b, c in zip(boxes, classes)]
    from oracle import ref

    got = rdata.add_rpn_blobs(cfg, mine_entries, [1.0, 1.0], np.random.RandomState(11),
                              bbox_overlaps=ref._mod("cython_bbox").bbox_overlaps)
    assert np.array_equal(got["im_info"], want["im_info"])
    keys = [k for k in want if k.startswith("rpn_")]
    assert len(keys) == 20
    for k in keys:
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
    # the in-repo IoU restatement the benchmark's data layer uses gives the same blobs
    got2 = rdata.add_rpn_blobs(cfg, mine_entries, [1.0, 1.0], np.random.RandomState(11))
    for k in keys:
        assert np.array_equal(got2[k], want[k]), k


GRAD_PARAMS = ["Box_Head.fc1.weight", "Box_Outs.bbox_pred.weight", "Mask_Head.conv_fcn.0.weight", "Mask_Outs.classify.bias",
               "RPN.FPN_RPN_conv.weight", "Conv_Body.posthoc_modules.3.weight", "Conv_Body.topdown_lateral_modules.0.conv_lateral.weight",
               "Conv_Body.conv_body.res5.2.conv3.weight", "Conv_Body.conv_body.res3.0.conv1.weight"]


@pytest.mark.parametrize("ground_truth", ["rectangles", "polygons"])
def test_training_forward_and_backward_equal_the_reference(ref_cfg, models, ground_truth):
    """ground_truth "rectangles": every instance's mask is its gt box; the reference's
    pycocotools rasteriser is replaced by the rectangle rasteriser of rcnn/targets.py.  "polygons": COCO-style polygon
    lists per instance; the reference runs its own polys_to_boxes / polys_to_mask_wrt_box with only pycocotools'
    frPyObjects / decode bound to the oracle's restatement of maskApi.c, this side rasterises through
    segms.polys_to_masks_wrt_boxes (here: its oracle stand-in, tests/cpu_backend.py; the HIP kernel is compared with the
    same oracle in tests/test_ops_gpu.py)."""
    import cpu_backend
    from detectron_pytorch_amd.rcnn import targets
    from detectron_pytorch_amd.segms import PackedPolygons

    ref, mine, cfg = models
    ref.train()
    mine.train()
    boxes, classes, data_np = scenario()
    polys = [polygons_for_boxes(b, seed=23 + i) for i, b in enumerate(boxes)] if ground_truth == "polygons" else [None, None]
    entries = [ref_model.roidb_entry(H, W, b, c, 81, segms=p) for b, c, p in zip(boxes, classes, polys)]
    blobs = ref_model.rpn_blobs(entries, [1.0, 1.0], seed=11)
    data = torch.from_numpy(data_np)
    g = 2 * NUM_GT
    priority = np.random.RandomState(7).permutation(g + 2000).astype(np.float32)
    ref.zero_grad()
    ret_ref, cap = ref_model.train_forward(ref, data, blobs, priority, rect_rasterizer if ground_truth == "rectangles" else None)
    sum(v.sum() for v in ret_ref["losses"].values()).backward()

    roidb = {"gt_boxes": torch.from_numpy(np.concatenate(boxes)), "gt_classes": torch.from_numpy(np.concatenate(classes)).long(),
             "gt_image": torch.tensor([0] * NUM_GT + [1] * NUM_GT)}
    if ground_truth == "polygons":
        roidb["gt_polygons"] = PackedPolygons.from_lists(polys[0] + polys[1])
    rpn_t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in blobs.items() if k.startswith("rpn_")}
    mine.zero_grad()
    with cpu_backend.cpu_ops(mine):
        # Proposals: the same rows.  Their ORDER is undefined among tied scores on both sides (np.argsort without a stable
        # kind at generate_proposals.py:131-138 and collect_and...py:85; fp32 sigmoid outputs of ~40 k anchors do collide),
        # and the sampling depends on the order, so the rest of the comparison runs on the reference's order.
        from detectron_pytorch_amd import fpn_proposals
        inner = fpn_proposals.generate_and_collect

        def collect_in_reference_order(*a, **k):
            rois, valid = inner(*a, **k)
            key = lambda x: x[np.lexsort(x.T[::-1])]  # noqa: E731
            assert np.array_equal(key(rois.numpy()), key(cap["rois"])), "collected proposals differ as a set"
            assert (rois.numpy() != cap["rois"]).any(1).mean() < 0.02, "more than tie-swaps differ"
            return torch.from_numpy(cap["rois"]), valid

        fpn_proposals.generate_and_collect = collect_in_reference_order
        ret = mine(data, torch.from_numpy(blobs["im_info"]), roidb=roidb, rpn_targets=rpn_t,
                   priority=torch.from_numpy(priority[:g + cap["rois"].shape[0]]))
        sum(ret["losses"].values()).backward()
    assert np.array_equal(ret["collected_rois"].numpy(), cap["rois"])
    # labelling: the reference's rows are the real rows of every image's block
    b, want = ret["blobs"], cap["blobs"]
    per, fper = cfg.TRAIN.BATCH_SIZE_PER_IM, int(round(cfg.TRAIN.FG_FRACTION * cfg.TRAIN.BATCH_SIZE_PER_IM))
    rows = torch.cat([i * per + torch.arange(int(n)) for i, n in enumerate(b["num_rois"])])
    assert rows.numel() == want["rois"].shape[0]
    assert np.array_equal(b["rois"][rows].numpy(), want["rois"])
    assert np.array_equal(b["labels_int32"][rows].numpy(), want["labels_int32"])
    assert (b["labels_int32"][rows] > 0).sum() >= g and (b["labels_int32"] == -1).sum() == 2 * per - rows.numel()
    np.testing.assert_allclose(b["bbox_targets"][rows].numpy(), want["bbox_targets"], rtol=0, atol=2e-6)
    assert np.array_equal(b["bbox_inside_weights"][rows].numpy(), want["bbox_inside_weights"])
    assert np.array_equal(b["bbox_outside_weights"][rows].numpy(), want["bbox_outside_weights"])
    frows = torch.cat([i * fper + torch.arange(int(n)) for i, n in enumerate(b["num_fg"])])
    assert np.array_equal(b["mask_rois"][frows].numpy(), want["mask_rois"])
    full = targets.expand_to_class_specific_mask_targets(b["masks_int32"], b["mask_class"], cfg.MODEL.NUM_CLASSES)
    assert np.array_equal(full[frows].numpy(), want["masks_int32"])
    assert np.array_equal(b["roi_has_mask_int32"][rows].numpy(), want["roi_has_mask_int32"])
    # FPN levels: the reference's per-level blobs are the rows of each level, in order
    for lvl in range(2, 6):
        sel = rows[b["rois_levels"][rows] == lvl]
        assert np.array_equal(b["rois"][sel].numpy(), want["rois_fpn%d" % lvl]), lvl
    assert len({int(x) for x in b["rois_levels"][rows]}) >= 2, "scenario should exercise several pyramid levels"
    # losses and metric
    for k, v in ret_ref["losses"].items():
        np.testing.assert_allclose(float(ret["losses"][k]), float(v), rtol=2e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(float(ret["metrics"]["accuracy_cls"]), float(ret_ref["metrics"]["accuracy_cls"]), rtol=1e-6)
    # gradients of the summed loss
    pr, pm = dict(ref.named_parameters()), dict(mine.named_parameters())
    for name in GRAD_PARAMS:
        a, bb = pr[name].grad, pm[name].grad
        assert a is not None and bb is not None, name
        err = (a - bb).abs().max().item() / max(a.abs().max().item(), 1e-12)
        assert err <= 2e-4, (name, err)


def test_polygon_helpers_equal_the_references_segms_module(ref_cfg):
    """The host-side polygon helpers against utils/segms.py imported from the reference: flip_segms (:33-60), and
    PackedPolygons + polys_to_boxes against polys_to_boxes (:121-132) on the packed form."""
    import utils.segms as ref_segms
    from detectron_pytorch_amd import segms
    from detectron_pytorch_amd import synthetic as syn

    polys, boxes, _ = syn.polygon_instances(9, seed=12)
    assert segms.flip_segms(polys, 800, 1333) == ref_segms.flip_segms(polys, 800, 1333)
    packed = segms.PackedPolygons.from_lists(polys)
    assert packed.num_instances == 9 and int(packed.poly_start[-1]) == packed.points.size(0)
    assert np.array_equal(segms.polys_to_boxes(packed).numpy(), ref_segms.polys_to_boxes(polys))
    flipped = segms.PackedPolygons.from_lists(segms.flip_segms(polys, 800, 1333))
    assert np.array_equal(segms.polys_to_boxes(flipped).numpy(), ref_segms.polys_to_boxes(ref_segms.flip_segms(polys, 800, 1333)))
    rect = segms.PackedPolygons.from_boxes(torch.from_numpy(boxes))
    assert np.array_equal(segms.polys_to_boxes(rect).numpy(), boxes)
    with pytest.raises(NotImplementedError):
        segms.flip_segms([{"counts": [1, 2], "size": [3, 1]}], 3, 1)


def test_optimizer_update_equals_the_reference(ref_cfg, models):
    """The UPDATE of a training iteration: the reference's own source text of the parameter-group construction
    (tools/train_net_step.py:262-307: weights with decay, biases at twice the learning rate without decay, frozen
    parameters left out) is executed on the reference model and its learning rate set with the reference's
    utils/net.py:update_learning_rate; rcnn.train.make_optimizer builds the groups here.  Both sides then take three SGD
    steps on identical (seeded) gradients; every trainable parameter must agree to fp32 rounding."""
    import copy
    import utils.net as net_utils
    from detectron_pytorch_amd.rcnn import train as rtrain

    ref, mine, cfg = models
    ref, mine = copy.deepcopy(ref), copy.deepcopy(mine)
    src = open(os.path.join(ref_model.REFERENCE, "tools", "train_net_step.py")).read().split("\n")
    text = "\n".join(l[4:] if l.startswith("    ") else l for l in src[261:307])      # ":262-307", de-indented
    assert text.lstrip().startswith("gn_param_nameset") and "torch.optim.SGD" in text
    ns = {"maskRCNN": ref, "cfg": ref_cfg, "nn": torch.nn, "torch": torch}
    exec(compile(text, "train_net_step.py:262-307", "exec"), ns)
    opt_ref = ns["optimizer"]
    # the learning rate of the bench's step: linear scaling rule to 2 images, first warm-up iteration
    lr = ref_cfg.SOLVER.BASE_LR * 2 / 16.0 * ref_cfg.SOLVER.WARM_UP_FACTOR
    net_utils.update_learning_rate(opt_ref, 0, lr)
    opt = rtrain.make_optimizer(mine, cfg, lr=lr)
    assert [len(g["params"]) for g in opt.param_groups] == [len(g["params"]) for g in opt_ref.param_groups[:2]]
    assert len(opt_ref.param_groups[2]["params"]) == 0                                   # no GroupNorm in this model
    for a, b in zip(opt.param_groups, opt_ref.param_groups):
        assert a["lr"] == b["lr"] and a["weight_decay"] == b["weight_decay"] and a["momentum"] == b["momentum"]
    assert opt.param_groups[1]["lr"] == 2 * opt.param_groups[0]["lr"] and opt.param_groups[1]["weight_decay"] == 0
    pr, pm = dict(ref.named_parameters()), dict(mine.named_parameters())
    assert list(pr) == list(pm)
    for step in range(3):
        gen = torch.Generator().manual_seed(100 + step)
        for name in pr:
            if pr[name].requires_grad:
                g = torch.randn(pr[name].shape, generator=gen) * 0.1
                pr[name].grad, pm[name].grad = g.clone(), g.clone()
        opt_ref.step()
        opt.step()
    worst = 0.0
    for name in pr:
        a, b = pr[name].detach(), pm[name].detach()
        assert a.requires_grad == b.requires_grad
        worst = max(worst, ((a - b).abs().max() / a.abs().max().clamp_min(1e-12)).item())
    assert worst <= 1e-6, worst


def test_inference_forward_and_box_decoding_equal_the_reference(ref_cfg, models):
    import cpu_backend
    from detectron_pytorch_amd.rcnn import inference

    ref, mine, cfg = models
    ref.eval()
    mine.eval()
    _, _, data_np = scenario(seed=9)
    data = torch.from_numpy(data_np[:1])
    im_info = torch.tensor([[float(H), float(W), 1.0]])
    with torch.no_grad():
        want = ref(data, im_info)
    with cpu_backend.cpu_ops(mine):
        got = mine(data, im_info)
    assert np.array_equal(got["rois"].numpy(), want["rois"])
    assert want["rois"].shape[0] > 100
    np.testing.assert_allclose(got["cls_score"].numpy(), want["cls_score"].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got["bbox_pred"].numpy(), want["bbox_pred"].numpy(), rtol=1e-5, atol=1e-7)
    # core/test.py:155-180: decode + clip, the reference's numpy functions against the device-side restatement
    import utils.boxes as box_utils

    boxes = want["rois"][:, 1:5]
    deltas = want["bbox_pred"].numpy() * 30      # large enough that some boxes leave the image / hit the exp clip
    ref_boxes = box_utils.clip_tiled_boxes(box_utils.bbox_transform(boxes, deltas, ref_cfg.MODEL.BBOX_REG_WEIGHTS), (H, W))
    mine_boxes = inference.clip_tiled_boxes(inference.bbox_transform(torch.from_numpy(boxes), torch.from_numpy(deltas),
                                                                     cfg.MODEL.BBOX_REG_WEIGHTS, cfg.BBOX_XFORM_CLIP), H, W)
    assert ref_boxes.dtype == np.float32
    np.testing.assert_allclose(mine_boxes.numpy(), ref_boxes, rtol=0, atol=1e-4)
    assert (np.abs(mine_boxes.numpy() - ref_boxes) > 0).mean() < 0.01   # numpy's fp64 exp vs torch's: last-bit cases only


def test_keypoint_branch_equals_the_reference():
    """tests/ref_keypoint_check.py in a process of its own (the reference's global cfg is configured differently there):
    e2e_keypoint_rcnn_R-50-FPN_1x.yaml -- seeded weights, Detectron names, keypoint RoI sampling, heat-map targets, loss,
    gradients."""
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_keypoint_check.py")], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "KEYPOINT_PARITY_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_softmax_rpn_activation_equals_the_reference():
    """tests/ref_softmax_rpn_check.py in a process of its own: RPN.CLS_ACTIVATION = 'softmax' (2 x A objectness channels,
    softmax over the pair, cross-entropy with ignore_index): seeded weights, Detectron names, collected proposals, losses,
    RPN gradients."""
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_softmax_rpn_check.py")], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "SOFTMAX_RPN_PARITY_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("which", ["x101", "faster"])
def test_builtin_configs_3_and_5_equal_the_reference(which):
    """tests/ref_config_check.py in a process of its own per configuration: BASELINE.json config 5
    (e2e_mask_rcnn_X-101-64x4d-FPN_1x.yaml + the keypoint head: 33 grouped 3x3 convolutions with groups = 64, the stride on the
    3x3) and config 3 (e2e_faster_rcnn_R-50-FPN_1x.yaml) -- yaml merge == built-in configuration, parameter names / shapes /
    seeded weights / trainable set / Detectron names == the reference's Generalized_RCNN, body + FPN outputs and the inference
    forward equal on a small image."""
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_config_check.py"), which], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "CONFIG_PARITY_OK " + which in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_reference_roi_feature_transform_runs_through_the_dropin_overlay():
    """tests/ref_overlay_check.py in a process of its own: the reference's REAL modeling/model_builder.py imported with
    detectron_pytorch_amd/dropin/lib in front of the reference's lib/ on sys.path; its roi_feature_transform (FPN branch with an
    empty level, and the single-level branch) is executed through the overlay's RoIAlignFunction (the HIP launch bound to the
    oracle, there is no GPU here), forward and backward, against the oracle and against roi_xform.roi_feature_transform."""
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_overlay_check.py")], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "OVERLAY_ROI_FEATURE_TRANSFORM_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
