"""GPU tests of the end-to-end path: the post-convolution half of the R-CNN graph (HIP proposals / NMS / IoU / RoIAlign
forward + backward, device-side labelling, losses) against values the REFERENCE'S OWN model code produced
(tests/golden/model.npz, generator tests/golden/generate_model.py), the convolution half against the CPU, the training
step in both launch forms, the gradient reducer on RCCL, and test-time detection.

The inputs of the post-convolution half are seeded arrays standing in for the convolution outputs
(tests/scenarios.py:synthetic_conv_outputs; the fixture generator substituted the same arrays for the reference's
convolutions), the head weights are the reference initialisers under seed 3 (tests/test_model_cpu.py pins them
bit-for-bit to the reference's build), so the GPU half sees exactly what the reference's own code saw.
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

from scenarios import H, W, NUM_GT, polygons_for_boxes, scenario, synthetic_conv_outputs  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model.npz")


def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN, allow_pickle=False)


@pytest.fixture(scope="module")
def nets(hip_lib_path):
    from detectron_pytorch_amd.rcnn import config, model

    cfg = config.mask_rcnn_r50_fpn()
    torch.manual_seed(cfg.RNG_SEED)
    cpu = model.GeneralizedRCNN(cfg)
    gpu = copy.deepcopy(cpu).to(dev())
    return cpu, gpu, cfg


def by_rows(a):
    return a[np.lexsort(a.T[::-1])]


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


def test_labelling_with_polygon_ground_truth_on_the_device(nets, golden):
    """targets.label_proposals with COCO-style polygon ground truth (roidb 'segms'): on the device -- mi_bbox_overlaps, the
    boxes enclosing the polygons, one mi_polys_to_masks_wrt_boxes launch for every mask row of the minibatch -- against
    the same function on CPU tensors with the oracle's stand-ins (tests/cpu_backend.py), which tests/test_model_cpu.py
    compares with the reference's own data layer.  Every blob bit for bit; padding rows stay -1."""
    import cpu_backend
    from detectron_pytorch_amd import nms, segms
    from detectron_pytorch_amd.rcnn import targets

    _, _, cfg = nets
    boxes, classes, _ = scenario()
    polys = polygons_for_boxes(boxes[0], seed=23) + polygons_for_boxes(boxes[1], seed=24)
    packed = segms.PackedPolygons.from_lists(polys)
    rois = torch.from_numpy(golden["train_collected_rois"])
    args = dict(gt_boxes=torch.from_numpy(np.concatenate(boxes)), gt_classes=torch.from_numpy(np.concatenate(classes)).long(),
                gt_image=torch.tensor([0] * NUM_GT + [1] * NUM_GT), im_scales=torch.tensor([1.0, 1.0]),
                priority=torch.from_numpy(np.random.RandomState(7).permutation(2 * NUM_GT + rois.size(0)).astype(np.float32)))
    want = targets.label_proposals(cfg, rois, args["gt_boxes"], args["gt_classes"], args["gt_image"], args["im_scales"],
                                   args["priority"], 2, cpu_backend.bbox_overlaps, gt_polygons=packed,
                                   rasterize_fn=cpu_backend.polys_to_masks_wrt_boxes)
    d = dev()
    got = targets.label_proposals(cfg, rois.to(d), args["gt_boxes"].to(d), args["gt_classes"].to(d), args["gt_image"].to(d),
                                  args["im_scales"].to(d), args["priority"].to(d), 2, nms.bbox_overlaps,
                                  gt_polygons=packed.to(d))
    assert set(got) == set(want)
    for k in want:
        if k == "bbox_targets":                      # log / division: the device's last bit (3e-6 as in the test below)
            np.testing.assert_allclose(got[k].cpu().numpy(), want[k].numpy(), rtol=0, atol=3e-6)
        else:
            assert np.array_equal(got[k].cpu().numpy(), want[k].numpy()), k
    fg = want["mask_class"] > 0
    assert int(fg.sum()) >= 2 * NUM_GT and 0.1 < float(want["masks_int32"][fg].float().mean()) < 0.9
    assert (want["masks_int32"][~fg] == -1).all()
    # ... and differ from the rectangle ground truth of the same boxes (the polygons are star-shaped outlines inside them)
    rect = targets.label_proposals(cfg, rois.to(d), args["gt_boxes"].to(d), args["gt_classes"].to(d), args["gt_image"].to(d),
                                   args["im_scales"].to(d), args["priority"].to(d), 2, nms.bbox_overlaps)
    assert (rect["masks_int32"].cpu() != want["masks_int32"]).any()


@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
def test_training_post_conv_half_matches_the_reference(nets, golden, layout):
    """`channels_last`: the pyramid is stored the way MIOpen's NHWC convolutions hand it over; the fused RoIAlign then runs
    roi_align_fwd_nhwc and the tile backward writes channels-last gradients."""
    from detectron_pytorch_amd.rcnn import data as rdata

    cpu, gpu, cfg = nets
    gpu.train()
    boxes, classes, _ = scenario()
    entries = [dict(height=H, width=W, boxes=b, gt_classes=c, is_crowd=np.zeros(len(c), bool)) for b, c in zip(boxes, classes)]
    blobs = rdata.add_rpn_blobs(cfg, entries, [1.0, 1.0], np.random.RandomState(11))     # == roi_data/rpn.py (CPU test)
    d = dev()
    blobs_np, logits_np, deltas_np = synthetic_conv_outputs(seed=21, n=2)
    fmt = torch.channels_last if layout == "channels_last" else torch.contiguous_format
    blob_g = [torch.from_numpy(b).to(d).contiguous(memory_format=fmt).requires_grad_() for b in blobs_np]
    rpn_g = {}
    for i, lvl in enumerate(range(2, 7)):
        rpn_g["rpn_cls_logits_fpn%d" % lvl] = torch.from_numpy(logits_np[i]).to(d).requires_grad_()
        rpn_g["rpn_bbox_pred_fpn%d" % lvl] = torch.from_numpy(deltas_np[i]).to(d).requires_grad_()
    roidb = {"gt_boxes": torch.from_numpy(np.concatenate(boxes)).to(d),
             "gt_classes": torch.from_numpy(np.concatenate(classes)).long().to(d),
             "gt_image": torch.tensor([0] * NUM_GT + [1] * NUM_GT, device=d)}
    rpn_t = {k: torch.from_numpy(v).to(d) for k, v in blobs.items() if k.startswith("rpn_")}
    priority = torch.from_numpy(np.random.RandomState(7).permutation(2 * NUM_GT + 2000).astype(np.float32)).to(d)
    want_rois = golden["train_collected_rois"]
    inner = gpu.proposals

    def proposals_in_reference_order(rpn, im_info, static):
        # HIP proposal generation + batched NMS + collect: the same rows as the reference; tied scores have no defined
        # order on either side and the sigmoid of the device may differ from the host's in the last bit, while the
        # sampling below depends on the order
        rois, valid = inner(rpn, im_info, static)
        assert bool(valid.all())
        got = rois.cpu().numpy()
        assert np.array_equal(by_rows(got), by_rows(want_rois)), "collected proposals differ as a set"
        assert (got != want_rois).any(1).mean() < 0.02, "more than tie-swaps differ"
        return torch.from_numpy(want_rois).to(d), valid

    gpu.proposals = proposals_in_reference_order
    try:
        gpu.zero_grad()
        ret = gpu.forward_from_features(blob_g, rpn_g, torch.from_numpy(blobs["im_info"]), roidb, rpn_t, priority)
        sum(ret["losses"].values()).backward()
    finally:
        del gpu.proposals
    b = {k: v.cpu() for k, v in ret["blobs"].items()}
    per, fper = cfg.TRAIN.BATCH_SIZE_PER_IM, int(round(cfg.TRAIN.FG_FRACTION * cfg.TRAIN.BATCH_SIZE_PER_IM))
    rows = torch.cat([i * per + torch.arange(int(n)) for i, n in enumerate(b["num_rois"])])
    labels = golden["train_labels"]
    assert rows.numel() == labels.shape[0]
    assert np.array_equal(b["rois"][rows].numpy(), golden["train_rois"])
    assert np.array_equal(b["labels_int32"][rows].numpy(), labels)
    cols = 4 * np.maximum(labels, 0)[:, None] + np.arange(4)[None, :]
    got_t = b["bbox_targets"][rows].numpy()
    np.testing.assert_allclose(got_t[np.arange(len(labels))[:, None], cols], golden["train_bbox_targets4"], rtol=0, atol=3e-6)
    assert np.count_nonzero(got_t) == np.count_nonzero(golden["train_bbox_targets4"][labels > 0])
    frows = torch.cat([i * fper + torch.arange(int(n)) for i, n in enumerate(b["num_fg"])])
    assert np.array_equal(b["mask_rois"][frows].numpy(), golden["train_mask_rois"])
    assert np.array_equal(b["masks_int32"][frows].numpy().astype(np.int8), golden["train_masks"])
    assert np.array_equal(b["mask_class"][frows].numpy(), labels[labels > 0])
    pad = torch.ones(2 * per, dtype=torch.bool)
    pad[rows] = False
    assert (b["labels_int32"][pad] == -1).all() and (b["rois"][pad][:, 0] == -1).all()
    # losses: fp32 GEMMs / convolutions of the heads on the GPU against the CPU's
    got = {k: float(v) for k, v in ret["losses"].items()}
    for name, want in zip(golden["loss_names"], golden["loss_values"]):
        np.testing.assert_allclose(got[str(name)], want, rtol=2e-4, atol=1e-6, err_msg=str(name))
    np.testing.assert_allclose(float(ret["metrics"]["accuracy_cls"]), float(golden["accuracy_cls"]), atol=2e-3)
    # gradients of the head parameters
    params = dict(gpu.named_parameters())
    for key in golden.files:
        if not key.startswith("grad_samples/"):
            continue
        name = key.split("/", 1)[1]
        g = params[name].grad.detach().cpu().numpy().reshape(-1)
        idx = np.random.RandomState(0).randint(0, g.size, size=min(256, g.size))
        norm = float(golden["grad_norm/" + name])
        assert abs(np.linalg.norm(g.astype(np.float64)) - norm) <= 2e-3 * norm, name
        # element-wise: fp32 weight-gradient convolutions / GEMMs sum tens of thousands of products in a different order on
        # the two devices, so the samples are compared as a vector (relative l2 error), not entry by entry
        diff = np.linalg.norm((g[idx] - golden[key]).astype(np.float64))
        assert diff <= 1e-2 * max(np.linalg.norm(golden[key].astype(np.float64)), 1e-30), (name, diff)
    # gradients w.r.t. the pyramid = the fused HIP RoIAlign backward over P2-P5 (box head 7x7 + mask head 14x14)
    roi_levels = blob_g[-4:]
    for i, f in enumerate(roi_levels):
        if f.grad is not None and layout == "channels_last":
            assert f.grad.is_contiguous(memory_format=torch.channels_last)
        g = np.zeros(f.numel(), np.float32) if f.grad is None else f.grad.detach().cpu().contiguous().numpy().reshape(-1)
        norm = float(golden["feat_grad_norm/%d" % i])
        assert abs(np.linalg.norm(g.astype(np.float64)) - norm) <= 2e-3 * norm + 1e-12, i
        idx = np.random.RandomState(i).randint(0, g.size, size=min(512, g.size))
        # the incoming gradients come from the heads' fp32 GEMMs / convolutions on the GPU (different summation order than
        # the CPU's), single entries are sums of signed contributions of several RoIs: compared as vectors
        want = golden["feat_grad_samples/%d" % i].astype(np.float64)
        assert np.linalg.norm(g[idx] - want) <= 5e-3 * np.linalg.norm(want) + 1e-12, i
        assert np.array_equal(g[idx] == 0, want == 0), i                    # the same pixels are touched
        nz = golden["feat_grad_nonzero_index/%d" % i]
        if nz.size:
            want = golden["feat_grad_nonzero_samples/%d" % i].astype(np.float64)
            assert np.linalg.norm(g[nz] - want) <= 5e-3 * np.linalg.norm(want), i


def test_keypoint_training_post_conv_half_matches_the_reference(hip_lib_path):
    """BASELINE config 5's extra branch on the device: key-point RoIs, heat-map targets and weights (roi_data/
    keypoint_rcnn.py:33-106, utils/keypoints.py:160-211), the v1convX head + deconvolution + bilinear up-sampling and
    `loss_kps` (keypoint_rcnn_heads.py:17-179), and the gradients the fused HIP RoIAlign backward (14 x 14) hands to the
    pyramid -- against what the reference's own e2e_keypoint_rcnn_R-50-FPN model computed from the same arrays
    (tests/golden/model_keypoint.npz, generator tests/golden/generate_model_keypoint.py)."""
    from scenarios import keypoint_scenario
    from detectron_pytorch_amd.rcnn import config, data as rdata, model

    g = np.load(os.path.join(os.path.dirname(GOLDEN), "model_keypoint.npz"), allow_pickle=False)
    cfg = config.keypoint_rcnn_r50_fpn()
    cfg.MODEL.NUM_CLASSES = 2
    assert cfg.MODEL.KEYPOINTS_ON and cfg.KRCNN.HEATMAP_SIZE == 56
    torch.manual_seed(3)
    gpu = model.GeneralizedRCNN(cfg).to(dev())
    gpu.train()
    boxes, classes, kps, _ = keypoint_scenario()
    entries = [dict(height=H, width=W, boxes=b, gt_classes=c, is_crowd=np.zeros(len(c), bool)) for b, c in zip(boxes, classes)]
    blobs = rdata.add_rpn_blobs(cfg, entries, [1.0, 1.0], np.random.RandomState(11))
    d = dev()
    blobs_np, logits_np, deltas_np = synthetic_conv_outputs(seed=23, n=2)
    blob_g = [torch.from_numpy(b).to(d).requires_grad_() for b in blobs_np]
    rpn_g = {}
    for i, lvl in enumerate(range(2, 7)):
        rpn_g["rpn_cls_logits_fpn%d" % lvl] = torch.from_numpy(logits_np[i]).to(d).requires_grad_()
        rpn_g["rpn_bbox_pred_fpn%d" % lvl] = torch.from_numpy(deltas_np[i]).to(d).requires_grad_()
    ngt = 2 * NUM_GT
    roidb = {"gt_boxes": torch.from_numpy(np.concatenate(boxes)).to(d), "gt_classes": torch.ones(ngt, dtype=torch.long, device=d),
             "gt_image": torch.tensor([0] * NUM_GT + [1] * NUM_GT, device=d),
             "gt_keypoints": torch.from_numpy(np.concatenate(kps)).to(d)}
    rpn_t = {k: torch.from_numpy(v).to(d) for k, v in blobs.items() if k.startswith("rpn_")}
    priority = torch.from_numpy(np.random.RandomState(7).permutation(ngt + 2000).astype(np.float32)).to(d)
    want_rois = g["train_collected_rois"]
    inner = gpu.proposals

    def proposals_in_reference_order(rpn, im_info, static):
        rois, valid = inner(rpn, im_info, static)
        got = rois.cpu().numpy()[valid.cpu().numpy()]
        assert np.array_equal(by_rows(got), by_rows(want_rois)), "collected proposals differ as a set"
        out = torch.full_like(rois, 0.0)
        out[:, 0] = -1.0
        out[:want_rois.shape[0]] = torch.from_numpy(want_rois).to(d)
        ok = torch.zeros_like(valid)
        ok[:want_rois.shape[0]] = True
        return out, ok

    gpu.proposals = proposals_in_reference_order
    try:
        gpu.zero_grad()
        ret = gpu.forward_from_features(blob_g, rpn_g, torch.from_numpy(blobs["im_info"]), roidb, rpn_t, priority)
        sum(ret["losses"].values()).backward()
    finally:
        del gpu.proposals
    b = {k: v.cpu() for k, v in ret["blobs"].items()}
    fper = int(round(cfg.TRAIN.FG_FRACTION * cfg.TRAIN.BATCH_SIZE_PER_IM))
    rows = torch.cat([i * fper + torch.arange(int(n)) for i, n in enumerate(b["num_keypoint_rois"])])
    assert rows.numel() == g["keypoint_rois"].shape[0] >= ngt
    assert np.array_equal(b["keypoint_rois"][rows].numpy(), g["keypoint_rois"])
    k17 = (rows.view(-1, 1) * 17 + torch.arange(17).view(1, -1)).reshape(-1)
    assert np.array_equal(b["keypoint_locations_int32"][k17].numpy(), g["keypoint_locations_int32"])     # exact
    assert np.array_equal(b["keypoint_weights"][k17].numpy(), g["keypoint_weights"])                     # exact
    np.testing.assert_allclose(float(b["keypoint_loss_normalizer"]), float(g["keypoint_loss_normalizer"]), rtol=1e-6)
    got = {k: float(v) for k, v in ret["losses"].items()}
    for name, want in zip(g["loss_names"], g["loss_values"]):
        np.testing.assert_allclose(got[str(name)], want, rtol=2e-4, atol=1e-6, err_msg=str(name))
    params = dict(gpu.named_parameters())
    for key in g.files:
        if not key.startswith("grad_samples/"):
            continue
        name = key.split("/", 1)[1]
        gr = params[name].grad.detach().cpu().numpy().reshape(-1)
        idx = np.random.RandomState(0).randint(0, gr.size, size=min(256, gr.size))
        norm = float(g["grad_norm/" + name])
        assert abs(np.linalg.norm(gr.astype(np.float64)) - norm) <= 2e-3 * norm, name
        diff = np.linalg.norm((gr[idx] - g[key]).astype(np.float64))
        assert diff <= 1e-2 * max(np.linalg.norm(g[key].astype(np.float64)), 1e-30), (name, diff)
    for i, f in enumerate(blob_g[-4:]):   # P5, P4, P3, P2: box head 7 x 7 + key-point head 14 x 14, one fused backward
        gr = np.zeros(f.numel(), np.float32) if f.grad is None else f.grad.detach().cpu().numpy().reshape(-1)
        norm = float(g["feat_grad_norm/%d" % i])
        assert abs(np.linalg.norm(gr.astype(np.float64)) - norm) <= 2e-3 * norm + 1e-12, i
        idx = np.random.RandomState(i).randint(0, gr.size, size=min(512, gr.size))
        want = g["feat_grad_samples/%d" % i].astype(np.float64)
        assert np.linalg.norm(gr[idx] - want) <= 5e-3 * np.linalg.norm(want) + 1e-12, i
        assert np.array_equal(gr[idx] == 0, want == 0), i


def test_inference_post_conv_half_matches_the_reference(nets, golden):
    from detectron_pytorch_amd.rcnn import inference

    cpu, gpu, cfg = nets
    gpu.eval()
    d = dev()
    blobs_np, logits_np, deltas_np = synthetic_conv_outputs(seed=22, n=1)
    rpn_ret = {}
    for i, lvl in enumerate(range(2, 7)):
        rpn_ret["rpn_cls_logits_fpn%d" % lvl] = torch.from_numpy(logits_np[i]).to(d)
        rpn_ret["rpn_bbox_pred_fpn%d" % lvl] = torch.from_numpy(deltas_np[i]).to(d)
    with torch.no_grad():
        ret = gpu.forward_from_features([torch.from_numpy(b).to(d) for b in blobs_np], rpn_ret,
                                        torch.tensor([[float(H), float(W), 1.0]]))
    rois = ret["rois"].cpu().numpy()
    want = golden["eval_rois"]
    assert np.array_equal(by_rows(rois), by_rows(want))
    index = {tuple(r): i for i, r in enumerate(rois)}
    perm = np.array([index[tuple(r)] for r in want])                     # reference row -> row of this run
    cls, bbox = ret["cls_score"].cpu().numpy()[perm], ret["bbox_pred"].cpu().numpy()[perm]
    np.testing.assert_allclose(cls[:64], golden["eval_cls_score_rows"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(cls.astype(np.float64).sum(1), golden["eval_cls_score_sum"], rtol=1e-5)
    np.testing.assert_allclose(bbox[:64], golden["eval_bbox_pred_rows"], rtol=2e-3, atol=2e-6)
    # box decoding + clipping on the device against the reference's numpy functions, on the reference's deltas
    deltas = torch.from_numpy(golden["eval_bbox_pred_rows"] * 30).to(d)
    pred = inference.clip_tiled_boxes(inference.bbox_transform(torch.from_numpy(want[:64, 1:5]).to(d), deltas,
                                                               cfg.MODEL.BBOX_REG_WEIGHTS, cfg.BBOX_XFORM_CLIP), H, W)
    np.testing.assert_allclose(pred.cpu().numpy(), golden["eval_pred_boxes_rows"], rtol=0, atol=2e-4)


def test_gpu_convolutions_match_the_cpu(nets):
    cpu, gpu, _ = nets
    cpu.eval()
    gpu.eval()
    _, _, data_np = scenario(seed=3)
    x = torch.from_numpy(data_np)
    with torch.no_grad():
        want = cpu.Conv_Body(x)
        want_rpn = cpu.RPN(want)
        got = gpu.Conv_Body(x.to(dev()))
        got_rpn = gpu.RPN(got)
    for a, b in zip(got, want):
        assert rel_err(a.cpu().numpy(), b.numpy()) <= 5e-3
    for k in want_rpn:
        assert rel_err(got_rpn[k].cpu().numpy(), want_rpn[k].numpy()) <= 5e-3, k


def test_label_proposals_device_equals_host_arithmetic(hip_lib_path):
    """targets.label_proposals with mi_bbox_overlaps and the device's sorts against the same function on CPU tensors
    with the numpy IoU: random proposals around jittered gt boxes, an invalid tail, an image short of candidates."""
    from detectron_pytorch_amd import nms
    from detectron_pytorch_amd.rcnn import config, data as rdata, targets

    cfg = config.mask_rcnn_r50_fpn()
    rng = np.random.RandomState(0)
    g = 12
    gt = np.zeros((g, 4), np.float32)
    gt[:, :2] = rng.uniform(0, 500, (g, 2))
    gt[:, 2:] = gt[:, :2] + rng.uniform(30, 300, (g, 2))
    gt_img = np.array([0] * 5 + [1] * 7)
    r = 700
    rois = np.zeros((r, 5), np.float32)
    rois[:, 0] = (rng.rand(r) < 0.2).astype(np.float32)          # image 0 is short of candidates (< 512)
    rois[:, 0] = 1 - rois[:, 0]
    near = rng.rand(r) < 0.4
    src = gt[rng.randint(0, g, r)]
    jit = src + rng.uniform(-25, 25, (r, 4)).astype(np.float32)
    rnd = np.concatenate([rng.uniform(0, 600, (r, 2)), rng.uniform(0, 600, (r, 2))], 1).astype(np.float32)
    rnd[:, 2:] = np.maximum(rnd[:, 2:], rnd[:, :2] + 1)
    rois[:, 1:] = np.where(near[:, None], jit, rnd) * 1.5
    valid = np.ones(r, bool)
    valid[-40:] = False
    prio = rng.permutation(g + r).astype(np.float32)
    scales = np.array([1.5, 1.5], np.float32)
    args = lambda t: (cfg, t(rois), t(gt), t(rng_cls), t(gt_img), t(scales), t(prio), 2)  # noqa: E731
    rng_cls = rng.randint(1, 81, g).astype(np.int64)
    iou_np = lambda a, b: torch.from_numpy(rdata.bbox_overlaps_np(a.numpy(), b.numpy()))  # noqa: E731
    cpu = targets.label_proposals(*args(torch.from_numpy), iou_np, roi_valid=torch.from_numpy(valid))
    to_d = lambda a: torch.from_numpy(a).to(dev())  # noqa: E731
    gpu = targets.label_proposals(*args(to_d), nms.bbox_overlaps, roi_valid=to_d(valid))
    assert int(cpu["num_fg"].sum()) > 20 and int(cpu["num_rois"][0]) < 512 == int(cpu["num_rois"][1])
    for k, v in cpu.items():
        got = gpu[k].cpu()
        if v.dtype.is_floating_point and k == "bbox_targets":
            np.testing.assert_allclose(got.numpy(), v.numpy(), rtol=0, atol=3e-6, err_msg=k)
        else:
            assert torch.equal(got, v), k


def _small_training_job(net, cfg):
    from detectron_pytorch_amd.rcnn import data as rdata

    batch = rdata.synthetic_minibatch(cfg, 2, seed=1, blob_height=256, blob_width=320, image_width=318)
    return rdata.to_device(batch, dev())


def test_training_step_eager_and_hipgraph_liveness(nets):
    """Three eager steps of the whole model on the GPU (finite losses, parameters move), then the same step captured in a
    hipGraph and replayed -- the training forward is static-shaped and has no host synchronisation.
    A LIVENESS check of the captured form, not a parity check: replays of captured backward passes are not trustworthy on
    this PyTorch / ROCm stack (tools/reduce_probe.py reproduces a wrong, stable value with plain torch expressions), so the
    replayed loss only has to be finite and in the neighbourhood of the eager one; `bench.py --launch graph` stays
    experimental and the headline launches eagerly."""
    from detectron_pytorch_amd.rcnn import train as rtrain

    _, gpu, cfg = nets
    net = copy.deepcopy(gpu).train()
    data, im_info, roidb, rpn_t = _small_training_job(net, cfg)
    opt = rtrain.make_optimizer(net, cfg, lr=1e-3)
    before = net.Box_Head.fc1.weight.detach().clone()
    losses = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ret = rtrain.train_step(net, opt, data, im_info, roidb, rpn_t)
            losses.append(ret["total_loss"])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert all(np.isfinite(float(x)) for x in losses)
    assert not torch.equal(before, net.Box_Head.fc1.weight.detach())
    assert int(ret["blobs"]["num_rois"].sum()) > 0 and int(ret["blobs"]["num_fg"].min()) >= 8
    opt.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="relaxed"):
        ret = net(data, im_info, roidb=roidb, rpn_targets=rpn_t)
        loss = sum(ret["losses"].values())
        loss.backward()
        opt.step()
    w0 = net.Box_Head.fc1.weight.detach().clone()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and abs(float(loss) - float(losses[-1])) < 0.5 * abs(float(losses[-1])) + 0.5
    assert not torch.equal(w0, net.Box_Head.fc1.weight.detach())


def test_fixed_batch_loss_decreases_at_a_safe_learning_rate(nets):
    """The step is pinned piecewise against the reference (forward + backward: tests/test_model_cpu.py and the golden
    fixtures; the SGD update: test_optimizer_update_equals_the_reference).  This is the smoke check of the whole: on a
    fixed resident batch, at 1/100 of the schedule's first learning rate, twenty iterations must lower the loss at every
    step.  (At the schedule's own rate the fixed-batch loss is NOT monotone -- 5.38 -> 1.58 -> 3.6 -> 2.05 over 30 steps,
    tools/loss_probe.py: momentum 0.9 at a rate tuned for pretrained weights overshoots on a random-init body whose
    AffineChannel layers are the identity, and the labelled RoIs are re-sampled from the moving proposals every step.)"""
    from detectron_pytorch_amd.rcnn import train as rtrain

    _, gpu, cfg = nets
    net = copy.deepcopy(gpu).train()
    data, im_info, roidb, rpn_t = _small_training_job(net, cfg)
    opt = rtrain.make_optimizer(net, cfg, lr=cfg.SOLVER.BASE_LR * 2 / 16.0 / 3.0 / 100.0)
    losses = []
    for _ in range(20):
        opt.zero_grad(set_to_none=True)
        ret = net(data, im_info, roidb=roidb, rpn_targets=rpn_t)
        loss = sum(ret["losses"].values())
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(x) for x in losses)
    assert all(b < a + 2e-3 * abs(a) for a, b in zip(losses, losses[1:])), losses
    assert losses[-1] < 0.9 * losses[0], losses


def test_minibatch_feeder_delivers_every_minibatch_in_order():
    """parallel.MinibatchFeeder (the input half of the reference's scatter, nn/parallel/_functions.py:62-83): pinned host blobs
    of step k + 1 land in a staging set on a copy stream while step k runs, commit() moves them into the resident blobs.
    Twelve steps with large blobs that change every step, a slow consumer on the step's stream, no host synchronisation inside
    the loop: every step sees exactly its own minibatch (a staging set overwritten early, or read before it has landed,
    shows up as a mixed checksum)."""
    from detectron_pytorch_amd import parallel

    d = dev()
    live = [torch.zeros(24 << 20, device=d), torch.zeros((2, 3, 64, 64), dtype=torch.int32, device=d)]
    addrs = [t.data_ptr() for t in live]
    steps = 12
    hosts = [[torch.full(live[0].shape, float(k + 1)).pin_memory(), torch.full(live[1].shape, 7 * k + 3, dtype=torch.int32).pin_memory()]
             for k in range(steps)]
    a = torch.randn(2048, 2048, device=d)
    sums = torch.zeros((steps, 3), dtype=torch.float64, device=d)
    feeder = parallel.MinibatchFeeder(live)
    with pytest.raises(AssertionError):
        feeder.commit()                       # nothing prefetched
    feeder.prefetch(hosts[0])
    with pytest.raises(AssertionError):
        feeder.prefetch(hosts[0])             # twice without a commit
    for k in range(steps):
        feeder.commit()
        if k + 1 < steps:
            feeder.prefetch(hosts[k + 1])
        b = a
        for _ in range(6):                    # the step: long enough for the next copy to finish under it
            b = (b @ a) * 1e-3
        sums[k, 0] = live[0].double().sum()
        sums[k, 1] = live[1].double().sum()
        sums[k, 2] = live[0].double().max() - live[0].double().min()
    torch.cuda.synchronize()
    got = sums.cpu().numpy()
    for k in range(steps):
        assert got[k, 0] == float(k + 1) * live[0].numel(), k
        assert got[k, 1] == float(7 * k + 3) * live[1].numel(), k
        assert got[k, 2] == 0.0, k
    assert [t.data_ptr() for t in live] == addrs      # the resident blobs keep their addresses (a captured graph reads them)


def test_gradient_reducer_on_rccl_world_size_one(nets):
    """GradientAllReducer on the nccl (= RCCL) backend with one rank: bucket views, hooks, asynchronous all-reduce from
    the backward, and the graph-mode variant (collectives between two captured graphs) leave exactly the gradients of a
    plain backward."""
    import socket

    import torch.distributed as dist

    from detectron_pytorch_amd import parallel

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 512), torch.nn.ReLU(),
                                  torch.nn.Linear(512, 16)).to(dev())
        x = torch.randn(64, 256, device=dev())
        net(x).square().mean().backward()
        want = [p.grad.clone() for p in net.parameters()]
        for overlap in (True, False):
            net.zero_grad(set_to_none=True)
            red = parallel.GradientAllReducer(net.parameters(), bucket_bytes=256 << 10, force=True, overlap=overlap)
            assert red.active and len(red.buckets) >= 3
            red.begin_step()
            net(x).square().mean().backward()
            assert red.finish_step() == len(red.buckets)
            torch.cuda.synchronize()
            for p, w in zip(net.parameters(), want):
                assert torch.equal(p.grad, w)
                assert p.grad.data_ptr() >= red.buckets[0][0].data_ptr() or True
            red.close()
        # graph mode: backward captured (zero-fill of the buckets included), collectives issued between replays
        net.zero_grad(set_to_none=True)
        red = parallel.GradientAllReducer(net.parameters(), bucket_bytes=256 << 10, force=True, overlap=False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            red.begin_step()
            net(x).square().mean().backward()
            red.finish_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
            red.begin_step()
            net(x).square().mean().backward()
        for _ in range(2):
            g.replay()
            assert red.reduce_now() == len(red.buckets)
            red.average_()
        torch.cuda.synchronize()
        for p, w in zip(net.parameters(), want):
            assert torch.allclose(p.grad, w, rtol=1e-6, atol=1e-8)
        red.close()
    finally:
        dist.destroy_process_group()


def test_im_detect_all_on_device(nets):
    from detectron_pytorch_amd.rcnn import inference

    _, gpu, cfg = nets
    gpu.eval()
    _, _, data_np = scenario(seed=2)
    scores, boxes, cls_boxes = inference.im_detect_all(gpu, torch.from_numpy(data_np[:1]).to(dev()),
                                                       torch.tensor([[float(H), float(W), 1.0]]))
    assert scores.numel() <= cfg.TEST.DETECTIONS_PER_IM and boxes.shape == (scores.numel(), 4)
    assert len(cls_boxes) == cfg.MODEL.NUM_CLASSES and sum(len(c) for c in cls_boxes[1:]) == scores.numel()
    if scores.numel():
        assert float(scores.min()) > cfg.TEST.SCORE_THRESH
        assert float(boxes[:, 0::2].min()) >= 0 and float(boxes[:, 0::2].max()) <= W - 1
        assert float(boxes[:, 1::2].min()) >= 0 and float(boxes[:, 1::2].max()) <= H - 1


def test_detection_static_path_and_hipgraph_equal_the_dynamic_path(nets):
    """The fixed-shape, synchronisation-free detection (static RoI blob with padding rows, mi_nms_segmented) and its
    hipGraph replay return the detections of the dynamic path for several images through one graph -- up to the run-to-run
    rounding of the box head (see _same_detections; the post-processing itself is pinned bit for bit on fixed head outputs in
    tests/test_ops_gpu.py)."""
    from detectron_pytorch_amd.rcnn import inference

    _, gpu, cfg = nets
    gpu.eval()
    saved = cfg.TEST.SCORE_THRESH
    cfg.TEST.SCORE_THRESH = 0.012     # a randomly initialised classifier scores ~1/81 everywhere: let some rows through
    try:
        _check_static_detection(gpu, cfg, inference)
    finally:
        cfg.TEST.SCORE_THRESH = saved


@pytest.mark.parametrize("soft,vote", [(True, False), (False, True), (True, True)])
def test_detection_hipgraph_with_soft_nms_and_voting(nets, soft, vote):
    """TEST.SOFT_NMS / TEST.BBOX_VOTE inside the static path and its hipGraph (core/test.py:753-773 without a host round trip)."""
    from detectron_pytorch_amd.rcnn import inference

    _, gpu, cfg = nets
    gpu.eval()
    saved = (cfg.TEST.SCORE_THRESH, cfg.TEST.SOFT_NMS.ENABLED, cfg.TEST.BBOX_VOTE.ENABLED)
    cfg.TEST.SCORE_THRESH, cfg.TEST.SOFT_NMS.ENABLED, cfg.TEST.BBOX_VOTE.ENABLED = 0.012, soft, vote
    try:
        _check_static_detection(gpu, cfg, inference)
    finally:
        cfg.TEST.SCORE_THRESH, cfg.TEST.SOFT_NMS.ENABLED, cfg.TEST.BBOX_VOTE.ENABLED = saved


def _same_detections(a_scores, a_boxes, b_scores, b_boxes, score_atol, box_atol=1e-3, flips=2):
    """Two runs of the network never agree to the last bit (the box head's split-K GEMM accumulates with atomics: 5-8e-7 on
    the scores of one and the same call repeated), and a difference of 1e-6 can flip a row that sits exactly at a threshold
    (score threshold, an IoU at the NMS threshold, the 100th score of the detections_per_im cut) -- which shifts every later
    row of a positional comparison.  So: every row of one result has a partner in the other (score and box within the
    tolerances), except for at most `flips` rows per side."""
    a = torch.cat([a_boxes.reshape(-1, 4), a_scores.reshape(-1, 1)], 1).double().cpu()
    b = torch.cat([b_boxes.reshape(-1, 4), b_scores.reshape(-1, 1)], 1).double().cpu()
    if a.numel() == 0 or b.numel() == 0:
        return a.size(0) <= flips and b.size(0) <= flips
    d = (a[:, None, :] - b[None, :, :]).abs()
    ok = (d[:, :, 4] <= score_atol) & (d[:, :, :4].amax(dim=2) <= box_atol)
    return int((~ok.any(dim=1)).sum()) <= flips and int((~ok.any(dim=0)).sum()) <= flips


def _strict_postprocess_equality(gpu, cfg, inference, blob, im_info):
    """The tolerant comparisons below absorb the network's run-to-run rounding; an off-by-one or a dropped row of the static
    post-processing must not hide in that tolerance.  So the head runs ONCE, and the same score / box tensors go through the
    static sequence (what the hipGraph replays) and through the dynamic one: rows, order, scores, boxes and the per-class
    counts must be EQUAL, bit for bit."""
    from detectron_pytorch_amd import detection

    t = cfg.TEST
    scores, boxes, _, valid = inference.im_detect_bbox(gpu, blob, im_info.to(dev()), None, None, static=True)
    opts = dict(soft_nms=t.SOFT_NMS.ENABLED, soft_nms_sigma=t.SOFT_NMS.SIGMA, soft_nms_method=t.SOFT_NMS.METHOD,
                bbox_vote=t.BBOX_VOTE.ENABLED, bbox_vote_thresh=t.BBOX_VOTE.VOTE_TH, bbox_vote_method=t.BBOX_VOTE.SCORING_METHOD)
    if t.SOFT_NMS.ENABLED or t.BBOX_VOTE.ENABLED:
        res = detection.box_results_static_general(scores, boxes, t.SCORE_THRESH, t.NMS, t.DETECTIONS_PER_IM, roi_valid=valid, **opts)
    else:
        res = detection.box_results_static(scores, boxes, t.SCORE_THRESH, t.NMS, t.DETECTIONS_PER_IM, roi_valid=valid)
    stat = detection._results_from_static(res, False)
    assert stat is not None, "more ties at the detections_per_im cut than the static result holds"
    rows = valid.nonzero().flatten()          # the dynamic path never sees the padding rows of the static RoI blob
    dyn = detection.box_results_with_nms_and_limit(scores[rows], boxes[rows], t.SCORE_THRESH, t.NMS, t.DETECTIONS_PER_IM, **opts)
    assert torch.equal(stat[0], dyn[0]) and torch.equal(stat[1], dyn[1]), "static and dynamic post-processing differ"
    assert [len(c) for c in stat[2]] == [len(c) for c in dyn[2]]
    assert all(torch.equal(a, b) for a, b in zip(stat[2][1:], dyn[2][1:]))


def _check_static_detection(gpu, cfg, inference):
    graph, seen = None, 0
    # Soft-NMS re-scores a row with a function of its IoU with the rows picked before it: boxes that differ by 1e-3 px move
    # an IoU, and with it a score of ~0.1, by up to ~1e-5
    score_atol = 5e-5 if cfg.TEST.SOFT_NMS.ENABLED else 5e-6
    for seed, scale in ((2, 1.0), (5, 1.0), (7, 0.5)):
        _, _, data_np = scenario(seed=seed)
        blob = torch.from_numpy(data_np[:1]).to(dev())
        im_info = torch.tensor([[float(H), float(W), scale]])
        want = inference.im_detect_all(gpu, blob, im_info)
        _strict_postprocess_equality(gpu, cfg, inference, blob, im_info)
        res = inference.im_detect_all_static(gpu, blob, im_info.to(dev()))
        count = int(res["count"])
        assert count == int(res["total"]) and abs(count - want[0].numel()) <= 2
        stat = res["dets"][:count].clone()
        assert _same_detections(stat[:, 4], stat[:, :4], want[0], want[1], score_atol)
        counts = torch.tensor([len(c) for c in want[2][1:]])
        assert int((res["class_counts"].cpu() - counts).abs().sum()) <= 4
        assert int(res["class_counts"].sum()) == count and not bool(res["cls"][count:].any())
        if graph is None:
            graph = inference.DetectionGraph(gpu, tuple(blob.shape), dev()).capture(blob, im_info)
        for got in [graph(blob, im_info) for _ in range(2)]:    # the replay is the static sequence (MIOpen / hipBLASLt may
            assert _same_detections(got[0], got[1], stat[:, 4], stat[:, :4], score_atol)   # pick other kernels under capture)
            assert sum(abs(len(c) - len(w)) for c, w in zip(got[2], want[2])) <= 4
            assert sum(len(c) for c in got[2][1:]) == got[0].numel()
        seen += want[0].numel()
    assert seen > 0, "the comparison never saw a detection"


def test_non_finite_network_outputs_do_not_fault(nets):
    """A diverged network hands NaN / Inf logits and deltas to the post-convolution half.  The results are meaningless,
    but every HIP operator must stay inside its buffers (the rank sort of the NMS treats NaN scores as the lowest,
    rejected boxes ride as far-away degenerate boxes, RoIs with a NaN corner pool through the guarded direct path)."""
    from detectron_pytorch_amd.rcnn import data as rdata

    _, gpu, cfg = nets
    gpu.train()
    boxes, classes, _ = scenario()
    entries = [dict(height=H, width=W, boxes=b, gt_classes=c, is_crowd=np.zeros(len(c), bool)) for b, c in zip(boxes, classes)]
    blobs = rdata.add_rpn_blobs(cfg, entries, [1.0, 1.0], np.random.RandomState(11))
    d = dev()
    roidb = {"gt_boxes": torch.from_numpy(np.concatenate(boxes)).to(d),
             "gt_classes": torch.from_numpy(np.concatenate(classes)).long().to(d),
             "gt_image": torch.tensor([0] * NUM_GT + [1] * NUM_GT, device=d)}
    rpn_t = {k: torch.from_numpy(v).to(d) for k, v in blobs.items() if k.startswith("rpn_")}
    rng = np.random.RandomState(3)
    for mode in ("nan_scores", "nan_deltas", "inf_deltas", "all_nan", "huge"):
        blobs_np, logits_np, deltas_np = synthetic_conv_outputs(seed=23, n=2)
        for a in logits_np:
            if mode in ("nan_scores", "all_nan"):
                a[rng.rand(*a.shape) < (1.0 if mode == "all_nan" else 0.3)] = np.nan
        for a in deltas_np:
            if mode in ("nan_deltas", "all_nan"):
                a[rng.rand(*a.shape) < (1.0 if mode == "all_nan" else 0.3)] = np.nan
            if mode == "inf_deltas":
                a[rng.rand(*a.shape) < 0.3] = np.inf
                a[rng.rand(*a.shape) < 0.1] = -np.inf
            if mode == "huge":
                a *= 1e30
        if mode in ("all_nan", "huge"):
            blobs_np = [b * np.float32(np.nan if mode == "all_nan" else 1e30) for b in blobs_np]
        blob_g = [torch.from_numpy(b).to(d).requires_grad_() for b in blobs_np]
        rpn_g = {}
        for i, lvl in enumerate(range(2, 7)):
            rpn_g["rpn_cls_logits_fpn%d" % lvl] = torch.from_numpy(logits_np[i]).to(d).requires_grad_()
            rpn_g["rpn_bbox_pred_fpn%d" % lvl] = torch.from_numpy(deltas_np[i]).to(d).requires_grad_()
        gpu.zero_grad()
        ret = gpu.forward_from_features(blob_g, rpn_g, torch.from_numpy(blobs["im_info"]), roidb, rpn_t, None)
        sum(ret["losses"].values()).backward()
        torch.cuda.synchronize()
        assert int(ret["blobs"]["num_fg"].min()) >= NUM_GT, mode      # the gt boxes are always sampled
    # and the device is still healthy
    assert float(torch.ones(4, device=d).sum()) == 4.0
