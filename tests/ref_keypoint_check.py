"""Run by tests/test_model_cpu.py in a process of its own (the reference's cfg is a process-wide global and the other tests
configure it for Mask R-CNN): the keypoint branch -- seeded weights, keypoint labelling (roi_data/keypoint_rcnn.py),
heat-map targets (utils/keypoints.py:160-211), loss and gradients -- against the reference's own code on the CPU."""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

from oracle import ref_model  # noqa: E402
import cpu_backend  # noqa: E402
from scenarios import H, W, NUM_GT, keypoint_scenario  # noqa: E402

YAML = "configs/baselines/e2e_keypoint_rcnn_R-50-FPN_1x.yaml"


def main():
    ref_cfg = ref_model.configure(YAML, MODEL__LOAD_IMAGENET_PRETRAINED_WEIGHTS=False, MODEL__NUM_CLASSES=2)
    from detectron_pytorch_amd.rcnn import config, model

    cfg = config.default_config().merge_from_file(os.path.join(ref_model.REFERENCE, YAML))
    cfg.MODEL.NUM_CLASSES = 2
    assert cfg.MODEL.KEYPOINTS_ON and cfg.KRCNN.HEATMAP_SIZE == ref_cfg.KRCNN.HEATMAP_SIZE == 56
    ref = ref_model.build_model(seed=3)
    torch.manual_seed(3)
    mine = model.GeneralizedRCNN(cfg)
    a, b = ref.state_dict(), mine.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a), "seeded weights differ"
    from detectron_pytorch_amd.rcnn import weights

    want_map, want_orph = ref.detectron_weight_mapping
    got_map, got_orph = weights.detectron_weight_mapping(mine)
    assert got_map == want_map and sorted(got_orph) == sorted(want_orph), "Detectron name mapping differs"

    boxes, classes, kps, data_np = keypoint_scenario()
    entries = [ref_model.roidb_entry(H, W, bx, c, 2, keypoints=k) for bx, c, k in zip(boxes, classes, kps)]
    blobs = ref_model.rpn_blobs(entries, [1.0, 1.0], seed=11)
    data = torch.from_numpy(data_np)
    g = 2 * NUM_GT
    priority = np.random.RandomState(7).permutation(g + 2000).astype(np.float32)
    ref.train()
    mine.train()
    ref.zero_grad()
    ret_ref, cap = ref_model.train_forward(ref, data, blobs, priority, None)
    sum(v.sum() for v in ret_ref["losses"].values()).backward()
    roidb = {"gt_boxes": torch.from_numpy(np.concatenate(boxes)), "gt_classes": torch.ones(g, dtype=torch.long),
             "gt_image": torch.tensor([0] * NUM_GT + [1] * NUM_GT), "gt_keypoints": torch.from_numpy(np.concatenate(kps))}
    rpn_t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in blobs.items() if k.startswith("rpn_")}
    mine.zero_grad()
    with cpu_backend.cpu_ops(mine):
        from detectron_pytorch_amd import fpn_proposals

        inner = fpn_proposals.generate_and_collect

        def collect_in_reference_order(*a, **k):
            rois, valid = inner(*a, **k)
            key = lambda x: x[np.lexsort(x.T[::-1])]  # noqa: E731
            assert np.array_equal(key(rois.numpy()), key(cap["rois"]))
            return torch.from_numpy(cap["rois"]), valid

        fpn_proposals.generate_and_collect = collect_in_reference_order
        ret = mine(data, torch.from_numpy(blobs["im_info"]), roidb=roidb, rpn_targets=rpn_t,
                   priority=torch.from_numpy(priority[:g + cap["rois"].shape[0]]))
        sum(ret["losses"].values()).backward()
    b, want = ret["blobs"], cap["blobs"]
    fper = int(round(cfg.TRAIN.FG_FRACTION * cfg.TRAIN.BATCH_SIZE_PER_IM))
    rows = torch.cat([i * fper + torch.arange(int(n)) for i, n in enumerate(b["num_keypoint_rois"])])
    assert rows.numel() == want["keypoint_rois"].shape[0] >= g, (rows.numel(), want["keypoint_rois"].shape)
    assert np.array_equal(b["keypoint_rois"][rows].numpy(), want["keypoint_rois"])
    k17 = (rows.view(-1, 1) * 17 + torch.arange(17).view(1, -1)).reshape(-1)
    assert np.array_equal(b["keypoint_locations_int32"][k17].numpy(), want["keypoint_locations_int32"])
    assert np.array_equal(b["keypoint_weights"][k17].numpy(), want["keypoint_weights"])
    assert want["keypoint_weights"].sum() > 20 and (want["keypoint_weights"] == 0).sum() > 20
    for lvl in range(2, 6):
        sel = rows[b["keypoint_rois_levels"][rows] == lvl]
        assert np.array_equal(b["keypoint_rois"][sel].numpy(), want["keypoint_rois_fpn%d" % lvl]), lvl
    np.testing.assert_allclose(float(b["keypoint_loss_normalizer"]), float(want["keypoint_loss_normalizer"]), rtol=1e-6)
    for k, v in ret_ref["losses"].items():
        np.testing.assert_allclose(float(ret["losses"][k]), float(v), rtol=2e-5, atol=1e-7, err_msg=k)
    pr, pm = dict(ref.named_parameters()), dict(mine.named_parameters())
    for name in ("Keypoint_Head.conv_fcn.0.weight", "Keypoint_Head.conv_fcn.14.bias", "Keypoint_Outs.classify.weight",
                 "Conv_Body.posthoc_modules.3.weight"):
        x, y = pr[name].grad, pm[name].grad
        err = (x - y).abs().max().item() / max(x.abs().max().item(), 1e-12)
        assert err <= 2e-4, (name, err)
    print("KEYPOINT_PARITY_OK loss_kps=%.6f rois=%d" % (float(ret["losses"]["loss_kps"]), rows.numel()))


if __name__ == "__main__":
    main()
