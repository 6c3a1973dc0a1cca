"""Run by tests/test_model_cpu.py in a process of its own (the reference's cfg is a process-wide global): Mask R-CNN R-50-FPN
with RPN.CLS_ACTIVATION = 'softmax' (jwyang's convention, config.py:661-663; FPN.py:335-336, 399-404, 438-445) -- seeded
weights (the objectness convolution has 2 x A channels), Detectron name mapping, collected proposals, every loss and the
gradients of the RPN parameters -- against the reference's own code on the CPU."""
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
from scenarios import H, W, NUM_GT, scenario  # noqa: E402
from test_model_cpu import rect_rasterizer  # noqa: E402

YAML = "configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml"


def main():
    ref_cfg = ref_model.configure(YAML, MODEL__LOAD_IMAGENET_PRETRAINED_WEIGHTS=False, MODEL__NUM_CLASSES=81,
                                  RPN__CLS_ACTIVATION="softmax")
    from detectron_pytorch_amd.rcnn import config, model, weights

    cfg = config.mask_rcnn_r50_fpn()
    cfg.RPN.CLS_ACTIVATION = "softmax"
    assert ref_cfg.RPN.CLS_ACTIVATION == "softmax"
    ref = ref_model.build_model(seed=3)
    torch.manual_seed(3)
    mine = model.GeneralizedRCNN(cfg)
    a, b = ref.state_dict(), mine.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a), "seeded weights differ"
    assert b["RPN.FPN_RPN_cls_score.weight"].shape[0] == 2 * len(cfg.FPN.RPN_ASPECT_RATIOS)
    want_map, want_orph = ref.detectron_weight_mapping
    got_map, got_orph = weights.detectron_weight_mapping(mine)
    assert got_map == want_map and sorted(got_orph) == sorted(want_orph), "Detectron name mapping differs"

    boxes, classes, data_np = scenario()
    entries = [ref_model.roidb_entry(H, W, bx, c, 81) for bx, c in zip(boxes, classes)]
    blobs = ref_model.rpn_blobs(entries, [1.0, 1.0], seed=11)
    data = torch.from_numpy(data_np)
    g = 2 * NUM_GT
    priority = np.random.RandomState(7).permutation(g + 2000).astype(np.float32)
    ref.train()
    mine.train()
    ref.zero_grad()
    ret_ref, cap = ref_model.train_forward(ref, data, blobs, priority, rect_rasterizer)
    sum(v.sum() for v in ret_ref["losses"].values()).backward()
    roidb = {"gt_boxes": torch.from_numpy(np.concatenate(boxes)), "gt_classes": torch.from_numpy(np.concatenate(classes)).long(),
             "gt_image": torch.tensor([0] * NUM_GT + [1] * NUM_GT)}
    rpn_t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in blobs.items() if k.startswith("rpn_")}
    mine.zero_grad()
    with cpu_backend.cpu_ops(mine):
        from detectron_pytorch_amd import fpn_proposals

        inner = fpn_proposals.generate_and_collect

        def collect_in_reference_order(*a, **k):
            rois, valid = inner(*a, **k)
            key = lambda x: x[np.lexsort(x.T[::-1])]  # noqa: E731
            assert np.array_equal(key(rois.numpy()), key(cap["rois"])), "collected proposals differ as a set"
            return torch.from_numpy(cap["rois"]), valid

        fpn_proposals.generate_and_collect = collect_in_reference_order
        ret = mine(data, torch.from_numpy(blobs["im_info"]), roidb=roidb, rpn_targets=rpn_t,
                   priority=torch.from_numpy(priority[:g + cap["rois"].shape[0]]))
        sum(ret["losses"].values()).backward()
    assert sorted(ret["losses"]) == sorted(ret_ref["losses"])
    for k, v in ret_ref["losses"].items():
        np.testing.assert_allclose(float(ret["losses"][k]), float(v), rtol=2e-5, atol=1e-7, err_msg=k)
    pr, pm = dict(ref.named_parameters()), dict(mine.named_parameters())
    for name in ("RPN.FPN_RPN_cls_score.weight", "RPN.FPN_RPN_cls_score.bias", "RPN.FPN_RPN_conv.weight",
                 "Conv_Body.posthoc_modules.3.weight"):
        x, y = pr[name].grad, pm[name].grad
        err = (x - y).abs().max().item() / max(x.abs().max().item(), 1e-12)
        assert err <= 2e-4, (name, err)
    print("SOFTMAX_RPN_PARITY_OK loss_rpn_cls_fpn2=%.6f rois=%d" % (float(ret["losses"]["loss_rpn_cls_fpn2"]),
                                                                    cap["rois"].shape[0]))


if __name__ == "__main__":
    main()
