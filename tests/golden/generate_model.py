"""Generates tests/golden/model.npz: outputs of the REFERENCE'S OWN Generalized_RCNN (lib/modeling/model_builder.py),
executed on the CPU from /root/reference through oracle/ref_model.py, for the post-convolution half of a training step
and of an inference step.  Run here (the GPU box has no /root/reference):

    python tests/golden/generate_model.py

Scenario (tests/test_model_cpu.py:scenario): 2 images of 256x320, 4 gt boxes each, weights = reference initialisers
under torch.manual_seed(3), RPN blobs from roi_data/rpn.py under np.random.seed(11), sampling permutation
RandomState(7).  Inputs are regenerated from these seeds by the tests; the file stores only what the reference computed:
collected proposals, labelled RoI blobs (compact), losses, gradient samples.  tests/test_e2e_gpu.py feeds the SAME
convolution outputs (computed on the CPU with this package's graph, which test_model_cpu.py pins bit-for-bit to the
reference's) to the HIP path on the GPU and compares.
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

from oracle import ref_model  # noqa: E402
import test_model_cpu as T  # noqa: E402

HEAD_PARAMS = ["Box_Head.fc1.weight", "Box_Head.fc2.bias", "Box_Outs.cls_score.weight", "Box_Outs.bbox_pred.weight",
               "Mask_Head.conv_fcn.0.weight", "Mask_Head.upconv.weight", "Mask_Outs.classify.weight",
               "Mask_Outs.classify.bias", "RPN.FPN_RPN_conv.weight", "RPN.FPN_RPN_bbox_pred.bias"]


def sample_index(numel, count=256, seed=0):
    return np.random.RandomState(seed).randint(0, numel, size=min(count, numel))


def main():
    ref_cfg = ref_model.configure("configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml",
                                  MODEL__LOAD_IMAGENET_PRETRAINED_WEIGHTS=False, MODEL__NUM_CLASSES=81)
    ref = ref_model.build_model(seed=3)
    out = {}
    # ---- training ----
    ref.train()
    boxes, classes, data_np = T.scenario()
    entries = [ref_model.roidb_entry(T.H, T.W, b, c, 81) for b, c in zip(boxes, classes)]
    blobs = ref_model.rpn_blobs(entries, [1.0, 1.0], seed=11)
    priority = np.random.RandomState(7).permutation(2 * T.NUM_GT + 2000).astype(np.float32)
    feats = {}
    body_forward = ref.Conv_Body.forward

    def keep_features(x):
        res = body_forward(x)
        for i, b in enumerate(res):
            feats[i] = b
        return res

    ref.Conv_Body.forward = keep_features
    ref.zero_grad()
    ret, cap = ref_model.train_forward(ref, torch.from_numpy(data_np), blobs, priority, T.rect_rasterizer)
    # gradients of the RoI-head losses w.r.t. the pyramid (what the RoIAlign backward delivers; the RPN head's own
    # contribution to the same maps is left out so that the post-convolution half can be compared in isolation)
    head_loss = sum(ret["losses"][k].sum() for k in ("loss_cls", "loss_bbox", "loss_mask"))
    roi_feats = [feats[i] for i in sorted(feats) if i >= len(feats) - 4]
    feat_grads = torch.autograd.grad(head_loss, roi_feats, retain_graph=True, allow_unused=True)
    feat_grads = [torch.zeros_like(f) if g is None else g for f, g in zip(roi_feats, feat_grads)]   # level without RoIs
    sum(v.sum() for v in ret["losses"].values()).backward()
    ref.Conv_Body.forward = body_forward
    out["train_collected_rois"] = cap["rois"]
    b = cap["blobs"]
    out["train_rois"] = b["rois"]
    out["train_labels"] = b["labels_int32"]
    rows = np.arange(b["rois"].shape[0])
    cols = 4 * np.maximum(b["labels_int32"], 0)[:, None] + np.arange(4)[None, :]
    out["train_bbox_targets4"] = b["bbox_targets"][rows[:, None], cols]          # the row's class columns
    assert np.count_nonzero(b["bbox_targets"]) == np.count_nonzero(out["train_bbox_targets4"][b["labels_int32"] > 0])
    out["train_mask_rois"] = b["mask_rois"]
    fg_cls = b["labels_int32"][b["labels_int32"] > 0]
    m2 = 28 * 28
    masks = b["masks_int32"].reshape(len(fg_cls), 81, m2)
    out["train_masks"] = masks[np.arange(len(fg_cls)), fg_cls].astype(np.int8)
    assert (np.delete(masks, 0, axis=1) >= 0).sum() == (out["train_masks"] >= 0).sum()
    out["loss_names"] = np.array(sorted(ret["losses"].keys()))
    out["loss_values"] = np.array([float(ret["losses"][k]) for k in sorted(ret["losses"].keys())], dtype=np.float64)
    out["accuracy_cls"] = np.float64(float(ret["metrics"]["accuracy_cls"]))
    params = dict(ref.named_parameters())
    for name in HEAD_PARAMS:
        g = params[name].grad.detach().numpy().reshape(-1)
        idx = sample_index(g.size)
        out["grad_norm/" + name] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out["grad_samples/" + name] = g[idx]
    for i, g in enumerate(feat_grads):          # i = 0..3: P5, P4, P3, P2
        g = g.detach().numpy().reshape(-1)
        out["feat_grad_norm/%d" % i] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out["feat_grad_samples/%d" % i] = g[sample_index(g.size, 512, seed=i)]
        nz = np.flatnonzero(g)
        out["feat_grad_nonzero_samples/%d" % i] = g[nz[sample_index(nz.size, 256, seed=10 + i)]] if nz.size else g[:0]
        out["feat_grad_nonzero_index/%d" % i] = nz[sample_index(nz.size, 256, seed=10 + i)] if nz.size else nz
    # ---- inference ----
    ref.eval()
    _, _, data_eval = T.scenario(seed=9)
    with torch.no_grad():
        want = ref(torch.from_numpy(data_eval[:1]), torch.tensor([[float(T.H), float(T.W), 1.0]]))
    out["eval_rois"] = want["rois"]
    cls = want["cls_score"].numpy()
    bbox = want["bbox_pred"].numpy()
    out["eval_cls_score_rows"] = cls[:64]
    out["eval_cls_score_sum"] = cls.astype(np.float64).sum(axis=1)
    out["eval_bbox_pred_rows"] = bbox[:64]
    import utils.boxes as box_utils

    deltas = bbox * 30
    pred = box_utils.clip_tiled_boxes(box_utils.bbox_transform(want["rois"][:, 1:5], deltas, ref_cfg.MODEL.BBOX_REG_WEIGHTS),
                                      (T.H, T.W))
    out["eval_pred_boxes_rows"] = pred[:64]
    path = os.path.join(ROOT, "tests", "golden", "model.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: getattr(v, "shape", None) for k, v in out.items()
                                                          if not k.startswith(("grad", "feat"))})


if __name__ == "__main__":
    main()
