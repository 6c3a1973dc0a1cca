"""Generates tests/golden/model_keypoint.npz: outputs of the REFERENCE'S OWN Generalized_RCNN built from
configs/baselines/e2e_keypoint_rcnn_R-50-FPN_1x.yaml (lib/modeling/keypoint_rcnn_heads.py, lib/roi_data/keypoint_rcnn.py,
lib/utils/keypoints.py), executed on the CPU from /root/reference through oracle/ref_model.py, for the post-convolution
half of a training step.  A process of its own (the reference's cfg is a process-wide global; generate_model.py configures
it for Mask R-CNN).  Run here (the GPU box has no /root/reference):

    python tests/golden/generate_model_keypoint.py

Scenario: tests/scenarios.py:keypoint_scenario (2 images of 256x320, 4 person boxes with 17 key points each), 2 classes,
weights = reference initialisers under torch.manual_seed(3), RPN blobs under np.random.seed(11), sampling permutation
RandomState(7); the outputs of the convolutions are the seeded arrays of synthetic_conv_outputs(seed=23) substituted for
the reference's backbone / RPN convolutions, as in generate_model.py.  tests/test_e2e_gpu.py feeds the same arrays to the
HIP path on the GPU.
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
warnings.filterwarnings("ignore")

from oracle import ref_model  # noqa: E402
from scenarios import H, W, NUM_GT, keypoint_scenario, synthetic_conv_outputs  # noqa: E402
from generate_model import sample_index, substitute_convolutions  # noqa: E402

YAML = "configs/baselines/e2e_keypoint_rcnn_R-50-FPN_1x.yaml"
HEAD_PARAMS = ["Box_Head.fc1.weight", "Box_Outs.cls_score.weight", "Keypoint_Head.conv_fcn.0.weight",
               "Keypoint_Head.conv_fcn.14.bias", "Keypoint_Outs.classify.weight", "Keypoint_Outs.classify.bias"]


def main():
    ref_model.configure(YAML, MODEL__LOAD_IMAGENET_PRETRAINED_WEIGHTS=False, MODEL__NUM_CLASSES=2)
    ref = ref_model.build_model(seed=3)
    ref.train()
    boxes, classes, kps, data_np = keypoint_scenario()
    entries = [ref_model.roidb_entry(H, W, bx, c, 2, keypoints=k) for bx, c, k in zip(boxes, classes, kps)]
    blobs = ref_model.rpn_blobs(entries, [1.0, 1.0], seed=11)
    priority = np.random.RandomState(7).permutation(2 * NUM_GT + 2000).astype(np.float32)
    blobs_np, logits_np, deltas_np = synthetic_conv_outputs(seed=23, n=2)
    feats = [torch.from_numpy(b).requires_grad_() for b in blobs_np]
    undo = substitute_convolutions(ref, feats, logits_np, deltas_np)
    ref.zero_grad()
    ret, cap = ref_model.train_forward(ref, torch.from_numpy(data_np), blobs, priority, None)
    sum(v.sum() for v in ret["losses"].values()).backward()
    undo()
    out = {}
    b = cap["blobs"]
    out["train_collected_rois"] = cap["rois"]
    out["train_rois"] = b["rois"]
    out["train_labels"] = b["labels_int32"]
    out["keypoint_rois"] = b["keypoint_rois"]
    out["keypoint_locations_int32"] = b["keypoint_locations_int32"]
    out["keypoint_weights"] = b["keypoint_weights"]
    out["keypoint_loss_normalizer"] = np.float64(float(b["keypoint_loss_normalizer"]))
    assert b["keypoint_weights"].sum() > 20 and (b["keypoint_weights"] == 0).sum() > 20
    out["loss_names"] = np.array(sorted(ret["losses"].keys()))
    out["loss_values"] = np.array([float(ret["losses"][k]) for k in sorted(ret["losses"].keys())], dtype=np.float64)
    params = dict(ref.named_parameters())
    for name in HEAD_PARAMS:
        g = params[name].grad.detach().numpy().reshape(-1)
        out["grad_norm/" + name] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out["grad_samples/" + name] = g[sample_index(g.size)]
    for i, f in enumerate(feats[-4:]):          # P5, P4, P3, P2
        g = (torch.zeros_like(f) if f.grad is None else f.grad).detach().numpy().reshape(-1)
        out["feat_grad_norm/%d" % i] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out["feat_grad_samples/%d" % i] = g[sample_index(g.size, 512, seed=i)]
    path = os.path.join(ROOT, "tests", "golden", "model_keypoint.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; loss_kps = %.6f, keypoint rois %d"
          % (float(ret["losses"]["loss_kps"]), b["keypoint_rois"].shape[0]))


if __name__ == "__main__":
    main()
