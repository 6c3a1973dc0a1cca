"""Generates tests/golden/model.npz: outputs of the REFERENCE'S OWN Generalized_RCNN (lib/modeling/model_builder.py),
executed on the CPU from /root/reference through oracle/ref_model.py, for the post-convolution half of a training step
and of an inference step.  Run here (the GPU box has no /root/reference):

    python tests/golden/generate_model.py

Scenario (tests/test_model_cpu.py:scenario): 2 images of 256x320, 4 gt boxes each, weights = reference initialisers
under torch.manual_seed(3), RPN blobs from roi_data/rpn.py under np.random.seed(11), sampling permutation
RandomState(7).  The OUTPUTS OF THE CONVOLUTIONS (pyramid P6..P2, RPN logits and deltas of every level) are seeded
numpy arrays (`synthetic_conv_outputs`) substituted for the reference's backbone / RPN convolutions: CPU convolution
results differ in the last bits between machines (oneDNN picks kernels by ISA and thread count), and the fixture must be
reproducible on the GPU box.  Everything after the convolutions is the reference's own code.  Inputs are regenerated from
the seeds by the tests; the file stores only what the reference computed: collected proposals, labelled RoI blobs
(compact), losses, gradient samples.  tests/test_e2e_gpu.py feeds the same arrays to the HIP path on the GPU.
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
from scenarios import synthetic_conv_outputs  # noqa: E402

HEAD_PARAMS = ["Box_Head.fc1.weight", "Box_Head.fc2.bias", "Box_Outs.cls_score.weight", "Box_Outs.bbox_pred.weight",
               "Mask_Head.conv_fcn.0.weight", "Mask_Head.upconv.weight", "Mask_Outs.classify.weight",
               "Mask_Outs.classify.bias"]


def sample_index(numel, count=256, seed=0):
    return np.random.RandomState(seed).randint(0, numel, size=min(count, numel))


class _Replay(torch.nn.Module):
    """Stands in for a convolution shared by all RPN levels (FPN.py:387-389): returns the next preset tensor."""

    def __init__(self, tensors):
        super().__init__()
        self.tensors, self.i = tensors, 0

    def forward(self, x):
        t = self.tensors[self.i % len(self.tensors)]
        self.i += 1
        assert t.shape[2:] == x.shape[2:]
        return t


def substitute_convolutions(ref, feats, logits_np, deltas_np):
    """Make the reference model see the preset convolution outputs; returns the function that restores it."""
    body_forward = ref.Conv_Body.forward
    cls, bbox = ref.RPN.FPN_RPN_cls_score, ref.RPN.FPN_RPN_bbox_pred
    ref.Conv_Body.forward = lambda x: list(feats)
    ref.RPN.FPN_RPN_cls_score = _Replay([torch.from_numpy(a).requires_grad_() for a in logits_np])
    ref.RPN.FPN_RPN_bbox_pred = _Replay([torch.from_numpy(a).requires_grad_() for a in deltas_np])

    def undo():
        ref.Conv_Body.forward = body_forward
        ref.RPN.FPN_RPN_cls_score, ref.RPN.FPN_RPN_bbox_pred = cls, bbox

    return undo


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
    blobs_np, logits_np, deltas_np = synthetic_conv_outputs(seed=21, n=2)
    feats = [torch.from_numpy(b).requires_grad_() for b in blobs_np]
    undo = substitute_convolutions(ref, feats, logits_np, deltas_np)
    ref.zero_grad()
    ret, cap = ref_model.train_forward(ref, torch.from_numpy(data_np), blobs, priority, T.rect_rasterizer)
    sum(v.sum() for v in ret["losses"].values()).backward()
    undo()
    feat_grads = [torch.zeros_like(f) if f.grad is None else f.grad for f in feats[-4:]]   # P5, P4, P3, P2
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
    blobs_np, logits_np, deltas_np = synthetic_conv_outputs(seed=22, n=1)
    undo = substitute_convolutions(ref, [torch.from_numpy(b) for b in blobs_np], logits_np, deltas_np)
    with torch.no_grad():
        want = ref(torch.from_numpy(data_eval[:1]), torch.tensor([[float(T.H), float(T.W), 1.0]]))
    undo()
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
