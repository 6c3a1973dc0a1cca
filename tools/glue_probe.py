"""Tuning aid: the static post-convolution inference glue (tools/hot_path_bench.inference_path's run_static) launched
eagerly N times, for `rocprofv3 --kernel-trace --stats` -- which kernels make up the per-image GPU time of the glue."""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron_pytorch_amd import detection, fpn_proposals, generate_proposals as gp, synthetic as syn  # noqa: E402
from detectron_pytorch_amd.roi_align import roi_align_fpn  # noqa: E402

device = torch.device("cuda", 0)
levels = [(2, 200, 336, 4, 32), (3, 100, 168, 8, 64), (4, 50, 84, 16, 128), (5, 25, 42, 32, 256), (6, 13, 21, 64, 512)]
ops, heads = [], []
for lvl, h, w, stride, size in levels:
    anchors = gp.generate_anchors(stride, (size,), (0.5, 1, 2))
    sc, dl = syn.rpn_head_outputs(1, 3, h, w, seed=lvl)
    ops.append(gp.GenerateProposalsOp(anchors, 1.0 / stride, 1000, 1000, 0.7, 0, as_numpy=False))
    heads.append((torch.from_numpy(sc).to(device), torch.from_numpy(dl).to(device)))
info = torch.tensor([[800, 1344, 1.0]], dtype=torch.float32, device=device)
feats = [torch.from_numpy(syn.feature_map(1, syn.FPN_DIM, h, w, seed=l)).to(device) for l, h, w, _, _ in levels[3::-1]]
scales = [1.0 / s for _, _, _, s, _ in levels[3::-1]]
cls_np, box_np = syn.detection_head_outputs(1000, 81, seed=7)
cls, box = torch.from_numpy(cls_np).to(device), torch.from_numpy(box_np).to(device)
part = sys.argv[2] if len(sys.argv) > 2 else "all"
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    if part in ("all", "proposals"):
        rois, valid, lv = fpn_proposals.generate_and_collect(ops, heads, info, 1000, static=True, with_levels=True)
        with torch.no_grad():
            pooled = roi_align_fpn(feats, scales, rois, 5 - lv, 7, 7, 2)
    if part in ("all", "detect"):
        res = detection.box_results_static(cls, box, roi_valid=None)
torch.cuda.synchronize()
