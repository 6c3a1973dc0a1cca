#!/usr/bin/env python3
"""Tuning aid: host-side stage timings (ms) of the composed inference path that bench.py reports as `inference_path`
(GenerateProposals P2-P6 -> collect -> fused RoIAlign -> class-batched post-processing), one image.
    python tools/prof_inference_path.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from detectron_pytorch_amd import detection, fpn_proposals, generate_proposals as gp, synthetic as syn, _lib
from detectron_pytorch_amd.nms import nms_device_many
from detectron_pytorch_amd.roi_align import roi_align_fpn
device = torch.device("cuda", 0)
levels = [(2, 200, 336, 4, 32), (3, 100, 168, 8, 64), (4, 50, 84, 16, 128), (5, 25, 42, 32, 256), (6, 13, 21, 64, 512)]
ops, heads = [], []
for lvl, h, w, stride, size in levels:
    anchors = gp.generate_anchors(stride, (size,), (0.5, 1, 2))
    sc, dl = syn.rpn_head_outputs(1, 3, h, w, seed=lvl)
    ops.append(gp.GenerateProposalsOp(anchors, 1.0 / stride, 1000, 1000, 0.7, 0, as_numpy=False))
    heads.append((torch.from_numpy(sc).to(device), torch.from_numpy(dl).to(device)))
info = torch.tensor([[800, 1344, 1.0]], dtype=torch.float32, device=device)
def T(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r=fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("topk P2", T(lambda: torch.topk(heads[0][0].view(1,-1), 1000, dim=1)))
print("decode all", T(lambda: [op.decode(sc, dl, info) for op,(sc,dl) in zip(ops,heads)]))
dec=[op.decode(sc, dl, info) for op,(sc,dl) in zip(ops,heads)]
probs=[d[0][i] for d in dec for i in range(1)]
print("nms many", T(lambda: nms_device_many(probs, 0.7, _lib.NMS_GE_ORIG_ASC)))
kept=nms_device_many(probs, 0.7, _lib.NMS_GE_ORIG_ASC)
print("select all", T(lambda: [op.select(d[0], d[1], kept[i:i+1]) for i,(op,d) in enumerate(zip(ops,dec))]))
print("generate_and_collect", T(lambda: fpn_proposals.generate_and_collect(ops, heads, info, 1000)))
rois=fpn_proposals.generate_and_collect(ops, heads, info, 1000)
feats=[torch.from_numpy(syn.feature_map(1, 256, h, w, seed=l)).to(device) for l,h,w,_,_ in levels[3::-1]]
scales=[1.0/s for _,_,_,s,_ in levels[3::-1]]
print("levels+roialign", T(lambda: roi_align_fpn(feats, scales, rois, 5-fpn_proposals.map_rois_to_fpn_levels(rois[:,1:5]), 7,7,2)))
cls_np, box_np = syn.detection_head_outputs(1000, 81, seed=7)
cls, box = torch.from_numpy(cls_np).to(device), torch.from_numpy(box_np).to(device)
print("detection", T(lambda: detection.box_results_with_nms_and_limit(cls, box)))
def full():
    rois = fpn_proposals.generate_and_collect(ops, heads, info, 1000)
    lv = fpn_proposals.map_rois_to_fpn_levels(rois[:, 1:5])
    with torch.no_grad():
        pooled = roi_align_fpn(feats, scales, rois, 5 - lv, 7, 7, 2)
    return detection.box_results_with_nms_and_limit(cls[:rois.size(0)], box[:rois.size(0)])
print("full pipeline (wall)", T(full))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for _ in range(10): full()
e.record(); e.synchronize(); print("full pipeline (events)", s.elapsed_time(e) / 10)
