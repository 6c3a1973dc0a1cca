"""Tuning aid: where does a Mask R-CNN test image spend its time (boxes / post-processing / mask head / result formats)?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron_pytorch_amd import detection  # noqa: E402
from detectron_pytorch_amd.rcnn import config, inference, model as rmodel, results  # noqa: E402

dev = torch.device("cuda", 0)
cfg = config.mask_rcnn_r50_fpn()
cfg.TEST.SCORE_THRESH = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
torch.manual_seed(cfg.RNG_SEED)
net = rmodel.GeneralizedRCNN(cfg).to(dev).eval()
data = torch.from_numpy((np.random.RandomState(0).randn(1, 3, 800, 1344) * 50).astype(np.float32)).to(dev)
im_info = torch.tensor([[800.0, 1344.0, 1.0]])


if os.environ.get("GC_FREEZE"):
    import gc

    gc.collect()
    gc.freeze()


def timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t) * 1e3


for it in range(5):
    (scores, boxes, blob), t_box = timed(lambda: inference.im_detect_bbox(net, data, im_info, (800, 1344)))
    (_, boxes_out, cls_boxes), t_post = timed(lambda: detection.box_results_with_nms_and_limit(
        scores, boxes, score_thresh=cfg.TEST.SCORE_THRESH, nms_thresh=cfg.TEST.NMS, detections_per_im=100))
    masks, t_mask = timed(lambda: inference.im_detect_mask(net, 1.0, boxes_out, blob))
    segms, t_segm = timed(lambda: results.segm_results(cls_boxes, masks, boxes_out, 800, 1344, cfg))
    print("iter %d: bbox %.2f ms, post %.2f ms, mask head %.2f ms, segm %.2f ms, dets %d" %
          (it, t_box, t_post, t_mask, t_segm, boxes_out.size(0)), flush=True)

for it in range(4):
    out, t_all = timed(lambda: inference.im_detect_all_results(net, data, im_info))
    print("im_detect_all_results: %.2f ms" % t_all, flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(6):
    out = inference.im_detect_all_results(net, data, im_info)
torch.cuda.synchronize()
print("loop of 6 without syncs: %.2f ms per image" % ((time.perf_counter() - t0) / 6 * 1e3), flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for it in range(3):
    out = inference.im_detect_all_results(net, data, im_info)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
