#!/usr/bin/env python3
"""Eager Mask R-CNN test images one after the other: wall time per image (the periodic host stall shows as 60-80 ms images).
env: CUDNN_BENCHMARK=1 torch.backends.cudnn.benchmark; GC=off|report|freeze_after_warmup; RESPECT_QUOTA=1 caps the intra-op
pool at the cgroup CPU quota (detectron_pytorch_amd.hostcpu).  usage: python tools/stall_probe.py [images]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron_pytorch_amd.rcnn import config, inference, model as rmodel  # noqa: E402

if os.environ.get("RESPECT_QUOTA"):
    from detectron_pytorch_amd import hostcpu  # noqa: E402

    print("cgroup quota %s CPUs; intra-op threads %d -> %d" % (hostcpu.cpu_quota(), torch.get_num_threads(),
                                                                 hostcpu.respect_cpu_quota()))
if os.environ.get("CUDNN_BENCHMARK"):
    torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
cfg = config.mask_rcnn_r50_fpn()
cfg.TEST.SCORE_THRESH = 0.0
torch.manual_seed(cfg.RNG_SEED)
net = rmodel.GeneralizedRCNN(cfg).to(dev).eval()
data = torch.from_numpy((np.random.RandomState(0).randn(1, 3, 800, 1344) * 50).astype(np.float32)).to(dev)
im_info = torch.tensor([[800.0, 1344.0, 1.0]])
import gc  # noqa: E402

gc_log = []
if os.environ.get("GC") == "off":
    gc.disable()
elif os.environ.get("GC") == "report":
    _t = [0.0]

    def _cb(phase, info):
        if phase == "start":
            _t[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3, info["collected"]))

    gc.callbacks.append(_cb)
elif os.environ.get("GC") == "freeze_after_warmup":
    inference.im_detect_all_results(net, data, im_info)
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
times = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    torch.cuda.synchronize()
    t = time.perf_counter()
    inference.im_detect_all_results(net, data, im_info)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t) * 1e3)
print("ms per image:", " ".join("%.1f" % t for t in times))
if gc_log:
    print("gc: tracked objects %d; collections >= 2 ms: %s" % (
        len(gc.get_objects()), " ".join("gen%d:%.1fms(%d)" % g for g in gc_log if g[1] >= 2.0)))
print("median %.1f  mean(after 4) %.1f  max(after 4) %.1f" % (np.median(times), np.mean(times[4:]), np.max(times[4:])))
