#!/usr/bin/env python3
"""Debug aid: step through forward / backward of one RoIAlign case with prints (run under `timeout`)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from detectron_pytorch_amd import synthetic as syn
from detectron_pytorch_amd.roi_align import roi_align_forward, roi_align_backward
import oracle

case = sys.argv[1] if len(sys.argv) > 1 else "canon"
dev = torch.device("cuda", 0)
if case == "canon":
    n, c, h, w, scale, res, sr, nrois = 1, 256, 200, 336, 0.25, 7, 2, 512
    feat = syn.feature_map(n, c, h, w, seed=0); rois = syn.rois_canonical(nrois, n, seed=0)
elif case == "small":
    n, c, h, w, scale, res, sr, nrois = 1, 32, 50, 84, 1.0 / 16, 7, 2, 16
    feat = syn.feature_map(n, c, h, w, seed=0); rois = syn.rois_canonical(nrois, n, seed=0, side=(64, 300))
else:
    n, c, h, w, scale, res, sr, nrois = 2, 64, 100, 168, 1.0 / 16, 7, 2, 128
    feat = syn.feature_map(n, c, h, w, seed=res + sr); rois = syn.rois_adversarial(nrois, n, h, w, scale, seed=nrois)
gtop = np.random.RandomState(7).randn(nrois, c, res, res).astype(np.float32)
f = torch.from_numpy(feat).to(dev); r = torch.from_numpy(rois).to(dev); g = torch.from_numpy(gtop).to(dev)
print("forward...", flush=True)
out, ws = roi_align_forward(f, r, res, res, scale, sr, return_workspace=True)
torch.cuda.synchronize(); print("forward done", flush=True)
ref = oracle.roi_align_forward(feat, rois, res, res, scale, sr, threads=8)
print("fwd max err", float(np.abs(out.cpu().numpy() - ref).max()), flush=True)
print("backward...", flush=True)
gin = roi_align_backward(g, r, tuple(f.shape), res, res, scale, sr, workspace=ws)
torch.cuda.synchronize(); print("backward done", flush=True)
refg = oracle.roi_align_backward(gtop, rois, feat.shape, scale, sr, threads=8)
err = np.abs(gin.cpu().numpy() - refg)
print("bwd max err", float(err.max()), "at", np.unravel_index(err.argmax(), err.shape), "ref absmax", float(np.abs(refg).max()), flush=True)
