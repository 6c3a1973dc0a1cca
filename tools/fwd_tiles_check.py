#!/usr/bin/env python3
"""Iteration aid for roi_align_fwd_tiles: a list of shapes, each checked against the oracle and timed (HIP events),
synchronising after every call so that a fault names its shape.  usage: python tools/fwd_tiles_check.py [quick]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402
from detectron_pytorch_amd.roi_align import roi_align_forward, roi_align_fpn  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def one(tag, feat, rois, res, scale, sr, time_it=True):
    print("..", tag, feat.shape, rois.shape, res, sr, flush=True)
    f, r = torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev)
    out = roi_align_forward(f, r, res, res, scale, sr)
    torch.cuda.synchronize()
    ref = oracle.roi_align_forward(feat, rois, res, res, scale, sr, threads=16)
    got = out.cpu().numpy()
    err = np.abs(got - ref)
    bad = np.argwhere(~(err <= 1e-5))
    us = timed(lambda: roi_align_forward(f, r, res, res, scale, sr)) if time_it else float("nan")
    print("%-28s max|d| %.3g  bad %d  %.1f us (python call)" % (tag, np.nanmax(err), len(bad), us), flush=True)
    if len(bad):
        rr = sorted(set(int(b[0]) for b in bad))
        print("   bad rois", rr[:20], "first", bad[0], got[tuple(bad[0])], ref[tuple(bad[0])])
        for q in rr[:5]:
            print("   roi", q, rois[q], "channels", sorted(set(int(b[1]) for b in bad if b[0] == q))[:8])
    return len(bad) == 0


ok = True
h, w, scale = syn.FPN_LEVELS[2]
ok &= one("small adversarial 7", syn.feature_map(2, 32, 50, 84, 0), syn.rois_adversarial(64, 2, 50, 84, 1 / 16., 1), 7, 1 / 16., 2)
ok &= one("small adversarial 14", syn.feature_map(1, 16, 50, 84, 16), syn.rois_adversarial(40, 1, 50, 84, 1 / 16., 40), 14, 1 / 16., 2)
ok &= one("adaptive sr0", syn.feature_map(2, 32, 50, 84, 7), syn.rois_adversarial(96, 2, 50, 84, 1 / 16., 96), 7, 1 / 16., 0)
ok &= one("sr3 res6", syn.feature_map(1, 32, 9, 70, 9), syn.rois_adversarial(40, 1, 9, 70, 1 / 16., 40), 6, 1 / 16., 3)
ok &= one("config2", syn.feature_map(1, 256, h, w, 0), syn.rois_canonical(512, 1, seed=0), 7, scale, 2)
if len(sys.argv) < 2:
    ok &= one("mask 128x14", syn.feature_map(1, 256, h, w, 0), syn.rois_canonical(128, 1, seed=1), 14, scale, 2)
    ok &= one("box 1024 2img", syn.feature_map(2, 256, h, w, 0), syn.rois_canonical(1024, 2, seed=0), 7, scale, 2)
    ok &= one("3000 rois 1 img", syn.feature_map(1, 64, h, w, 0), syn.rois_canonical(3000, 1, seed=3), 7, scale, 2)
    hot = syn.rois_canonical(600, 1, seed=4)
    hot[:, 1:] = hot[:1, 1:] + np.random.RandomState(0).uniform(-6, 6, (600, 4)).astype(np.float32)
    ok &= one("600 rois on one spot", syn.feature_map(1, 64, h, w, 0), hot, 7, scale, 2)
    # FPN-fused
    frois, flv = syn.rois_fpn_distributed(1000, batch=2, seed=5)
    maps = [syn.feature_map(2, 256, syn.FPN_LEVELS[l][0], syn.FPN_LEVELS[l][1], seed=l) for l in (5, 4, 3, 2)]
    scales = [syn.FPN_LEVELS[l][2] for l in (5, 4, 3, 2)]
    idx = np.array([(5, 4, 3, 2).index(int(l)) for l in flv], dtype=np.int32)
    tm = [torch.from_numpy(m).to(dev) for m in maps]
    tr, ti = torch.from_numpy(frois).to(dev), torch.from_numpy(idx).to(dev)
    print(".. fpn", flush=True)
    out = roi_align_fpn(tm, scales, tr, ti, 7, 7, 2)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    worst = 0.0
    for li in range(4):
        sel = np.nonzero(idx == li)[0]
        if len(sel):
            ref = oracle.roi_align_forward(maps[li], frois[sel], 7, 7, scales[li], 2, threads=16)
            worst = max(worst, float(np.abs(got[sel] - ref).max()))
    us = timed(lambda: roi_align_fpn(tm, scales, tr, ti, 7, 7, 2))
    print("%-28s max|d| %.3g  %.1f us (python call)" % ("fpn 1000 P2-P5 2img", worst, us))
    ok &= worst <= 1e-5
print("ALL OK" if ok else "FAILURES")
