#!/usr/bin/env python3
"""Iteration aid for roi_align_fwd_pipe (MI_ROI_ALIGN_IMPL=pipe): (1) outputs against the default record kernels (bit for
bit on NCHW fast-path RoIs, 1e-5 otherwise) on the shapes a step uses, (2) HIP-event time per mi_roi_align_forward_ws
call, default against pipe, NCHW and channels-last, (3) with a tuning build, the ablation masks.
usage: python tools/pipe_check.py [check] [time] [ablate]   (default: check time)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402
from tools.hot_path_bench import time_kernel  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
stream = _lib.current_stream_handle(dev)
what = sys.argv[1:] or ["check", "time"]


def set_env(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    lib.mi_dbg_reload_tuning()


def forward(feat, rois, res, scale, sr, nhwc, ws=None, out=None):
    n, c, h, w = feat.shape
    r = rois.shape[0]
    f = feat.permute(0, 2, 3, 1).contiguous() if nhwc else feat
    if out is None:
        out = torch.full((r, c, res, res), float("nan"), device=dev)
    if ws is None:
        ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(r), dtype=torch.uint8, device=dev)
    rc = lib.mi_roi_align_forward_ws(f.data_ptr(), rois.data_ptr(), out.data_ptr(), n, c, h, w, r, res, res, scale, sr, 0,
                                     1 if nhwc else 0, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, lib.mi_last_error()
    return out


def cases():
    h, w, scale = syn.FPN_LEVELS[2]
    yield "config2 512x256x7x7", syn.feature_map(1, 256, h, w, seed=0), syn.rois_canonical(512, 1, seed=0), 7, scale, 2
    yield "mask 128x256x14x14", syn.feature_map(1, 256, h, w, seed=1), syn.rois_canonical(128, 1, seed=1), 14, scale, 2
    yield "two images 1024x256x7x7", syn.feature_map(2, 256, h, w, seed=2), syn.rois_canonical(1024, 2, seed=2), 7, scale, 2
    h5, w5, s5 = syn.FPN_LEVELS[4]
    yield "adversarial 160x64 P4", syn.feature_map(2, 64, h5, w5, seed=3), syn.rois_adversarial(160, 2, h5, w5, s5, seed=4), 7, s5, 2
    yield "adaptive grid 96x32", syn.feature_map(2, 32, h5, w5, seed=5), syn.rois_adversarial(96, 2, h5, w5, s5, seed=6), 7, s5, 0
    yield "big windows 64x32x14x14", syn.feature_map(1, 32, h, w, seed=7), syn.rois_canonical(64, 1, seed=8, side=(200.0, 700.0)), 14, scale, 2
    yield "3 rois", syn.feature_map(1, 32, h5, w5, seed=9), syn.rois_canonical(3, 1, seed=9), 7, s5, 2
    yield "sr 3", syn.feature_map(1, 96, h5, w5, seed=10), syn.rois_adversarial(77, 1, h5, w5, s5, seed=11), 7, s5, 3


if "check" in what:
    bad = 0
    for name, feat_np, rois_np, res, scale, sr in cases():
        feat, rois = torch.from_numpy(feat_np).to(dev), torch.from_numpy(rois_np).to(dev)
        for nhwc in (False, True):
            set_env(MI_ROI_ALIGN_IMPL=None)
            want = forward(feat, rois, res, scale, sr, nhwc)
            set_env(MI_ROI_ALIGN_IMPL="pipe")
            got = forward(feat, rois, res, scale, sr, nhwc)
            torch.cuda.synchronize()
            nan = int(torch.isnan(got).sum())
            err = float((got - want).abs().max()) if nan == 0 else float("nan")
            equal = bool(torch.equal(got, want))
            ok = nan == 0 and err <= 1e-5
            bad += 0 if ok else 1
            print("%-28s %s  max|pipe - records| %.2e  bit-equal %s  nan %d  %s" %
                  (name, "NHWC" if nhwc else "NCHW", err, equal, nan, "ok" if ok else "FAIL"), flush=True)
    set_env(MI_ROI_ALIGN_IMPL=None)
    print("check:", "all ok" if bad == 0 else "%d FAILED" % bad, flush=True)

if "time" in what or "ablate" in what:
    h, w, scale = syn.FPN_LEVELS[2]
    shapes = [("config2", 1, 512, 7), ("mask128", 1, 128, 14), ("box1024x2img", 2, 1024, 7)]
    results = []
    masks = [int(m) for m in os.environ.get("MASKS", "0 1 2 4 3 5 6 7").split()] if "ablate" in what else [0]
    for sname, n, r, res in shapes:
        feat = torch.from_numpy(syn.feature_map(n, 256, h, w, seed=0)).to(dev)
        rois = torch.from_numpy(syn.rois_canonical(r, n, seed=0)).to(dev)
        out = torch.empty((r, 256, res, res), device=dev)
        ws = torch.empty(lib.mi_roi_align_forward_workspace_bytes(r), dtype=torch.uint8, device=dev)
        for nhwc in (False, True):
            f = feat.permute(0, 2, 3, 1).contiguous() if nhwc else feat

            def launch():
                assert lib.mi_roi_align_forward_ws(f.data_ptr(), rois.data_ptr(), out.data_ptr(), n, 256, h, w, r, res, res,
                                                   scale, 2, 0, 1 if nhwc else 0, ws.data_ptr(), ws.numel(), stream) == 0

            for impl in (None, "pipe"):
                for m in (masks if impl == "pipe" or "ablate" in what else [0]):
                    set_env(MI_ROI_ALIGN_IMPL=impl, MI_ROI_ALIGN_ABLATE=m if m else None)
                    us = time_kernel(launch, 200) * 1e6
                    rec = {"shape": sname, "layout": "NHWC" if nhwc else "NCHW", "impl": impl or "records", "ablate": m,
                           "us_per_call": round(us, 2)}
                    results.append(rec)
                    print(json.dumps(rec), flush=True)
    set_env(MI_ROI_ALIGN_IMPL=None, MI_ROI_ALIGN_ABLATE=None)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pipe_times.jsonl"), "a") as fh:
        for rec in results:
            fh.write(json.dumps(rec) + "\n")
