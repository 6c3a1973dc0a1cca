#!/usr/bin/env python3
"""Iteration aid for the RoIAlign backward: the fused-FPN backward on the RoIs of a training step (tests/golden/
step_rois.npz; made by this script on a GPU box when absent) and the config-2 backward, for the slice lengths in SLICES
(MI_ROI_ALIGN_BWD_SLICE; 0 = unplanned).  usage: SLICES="0 16 32" python tools/bwd_time.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402
from detectron_pytorch_amd.roi_align import roi_align_backward, roi_align_forward  # noqa: E402
from tools.hot_path_bench import time_kernel  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "step_rois.npz")


def load_step_rois(dev):
    if os.path.exists(FIXTURE):
        g = np.load(FIXTURE)
        return tuple(torch.from_numpy(g[k]).to(dev) for k in ("rois", "levels", "mask_rois", "mask_levels"))
    from tools.bwd_clustered import step_rois

    t = step_rois(dev)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "step_rois.npz"),
                        **{k: v.cpu().numpy() for k, v in zip(("rois", "levels", "mask_rois", "mask_levels"), t)})
    return t


def fpn_case(dev, rois, lvls, res, iters):
    from tools.bwd_clustered import time_fpn

    return time_fpn(dev, rois, (5 - lvls).clamp(0, 3), res, iters)[1]


def config2_case(dev, iters):
    h, w, scale = syn.FPN_LEVELS[2]
    rois = torch.from_numpy(syn.rois_canonical(512, 1, seed=0)).to(dev)
    feat = torch.from_numpy(syn.feature_map(1, 256, h, w, seed=0)).to(dev)
    out, ws = roi_align_forward(feat, rois, 7, 7, scale, 2, return_workspace=True)
    g = torch.randn_like(out)
    return time_kernel(lambda: roi_align_backward(g, rois, (1, 256, h, w), 7, 7, scale, 2, workspace=ws), iters) * 1e6


def main():
    if os.environ.get("MI_LIB_OVERRIDE"):      # tuning builds of the library
        _lib.LIB_PATH = os.path.abspath(os.environ["MI_LIB_OVERRIDE"])
    dev = torch.device("cuda", 0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rois, lvls, mrois, mlvls = load_step_rois(dev)
    cases = os.environ.get("CASES", "box mask cfg2").split()
    for s in os.environ.get("SLICES", "0 32").split():
        os.environ["MI_ROI_ALIGN_BWD_SLICE"] = s
        _lib.lib().mi_dbg_reload_tuning()
        line = "slice %3s:" % s
        if "box" in cases:
            line += " step box 1024x7x7 bwd %.1f us |" % fpn_case(dev, rois, lvls, 7, iters)
        if "mask" in cases:
            line += " step mask %dx14x14 bwd %.1f us |" % (mrois.size(0), fpn_case(dev, mrois, mlvls, 14, iters))
        if "cfg2" in cases:
            line += " config-2 bwd %.1f us" % config2_case(dev, iters)
        print(line, flush=True)


if __name__ == "__main__":
    main()
