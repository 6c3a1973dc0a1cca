#!/usr/bin/env python3
"""Tuning aid: where a visit of roi_align_bwd_tiles goes (config 2 and the RoIs of a training step).  Needs a TUNING build of
the library (MI_TUNING_BUILD=1 python -m detectron_pytorch_amd.build, or MI_LIB_OVERRIDE=<that .so>): wave 0 of every
workgroup sums clock64() differences (shader clock, MI_SHADER_MHZ, default 2100) per phase over its visits:
wait for the landing + barrier B1 | issue of the next RoI's pieces | pass 1 + barrier B2 | pass 2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import _lib, synthetic as syn  # noqa: E402

if os.environ.get("MI_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MI_LIB_OVERRIDE"])
from detectron_pytorch_amd.roi_align import _backward_workspace_bytes  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
stream = _lib.current_stream_handle(dev)
MHZ = float(os.environ.get("MI_SHADER_MHZ", "2100"))
h, w, scale = syn.FPN_LEVELS[2]
c, r, res, sr = syn.FPN_DIM, 512, 7, 2
rois = torch.from_numpy(syn.rois_canonical(r, 1, seed=0)).to(dev)
gtop = torch.randn(r, c, res, res, device=dev)
gin = torch.zeros(1, c, h, w, device=dev)
for planned in (False, True):
    nbytes = max(lib.mi_roi_align_forward_workspace_bytes(r), _backward_workspace_bytes([(h, w)], 1, r) if planned else 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tl = torch.zeros((8192, 8), dtype=torch.int64, device=dev)

    def launch():
        assert lib.mi_roi_align_backward_ws(gtop.data_ptr(), rois.data_ptr(), gin.data_ptr(), 1, c, h, w, r, res, res, scale, sr,
                                            0, 0, ws.data_ptr(), ws.numel(), 2, stream) == 0

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    lib.mi_dbg_roi_align_timeline(tl.data_ptr())
    launch()
    torch.cuda.synchronize()
    lib.mi_dbg_roi_align_timeline(None)
    t = tl.cpu().numpy()
    t = t[t[:, 5] != 0]
    assert len(t), "no stamps: this is not a tuning build of the library"
    visits = t[:, 6].astype(np.float64)
    busy = visits > 0
    names = ["wait for landing + barrier B1", "issue of the next RoI's pieces", "pass 1 + barrier B2", "pass 2 (wave 0)"]
    print("config 2, %s launch: %d workgroups, %d with RoIs, visits per workgroup mean %.1f max %d" % (
        "planned" if planned else "unplanned", len(t), busy.sum(), visits[busy].mean(), visits.max()))
    for k in range(4):
        per = t[busy, k] / visits[busy] / MHZ
        print("  %-32s %.3f us per visit (p90 %.3f)" % (names[k], per.mean(), np.percentile(per, 90)))
    tot = t[busy, :4].sum(1) / visits[busy] / MHZ
    print("  %-32s %.3f us per visit" % ("sum", tot.mean()))
    print("  list + first issue %.2f us; workgroup life (before the stores) mean %.2f p90 %.2f max %.2f us" % (
        (t[busy, 4] / MHZ).mean(), (t[busy, 5] / MHZ).mean(), np.percentile(t[busy, 5] / MHZ, 90), (t[:, 5] / MHZ).max()))
