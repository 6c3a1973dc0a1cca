#!/usr/bin/env python3
"""RoIAlign over the RoIs a training step really samples (config-2 variant (iii)): one forward of the bench's Mask R-CNN
step gives the 1024 labelled box RoIs of the 2-image minibatch with their FPN levels (they cluster on the 8 gt boxes per
image) and the <= 256 mask RoIs; the fused FPN call is then timed forward and backward on seeded pyramid maps, next to the
same number of uniformly spread RoIs.  usage: python tools/bwd_clustered.py [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron_pytorch_amd import synthetic as syn  # noqa: E402
from tools.hot_path_bench import time_kernel  # noqa: E402


def step_rois(device):
    """(box rois [1024,5], levels), (mask rois, levels) of one training forward of the bench's harness."""
    import bench

    work = bench.TrainHarness(device, 0, 1, "f32", "eager")
    for _ in range(int(os.environ.get("STEPS_BEFORE", "12"))):   # the proposals of a trained-for-a-while RPN cluster more
        work.eager_step()
    with torch.no_grad():
        ret = work.net(work.data, work.im_info, roidb=work.roidb, rpn_targets=work.rpn_targets)
    b = ret["blobs"]
    out = (b["rois"].clone(), b["rois_levels"].clone(), b["mask_rois"].clone(), b["mask_rois_levels"].clone())
    del work
    torch.cuda.empty_cache()
    return out


def time_fpn(device, rois, idx, res, iters):
    """(forward us, backward us) of mi_roi_align_forward_fpn / _backward_fpn on seeded P2..P5 maps of 2 images: the C-ABI
    calls on preallocated buffers with the workspace the autograd Function allocates (records + backward plan), HIP events
    on the launch stream -- no allocator, no autograd bookkeeping in the timed region."""
    import ctypes

    from detectron_pytorch_amd import _lib
    from detectron_pytorch_amd.roi_align import _backward_workspace_bytes, _fpn_table

    lib = _lib.lib()
    maps = [torch.from_numpy(syn.feature_map(2, 256, syn.FPN_LEVELS[l][0], syn.FPN_LEVELS[l][1], seed=l)).to(device)
            for l in (5, 4, 3, 2)]
    grads = [torch.empty_like(m) for m in maps]
    scales = [syn.FPN_LEVELS[l][2] for l in (5, 4, 3, 2)]
    rois, idx = rois.contiguous(), idx.to(torch.int32).contiguous()
    r = int(rois.size(0))
    out = torch.empty((r, 256, res, res), device=device)
    g = torch.randn_like(out)
    ws_bytes = max(lib.mi_roi_align_forward_workspace_bytes(r),
                   _backward_workspace_bytes([(m.size(2), m.size(3)) for m in maps], 2, r))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    ftab, gtab = _fpn_table(maps, scales), _fpn_table(grads, scales, grads=True)
    stream = _lib.current_stream_handle(device)
    ready = True  # the fused forward leaves its records in the workspace
    flags = (_lib.ROI_ALIGN_RECORDS_READY if ready else 0) | _lib.ROI_ALIGN_OVERWRITE

    def fwd():
        _lib.check(lib.mi_roi_align_forward_fpn(ctypes.byref(ftab), rois.data_ptr(), idx.data_ptr(), out.data_ptr(), 2, 256, r,
                                                res, res, 2, _lib.LAYOUT_NCHW, ws.data_ptr(), ws_bytes, stream), "fwd")

    def bwd():
        _lib.check(lib.mi_roi_align_backward_fpn(ctypes.byref(gtab), g.data_ptr(), rois.data_ptr(), idx.data_ptr(), 2, 256, r,
                                                 res, res, 2, _lib.LAYOUT_NCHW, ws.data_ptr(), ws_bytes, flags, stream), "bwd")

    f = time_kernel(fwd, iters) * 1e6
    b = time_kernel(bwd, iters) * 1e6
    return round(f, 1), round(b, 1)


def tile_list_lengths(rois, lvls, res, th=16, tw=32):
    """RoIs per 16 x 32 gradient tile (what a backward workgroup walks), from the windows the samples touch."""
    r = rois.cpu().numpy()
    l = lvls.cpu().numpy()
    out = []
    for lvl in (2, 3, 4, 5):
        h, w, sc = syn.FPN_LEVELS[lvl]
        for n in (0, 1):
            sel = r[(l == lvl) & (r[:, 0] == n)]
            cnt = np.zeros(((h + th - 1) // th, (w + tw - 1) // tw), np.int32)
            for _, x1, y1, x2, y2 in sel:
                xs, ys = x1 * sc, y1 * sc
                xe, ye = xs + max(x2 * sc - xs, 1.0), ys + max(y2 * sc - ys, 1.0)
                a0, a1 = int(max(np.floor(xs), 0)) // tw, int(min(np.floor(xe) + 1, w - 1)) // tw
                b0, b1 = int(max(np.floor(ys), 0)) // th, int(min(np.floor(ye) + 1, h - 1)) // th
                cnt[b0:b1 + 1, a0:a1 + 1] += 1
            out.append(cnt.reshape(-1))
    c = np.concatenate(out)
    return {"tiles": int(c.size), "empty": int((c == 0).sum()), "mean_nonempty": round(float(c[c > 0].mean()), 1) if (c > 0).any() else 0,
            "p90": int(np.percentile(c, 90)), "p99": int(np.percentile(c, 99)), "max": int(c.max()),
            "sum": int(c.sum()), "over32": int((c > 32).sum()), "over64": int((c > 64).sum())}


def measure(dev, iters):
    rois, lvls, mrois, mlvls = step_rois(dev)
    res = {"tile_lists_box": tile_list_lengths(rois, lvls, 7), "tile_lists_mask": tile_list_lengths(mrois, mlvls, 14)}
    for name, r, l, size in (("box_head_1024_7x7", rois, lvls, 7), ("mask_head_256_14x14", mrois, mlvls, 14)):
        real = r[:, 0] >= 0
        idx = (5 - l).clamp(0, 3).to(torch.int32)
        f, b = time_fpn(dev, r.contiguous(), idx, size, iters)
        # the same number of RoIs, spread uniformly with FPN-consistent sizes
        ur, ul = syn.rois_fpn_distributed(int(r.size(0)), batch=2, seed=7)
        uidx = torch.from_numpy(np.array([(5, 4, 3, 2).index(int(x)) for x in ul], dtype=np.int32)).to(dev)
        uf, ub = time_fpn(dev, torch.from_numpy(ur).to(dev), uidx, size, iters)
        per_level = {int(k): int(((l == k) & real).sum()) for k in (2, 3, 4, 5)}
        res[name] = {"step_rois": {"fwd_us": f, "bwd_us": b, "real_rows": int(real.sum()), "per_level": per_level},
                     "uniform_rois": {"fwd_us": uf, "bwd_us": ub}}
    res["what"] = ("fused FPN RoIAlign (P2-P5, 2 images) on the RoIs a training step samples after 12 iterations on the fixed "
                   "batch (they cluster on the 8 gt boxes per image) next to as many uniformly spread RoIs; the C-ABI calls "
                   "on preallocated buffers, HIP events (kernel durations: profiles/)")
    return res


if __name__ == "__main__":
    print(json.dumps(measure(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 30)), flush=True)
