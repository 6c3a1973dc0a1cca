"""RoIFeatureTransform: the caller of the RoI operators.

Mirrors `Generalized_RCNN.roi_feature_transform` (lib/modeling/model_builder.py:252-324) as a free function, plus
the two numpy helpers that build its inputs (lib/utils/fpn.py:11-28 `map_rois_to_fpn_levels`,
lib/utils/fpn.py:31-58 `add_multilevel_roi_blobs`). The reference reads k_min/k_max/grid_size from its global cfg;
here they are arguments with the reference's defaults (FPN.ROI_MIN_LEVEL=2, FPN.ROI_MAX_LEVEL=5,
POOLING_SIZE*2 for RoICrop).

Everything numeric runs in the HIP operators; this file is dispatch, concatenation and the restore permutation.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .roi_align import RoIAlignFunction, roi_align_fpn, roi_align_fpn_supported
from .roi_pool import RoIPoolFunction
from .roi_crop import RoICropFunction

METHODS = ("RoIPoolF", "RoICrop", "RoIAlign")


def map_rois_to_fpn_levels(rois_xyxy, k_min=2, k_max=5, s0=224.0, lvl0=4):
    """FPN paper eq. 1 as the reference evaluates it (lib/utils/fpn.py:11-28): areas use the +1 convention,
    level = clip(floor(lvl0 + log2(sqrt(area) / s0 + 1e-6)), k_min, k_max). numpy in, int64 numpy out."""
    rois_xyxy = np.asarray(rois_xyxy)
    w = rois_xyxy[:, 2] - rois_xyxy[:, 0] + 1
    h = rois_xyxy[:, 3] - rois_xyxy[:, 1] + 1
    side = np.sqrt(np.maximum(w * h, 0))
    lvls = np.floor(lvl0 + np.log2(side / s0 + 1e-6))
    return np.clip(lvls, k_min, k_max).astype(np.int64)


def add_multilevel_roi_blobs(blobs, blob_prefix, rois, target_lvls, lvl_min, lvl_max):
    """lib/utils/fpn.py:31-58: split `rois` [R,5] per level into blobs[prefix+'_fpn<l>'] (level order, original
    order inside a level) and store the permutation that undoes the level-major concatenation in
    blobs[prefix+'_idx_restore_int32']."""
    order = []
    for lvl in range(lvl_min, lvl_max + 1):
        idx = np.nonzero(target_lvls == lvl)[0]
        blobs["%s_fpn%d" % (blob_prefix, lvl)] = rois[idx, :]
        order.append(idx)
    order = np.concatenate(order) if order else np.empty((0,), dtype=np.int64)
    restore = np.argsort(order, kind="stable").astype(np.int32)
    blobs[blob_prefix + "_idx_restore_int32"] = restore
    return blobs


def affine_grid_gen(rois, input_size, grid_size, stride=16.0):
    """lib/utils/net.py:110-132: the affine sampling grid of each RoI on a stride-16 feature map, (x, y) last."""
    rois = rois.detach()
    x1, y1, x2, y2 = (rois[:, i:i + 1] / stride for i in (1, 2, 3, 4))
    height, width = int(input_size[0]), int(input_size[1])
    zero = torch.zeros_like(x1)
    theta = torch.cat([(x2 - x1) / (width - 1), zero, (x1 + x2 - width + 1) / (width - 1),
                       zero, (y2 - y1) / (height - 1), (y1 + y2 - height + 1) / (height - 1)], 1).view(-1, 2, 3)
    return F.affine_grid(theta, torch.Size((rois.size(0), 1, grid_size, grid_size)), align_corners=True)


def _as_device_rois(rois, device):
    if isinstance(rois, np.ndarray):
        rois = torch.from_numpy(np.ascontiguousarray(rois, dtype=np.float32))
    return rois.to(device=device, dtype=torch.float32, non_blocking=True)


def _one_level(features, rois, method, resolution, scale, sampling_ratio, grid_size, crop_max_pool):
    if method == "RoIPoolF":
        return RoIPoolFunction(resolution, resolution, scale)(features, rois)
    if method == "RoIAlign":
        return RoIAlignFunction(resolution, resolution, scale, sampling_ratio)(features, rois)
    grid_xy = affine_grid_gen(rois, features.shape[2:], grid_size)
    grid_yx = torch.stack([grid_xy[..., 1], grid_xy[..., 0]], 3).contiguous()
    out = RoICropFunction()(features, grid_yx.detach())
    return F.max_pool2d(out, 2, 2) if crop_max_pool else out


def roi_feature_transform(blobs_in, rpn_ret, blob_rois="rois", method="RoIPoolF", resolution=7,
                          spatial_scale=1.0 / 16.0, sampling_ratio=0, k_min=2, k_max=5, grid_size=14,
                          crop_resize_with_max_pool=True, fused=True):
    """model_builder.py:252-324. `blobs_in`: one [N,C,H,W] device tensor, or the FPN list ordered coarsest level
    first (then `spatial_scale` is a list in the same order). `rpn_ret[blob_rois]` / `rpn_ret[blob_rois+'_fpn<l>']`
    hold [R,5] rois (numpy, as the reference's data layer produces them, or device tensors). Levels without rois are
    skipped; the level-major result is put back in dataloader order with `<blob_rois>_idx_restore_int32`.
    RoIAlign over NCHW maps takes the FPN-fused entry point (one call for all levels, no concat / restore gather of the
    pooled features) unless `fused=False`."""
    if method not in METHODS:
        raise AssertionError("Unknown pooling method: {}".format(method))
    if not isinstance(blobs_in, (list, tuple)):
        rois = _as_device_rois(rpn_ret[blob_rois], blobs_in.device)
        return _one_level(blobs_in, rois, method, resolution, spatial_scale, sampling_ratio, grid_size,
                          crop_resize_with_max_pool)
    if len(blobs_in) != k_max - k_min + 1:
        raise AssertionError("expected %d FPN levels, got %d" % (k_max - k_min + 1, len(blobs_in)))
    levels = rpn_ret.get(blob_rois + "_levels") if hasattr(rpn_ret, "get") else None
    if levels is not None and method == "RoIAlign" and fused:
        # device-side producers (fpn_proposals.distribute, rcnn.targets.label_proposals) hand over the un-split RoIs and
        # the FPN level of each: nothing to split, concatenate or restore -- one fused call, output in the order of the rois
        rois = _as_device_rois(rpn_ret[blob_rois], blobs_in[0].device)
        if roi_align_fpn_supported(list(blobs_in), rois.size(0), resolution, resolution):
            # `<blob>_records`: the producer of the blob already wrote its RoIAlign records (roi_align.PreparedRecords)
            return roi_align_fpn(list(blobs_in), list(spatial_scale), rois, (k_max - levels).to(torch.int32), resolution,
                                 resolution, sampling_ratio, prepared=rpn_ret.get(blob_rois + "_records"))
    if levels is not None and ("%s_fpn%d" % (blob_rois, k_min)) not in rpn_ret:
        # the device-side producers emit only `<blob>` + `<blob>_levels`; when the fused call is not available (RoIPool /
        # RoICrop, MI_ROI_ALIGN_IMPL=direct, MI_ROI_ALIGN_NO_WS, > 8192 RoIs, FPN.DIM not a multiple of 32, fused=False)
        # the levels are split here: one operator call per level, results scattered back to the order of the rois.  Rows
        # whose level is outside [k_min, k_max] (padding) stay zero.  (The row counts come to the host: a fallback.)
        rois = _as_device_rois(rpn_ret[blob_rois], blobs_in[0].device)
        out = None
        for lvl in range(k_min, k_max + 1):
            idx = torch.nonzero(levels == lvl, as_tuple=False).flatten()
            if idx.numel() == 0:
                continue
            y = _one_level(blobs_in[k_max - lvl], rois.index_select(0, idx).contiguous(), method, resolution,
                           spatial_scale[k_max - lvl], sampling_ratio, grid_size, crop_resize_with_max_pool)
            if out is None:
                out = y.new_zeros((rois.size(0),) + tuple(y.shape[1:]))
            out = out.index_copy(0, idx, y)
        if out is None:  # no RoI on any level: the shape still has to be right
            c = blobs_in[0].size(1)
            # (RoICrop samples a grid_size x grid_size grid, halved by the max-pool: _one_level's shapes)
            res = resolution if method != "RoICrop" else (grid_size // 2 if crop_resize_with_max_pool else grid_size)
            out = blobs_in[0].new_zeros((rois.size(0), c, res, res))
        return out
    level_rois = [rpn_ret["%s_fpn%d" % (blob_rois, lvl)] for lvl in range(k_min, k_max + 1)]
    restore = rpn_ret[blob_rois + "_idx_restore_int32"]
    device = blobs_in[0].device
    if isinstance(restore, np.ndarray):
        restore = torch.from_numpy(np.ascontiguousarray(restore))
    restore = restore.to(device=device, non_blocking=True)
    if restore.dtype not in (torch.int32, torch.int64):
        restore = restore.to(torch.int64)
    num_rois = sum(len(x) for x in level_rois)
    if fused and method == "RoIAlign" and roi_align_fpn_supported(list(blobs_in), num_rois, resolution, resolution):
        # one call for the whole pyramid: the RoIs go back to dataloader order BEFORE pooling (a [R,5] gather instead
        # of the [R,C,res,res] one at :306), each with the index of its map in blobs_in (coarsest level first).
        # rpn_ret[blob_rois] -- the un-split blob the reference's data layer also provides -- already is that order.
        # The map index of every row comes from the restore index and the levels' row counts (host-side sizes) in ONE
        # launch (mi_fpn_level_index_from_restore; until round 6: four slice fills, a dtype conversion and a gather).
        import ctypes

        counts = [len(x) for x in level_rois]
        map_index = [k_max - lvl for lvl in range(k_min, k_max + 1)]
        lvl_of_roi = torch.empty((num_rois,), dtype=torch.int32, device=device)
        restore = restore.contiguous()
        with torch.cuda.device(device):
            rc = _lib.lib().mi_fpn_level_index_from_restore(
                restore.data_ptr(), 1 if restore.dtype == torch.int64 else 0, num_rois, len(counts),
                (ctypes.c_int * len(counts))(*counts), (ctypes.c_int * len(counts))(*map_index), lvl_of_roi.data_ptr(),
                _lib.current_stream_handle(device))
        _lib.check(rc, "mi_fpn_level_index_from_restore")
        if blob_rois in rpn_ret and len(rpn_ret[blob_rois]) == num_rois:
            rois = _as_device_rois(rpn_ret[blob_rois], device)
        else:
            rois = torch.cat([_as_device_rois(x, device) for x in level_rois if len(x)], dim=0)[restore.long()]
        return roi_align_fpn(list(blobs_in), list(spatial_scale), rois, lvl_of_roi, resolution, resolution,
                             sampling_ratio)
    restore = restore.long()
    pooled = []
    for lvl, rois_l in zip(range(k_min, k_max + 1), level_rois):
        features = blobs_in[k_max - lvl]
        if len(rois_l) == 0:
            continue
        rois = _as_device_rois(rois_l, features.device)
        pooled.append(_one_level(features, rois, method, resolution, spatial_scale[k_max - lvl], sampling_ratio,
                                 grid_size, crop_resize_with_max_pool))
    return torch.cat(pooled, dim=0)[restore]
