"""NMS / IoU -- host-side mirror of the reference's NMS entry points, backed by mi_nms / mi_bbox_overlaps.

Reference interfaces kept:
  * utils.cython_nms.nms(dets f32[n,5] numpy, thresh) -> int64 ascending ORIGINAL indices
        (lib/utils/cython_nms.pyx:37-87; the NMS the model really runs, via utils/boxes.py:320-324)
  * model.nms.nms_gpu.nms_gpu(dets cuda[n,5] sorted, thresh) -> int32 [k,1] positions
        (lib/model/nms/nms_gpu.py:7-12) and model.nms.nms_wrapper.nms (nms_wrapper.py:11-18)
  * utils.cython_bbox.bbox_overlaps(boxes f32[N,4], query f32[K,4]) -> f32[N,K]
        (lib/utils/cython_bbox.pyx:32-73)
New, device-native: `nms_device(dets_cuda, thresh, mode)` returns (keep, num_keep) as device tensors
without any host synchronisation, for callers that stay on the GPU (proposal generation).
"""
import numpy as np
import torch

from . import _lib


def nms_device(dets, thresh, mode=_lib.NMS_GE_ORIG_ASC):
    """dets: cuda float32 [n,5].  Returns (keep, num_keep): keep is int64[n] (GE_ORIG_ASC) or int32[n]
    (GT_SORTED_POS) with only the first num_keep[0] entries defined; num_keep is int32[1] on device."""
    _lib.require_cuda(dets, "dets")
    if dets.dtype != torch.float32 or dets.dim() != 2 or dets.size(1) != 5:
        raise ValueError("dets must be float32 [n, 5] (x1, y1, x2, y2, score)")
    dets = dets.contiguous()
    n = dets.size(0)
    dev = dets.device
    keep_dtype = torch.int64 if mode == _lib.NMS_GE_ORIG_ASC else torch.int32
    keep = torch.empty((max(n, 1),), dtype=keep_dtype, device=dev)
    num_keep = torch.empty((1,), dtype=torch.int32, device=dev)
    ws_bytes = _lib.lib().mi_nms_workspace_bytes(n)
    workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mi_nms(dets.data_ptr(), n, float(thresh), int(mode), keep.data_ptr(), num_keep.data_ptr(),
                               workspace.data_ptr(), ws_bytes, _lib.current_stream_handle(dev))
    _lib.check(rc, "mi_nms")
    return keep, num_keep


_STREAM_POOL = {}
_BATCH_MAX_BOXES = 4096


def nms_device_many(dets_list, thresh, mode=_lib.NMS_GE_ORIG_ASC, max_streams=8):
    """Independent NMS problems (the RPN runs one per FPN level and image: modeling/generate_proposals.py:91-99,161).
    One problem's greedy reduce occupies a single CU and its launches are a dependent chain, so ten problems issued back
    to back leave a 256-CU chip idle.  Problems of up to 4096 boxes go through mi_nms_batched (one launch per stage for
    all of them, one host call); anything larger is fanned out over side HIP streams through mi_nms.
    Returns [(keep, num_keep), ...] with the nms_device() conventions, usable on the caller's current stream."""
    import ctypes

    if not dets_list:
        return []
    dev = dets_list[0].device
    for d in dets_list:
        _lib.require_cuda(d, "dets")
        if d.dtype != torch.float32 or d.dim() != 2 or d.size(1) != 5 or d.device != dev:
            raise ValueError("every dets must be float32 [n, 5] on the same device")
    if all(d.size(0) <= _BATCH_MAX_BOXES for d in dets_list):
        lib = _lib.lib()
        p = len(dets_list)
        dets_list = [d.contiguous() for d in dets_list]
        ns = [int(d.size(0)) for d in dets_list]
        keep_dtype = torch.int64 if mode == _lib.NMS_GE_ORIG_ASC else torch.int32
        offs = [0]
        for n in ns:
            offs.append(offs[-1] + max(n, 1))
        keep_all = torch.empty((offs[-1],), dtype=keep_dtype, device=dev)
        num_all = torch.empty((p,), dtype=torch.int32, device=dev)
        n_arr = (ctypes.c_int * p)(*ns)
        ws_bytes = lib.mi_nms_batched_workspace_bytes(p, n_arr)
        workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        item = keep_all.element_size()
        dets_arr = (ctypes.c_void_p * p)(*[d.data_ptr() for d in dets_list])
        keep_arr = (ctypes.c_void_p * p)(*[keep_all.data_ptr() + offs[i] * item for i in range(p)])
        num_arr = (ctypes.c_void_p * p)(*[num_all.data_ptr() + 4 * i for i in range(p)])
        with torch.cuda.device(dev):
            rc = lib.mi_nms_batched(p, dets_arr, n_arr, float(thresh), int(mode), keep_arr, num_arr, workspace.data_ptr(),
                                    ws_bytes, _lib.current_stream_handle(dev))
        _lib.check(rc, "mi_nms_batched")
        return [(keep_all[offs[i]:offs[i + 1]], num_all[i:i + 1]) for i in range(p)]
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, max_streams)
    if key not in _STREAM_POOL:
        _STREAM_POOL[key] = [torch.cuda.Stream(device=dev) for _ in range(max_streams)]
    pool = _STREAM_POOL[key]
    ready = torch.cuda.Event()
    ready.record(cur)
    outs, used = [], []
    for i, dets in enumerate(dets_list):
        side = pool[i % len(pool)]
        if i < len(pool):
            side.wait_event(ready)  # inputs produced on the caller's stream
            used.append(side)
        with torch.cuda.stream(side):
            keep, num_keep = nms_device(dets, thresh, mode)
        keep.record_stream(cur)
        num_keep.record_stream(cur)
        outs.append((keep, num_keep))
    for side in used:
        cur.wait_stream(side)
    return outs


def nms_segmented(scores, boxes, score_thresh, nms_thresh, first_class=1, with_masked_scores=False):
    """The per-class loop of core/test.py:748-771 as one asynchronous call on the blobs as the network leaves them:
    scores [R, C], boxes [R, 4C] float32 device tensors.  Class j >= first_class is a segment; its rows with
    score > score_thresh go through cython-semantics NMS.  Returns (kept int32 [C - first_class, R] 0/1 flags, num_keep
    int32 [C - first_class]) -- device tensors, no host synchronisation, shapes fixed by the inputs' shapes alone;
    `with_masked_scores`: a third tensor float32 [C - first_class, R], the score where the row survives, -inf elsewhere."""
    _lib.require_cuda(scores, "scores")
    if scores.dtype != torch.float32 or boxes.dtype != torch.float32 or scores.dim() != 2:
        raise TypeError("nms_segmented expects float32 scores [R, C] and boxes [R, 4C]")
    r, c = scores.shape
    if boxes.shape != (r, 4 * c):
        raise ValueError("boxes must be [R, 4 * num_classes]")
    scores, boxes = scores.contiguous(), boxes.contiguous()
    nseg = c - first_class
    dev = scores.device
    kept = torch.empty((max(nseg, 0), r), dtype=torch.int32, device=dev)
    num_keep = torch.empty((max(nseg, 0),), dtype=torch.int32, device=dev)
    masked = torch.empty((max(nseg, 0), r), dtype=torch.float32, device=dev) if with_masked_scores else None
    if nseg <= 0 or r == 0:
        num_keep.zero_()
        return (kept, num_keep, masked) if with_masked_scores else (kept, num_keep)
    lib = _lib.lib()
    ws_bytes = lib.mi_nms_segmented_workspace_bytes(nseg, r)
    workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.mi_nms_segmented(boxes.data_ptr() + 16 * first_class, 4, 4 * c, scores.data_ptr() + 4 * first_class, 1, c,
                                  nseg, r, float(score_thresh), float(nms_thresh), kept.data_ptr(), num_keep.data_ptr(),
                                  masked.data_ptr() if with_masked_scores else None, workspace.data_ptr(), ws_bytes,
                                  _lib.current_stream_handle(dev))
    _lib.check(rc, "mi_nms_segmented")
    return (kept, num_keep, masked) if with_masked_scores else (kept, num_keep)


def nms_gpu(dets, thresh):
    """model.nms.nms_gpu.nms_gpu: dets pre-sorted by descending score; returns int32 [k, 1] (one host
    sync to slice, exactly where the reference has `keep[:num_out[0]]`, nms_gpu.py:11)."""
    keep, num_keep = nms_device(dets, thresh, _lib.NMS_GT_SORTED_POS)
    return keep[:int(num_keep.item())].view(-1, 1)


def nms(dets, thresh, force_cpu=False):
    """model.nms.nms_wrapper.nms (nms_wrapper.py:11-18).  `force_cpu` is accepted and ignored, as there."""
    if dets.shape[0] == 0:
        return []
    return nms_gpu(dets, thresh)


def cython_nms(dets, thresh, device=None):
    """utils.cython_nms.nms replacement: numpy (or tensor) in, numpy int64 ascending original indices out.
    Bit-exact with the reference for inputs without tied scores (tie rule: higher index first)."""
    if isinstance(dets, np.ndarray):
        if dets.shape[0] == 0:
            return np.zeros((0,), dtype=np.int64)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        t = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32)).to(dev)
        keep, num_keep = nms_device(t, thresh, _lib.NMS_GE_ORIG_ASC)
        return keep[:int(num_keep.item())].cpu().numpy()
    keep, num_keep = nms_device(dets, thresh, _lib.NMS_GE_ORIG_ASC)
    return keep[:int(num_keep.item())]


def bbox_overlaps(boxes, query_boxes, device=None):
    """utils.cython_bbox.bbox_overlaps replacement (numpy in -> numpy out; tensors in -> tensor out)."""
    as_numpy = isinstance(boxes, np.ndarray)
    if as_numpy:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        boxes = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).to(dev)
        query_boxes = torch.from_numpy(np.ascontiguousarray(query_boxes, dtype=np.float32)).to(dev)
    _lib.require_cuda(boxes, "boxes")
    boxes = boxes.contiguous()
    query_boxes = query_boxes.contiguous()
    if boxes.dtype != torch.float32 or query_boxes.dtype != torch.float32:
        raise TypeError("bbox_overlaps supports float32 only")
    n, k = boxes.size(0), query_boxes.size(0)
    out = torch.empty((n, k), dtype=torch.float32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        rc = _lib.lib().mi_bbox_overlaps(boxes.data_ptr(), n, query_boxes.data_ptr(), k, out.data_ptr(),
                                         _lib.current_stream_handle(boxes.device))
    _lib.check(rc, "mi_bbox_overlaps")
    return out.cpu().numpy() if as_numpy else out


SOFT_NMS_METHODS = {"hard": 0, "linear": 1, "gaussian": 2}  # utils/boxes.py:334


def soft_nms_device(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method=1):
    """On-device Soft-NMS, no host synchronisation: returns (out_dets [n,5], out_inds int64 [n], num_out int32 [1]);
    the first num_out rows are the reference's boxes[:N] / inds[:N]."""
    _lib.require_cuda(dets, "dets")
    if dets.dtype != torch.float32 or dets.dim() != 2 or dets.size(1) != 5:
        raise TypeError("soft_nms expects float32 dets [n, 5]")
    dets = dets.contiguous()
    n = dets.size(0)
    out_dets = torch.empty((n, 5), dtype=torch.float32, device=dets.device)
    out_inds = torch.empty((n,), dtype=torch.int64, device=dets.device)
    num_out = torch.empty((1,), dtype=torch.int32, device=dets.device)
    with torch.cuda.device(dets.device):
        rc = _lib.lib().mi_soft_nms(dets.data_ptr(), n, float(sigma), float(overlap_thresh), float(score_thresh),
                                    int(method), out_dets.data_ptr(), out_inds.data_ptr(), num_out.data_ptr(),
                                    _lib.current_stream_handle(dets.device))
    _lib.check(rc, "mi_soft_nms")
    return out_dets, out_inds, num_out


def soft_nms(boxes_in, sigma=0.5, Nt=0.3, threshold=0.001, method=0, device=None):
    """utils.cython_nms.soft_nms replacement (lib/utils/cython_nms.pyx:98-203): numpy float32 [n,5] in ->
    (boxes[:N] float32 [N,5], inds[:N] int64) out, same rows in the same order as the reference; tensors in ->
    tensors out.  `method`: 0 hard, 1 linear, 2 gaussian (what utils/boxes.py:337-343 passes)."""
    as_numpy = isinstance(boxes_in, np.ndarray)
    if as_numpy:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        boxes_in = torch.from_numpy(np.ascontiguousarray(boxes_in, dtype=np.float32)).to(dev)
    out_dets, out_inds, num_out = soft_nms_device(boxes_in, sigma, Nt, threshold, int(method))
    k = int(num_out.item())
    if as_numpy:
        return out_dets[:k].cpu().numpy(), out_inds[:k].cpu().numpy()
    return out_dets[:k], out_inds[:k]


def box_utils_soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method="linear"):
    """utils.boxes.soft_nms (lib/utils/boxes.py:327-344): the named-method wrapper the test-time code calls."""
    if dets.shape[0] == 0:
        return dets, []
    if method not in SOFT_NMS_METHODS:
        raise AssertionError("Unknown soft_nms method: {}".format(method))
    return soft_nms(dets, sigma, overlap_thresh, score_thresh, SOFT_NMS_METHODS[method])
