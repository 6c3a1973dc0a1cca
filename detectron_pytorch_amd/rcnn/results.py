"""Test-time result formats on the device: instance masks pasted into the image and
run-length encoded, keypoints decoded from heat maps.

Reference: lib/core/test.py:793-847 `segm_results` (per detection: zero-pad the M x M mask, cv2.resize to the expanded
box, binarise, paste, pycocotools.mask.encode) and :850-866 `keypoint_results` / lib/utils/keypoints.py:106-157
`heatmaps_to_keypoints` (per RoI: cv2.resize INTER_CUBIC of 17 heat maps, arg-max, softmax probability).  Both run on
the host there, one detection at a time, through OpenCV and pycocotools.  Here `mi_mask_paste_rle` produces the run
lengths of all detections in one launch without materialising any image, `mi_keypoint_decode` reduces every (RoI,
keypoint) map in one launch without storing the resized map; COCO's string form of the run lengths is written by the
same kernel (`rle_to_string` below is the host restatement of it, kept for callers that hold run lengths).

OpenCV and pycocotools are not part of this repository's environment: the kernels follow their published algorithms
(csrc/results.hip) and are tested against a CPU restatement of the same (oracle/results.py, parity unpinned).
"""
import numpy as np
import torch

from .. import _lib


def expand_boxes(boxes, scale):
    """lib/utils/boxes.py:233-249 for float32 boxes [R, 4] (device)."""
    w_half = (boxes[:, 2] - boxes[:, 0]) * .5
    h_half = (boxes[:, 3] - boxes[:, 1]) * .5
    x_c = (boxes[:, 2] + boxes[:, 0]) * .5
    y_c = (boxes[:, 3] + boxes[:, 1]) * .5
    w_half = w_half * scale
    h_half = h_half * scale
    return torch.stack([x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half], dim=1)


def rle_to_string(counts):
    """maskApi.c rleToString: run lengths (1-D integer array) -> COCO's compressed ASCII form.  Vectorised over the runs:
    each run is written as 5-bit groups, least significant first, bit 5 = "more groups follow", + 48."""
    x = np.asarray(counts, dtype=np.int64).copy()
    if x.size > 3:
        x[3:] -= np.asarray(counts, dtype=np.int64)[1:-2]
    n = x.size
    chars = np.zeros((n, 14), np.uint8)
    used = np.zeros((n, 14), bool)
    alive = np.ones(n, bool)
    level = 0
    while alive.any():
        c = x & 0x1f
        x = x >> 5                                              # arithmetic shift, as of a signed long
        more = np.where((c & 0x10) != 0, x != -1, x != 0)
        chars[:, level] = np.where(alive, (c | (more.astype(np.int64) << 5)) + 48, 0)
        used[:, level] = alive
        alive = alive & more
        level += 1
    return chars[used].tobytes().decode("ascii")                # row-major: run by run, group by group


def mask_rle_static(masks, boxes_int, im_h, im_w, thresh=0.5, capacity=1024, str_cap=4096):
    """One asynchronous `mi_mask_paste_rle` launch with fixed buffer sizes (what a hipGraph captures): returns device
    tensors (counts int32 [D, capacity], sizes int32 [2, D] = (number of runs, number of string bytes), strings uint8
    [D, str_cap]).  A row whose sizes exceed the capacities is incomplete; the caller checks `sizes` on the host."""
    _lib.require_cuda(masks, "masks")
    masks = masks.contiguous().float()
    boxes_int = boxes_int.contiguous().to(torch.int32)
    d, m = masks.size(0), masks.size(-1)
    dev = masks.device
    counts = torch.empty((d, capacity), dtype=torch.int32, device=dev)
    sizes = torch.empty((2, d), dtype=torch.int32, device=dev)
    strings = torch.empty((d, str_cap), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().mi_mask_paste_rle(masks.data_ptr(), boxes_int.data_ptr(), d, m, int(im_h), int(im_w), float(thresh),
                                          int(capacity), counts.data_ptr(), sizes[0].data_ptr(), int(str_cap),
                                          strings.data_ptr(), sizes[1].data_ptr(), _lib.current_stream_handle(dev))
    _lib.check(rc, "mi_mask_paste_rle")
    return counts, sizes, strings


def mask_rle(masks, boxes_int, im_h, im_w, thresh=0.5, capacity=1024, with_strings=True):
    """`mi_mask_paste_rle`: masks [D, M, M] float32, boxes_int [D, 4] int32 (device).  Returns (counts [D, cap] int64 on the
    host, num_counts [D], strings): strings[d] is COCO's compressed ASCII form of row d (None when `with_strings` is
    off).  One launch and one device-to-host copy; repeated with larger buffers only when a mask has more runs than
    `capacity` (the kernel reports the true sizes)."""
    _lib.require_cuda(masks, "masks")
    masks = masks.contiguous().float()
    boxes_int = boxes_int.contiguous().to(torch.int32)
    d, m = masks.size(0), masks.size(-1)
    dev = masks.device
    str_cap = 4 * capacity
    while True:
        counts = torch.empty((d, capacity), dtype=torch.int32, device=dev)   # uint32 run lengths
        sizes = torch.empty((2, d), dtype=torch.int32, device=dev)           # number of runs, number of string bytes
        strings = torch.empty((d, str_cap), dtype=torch.uint8, device=dev) if with_strings else None
        with torch.cuda.device(dev):
            rc = _lib.lib().mi_mask_paste_rle(masks.data_ptr(), boxes_int.data_ptr(), d, m, int(im_h), int(im_w),
                                              float(thresh), int(capacity), counts.data_ptr(), sizes[0].data_ptr(),
                                              int(str_cap), strings.data_ptr() if with_strings else None,
                                              sizes[1].data_ptr() if with_strings else None,
                                              _lib.current_stream_handle(dev))
        _lib.check(rc, "mi_mask_paste_rle")
        sizes_h = sizes.cpu().numpy()
        need = int(sizes_h[0].max()) if d else 0
        need_bytes = int(sizes_h[1].max()) if (d and with_strings and need <= capacity) else 0
        if need <= capacity and need_bytes <= str_cap:
            out_strings = None
            if with_strings:
                raw = strings.cpu().numpy()
                out_strings = [raw[i, :sizes_h[1, i]].tobytes().decode("ascii") for i in range(d)]
            return counts.cpu().numpy().view(np.uint32).astype(np.int64), sizes_h[0], out_strings
        if need > capacity:
            capacity = 1 << int(np.ceil(np.log2(need)))
            str_cap = max(str_cap, 4 * capacity)
        else:
            str_cap = 1 << int(np.ceil(np.log2(need_bytes)))


def mask_rle_counts(masks, boxes_int, im_h, im_w, thresh=0.5, capacity=1024):
    """Run lengths only: (counts [D, cap], num_counts [D])."""
    counts, num, _ = mask_rle(masks, boxes_int, im_h, im_w, thresh, capacity, with_strings=False)
    return counts, num


def segm_results(cls_boxes, masks, ref_boxes, im_h, im_w, cfg):
    """lib/core/test.py:793-847.  cls_boxes: per class [k_j, 5] (only the lengths are used, as there); masks [R, K, M, M]
    device tensor in the class-major row order of cls_boxes; ref_boxes [R, 4] device tensor.  Returns cls_segms: per class
    a list of {'size': [h, w], 'counts': str}."""
    num_classes = cfg.MODEL.NUM_CLASSES
    cls_segms = [[] for _ in range(num_classes)]
    lengths = [0] + [int(len(cls_boxes[j])) for j in range(1, num_classes)]
    r = sum(lengths)
    assert r == masks.size(0), "masks and cls_boxes disagree (test.py:846)"
    if r == 0:
        return cls_segms
    m = cfg.MRCNN.RESOLUTION
    cls_of_row = torch.repeat_interleave(torch.arange(num_classes), torch.tensor(lengths)).to(masks.device)
    rows = torch.arange(r, device=masks.device)
    sel = masks[rows, cls_of_row] if cfg.MRCNN.CLS_SPECIFIC_MASK else masks[:, 0]
    boxes_int = expand_boxes(ref_boxes.float(), (m + 2.0) / m).to(torch.int32)        # :803-805 (truncation)
    _, _, strings = mask_rle(sel, boxes_int, im_h, im_w, cfg.MRCNN.THRESH_BINARIZE)
    ind = 0
    for j in range(1, num_classes):
        for _ in range(lengths[j]):
            cls_segms[j].append({"size": [int(im_h), int(im_w)], "counts": strings[ind]})
            ind += 1
    return cls_segms


def heatmaps_to_keypoints(maps, rois, min_size=0):
    """lib/utils/keypoints.py:106-157 on the device: maps [R, K, H, H] float32 logits, rois [R, 4] -> xy_preds [R, 4, K]
    (x, y, logit, probability)."""
    _lib.require_cuda(maps, "maps")
    maps = maps.contiguous().float()
    rois = rois.contiguous().float()
    r, k, h, _ = maps.shape
    out = torch.empty((r, 4, k), dtype=torch.float32, device=maps.device)
    with torch.cuda.device(maps.device):
        rc = _lib.lib().mi_keypoint_decode(maps.data_ptr(), rois.data_ptr(), r, k, h, int(min_size), out.data_ptr(),
                                           _lib.current_stream_handle(maps.device))
    _lib.check(rc, "mi_keypoint_decode")
    return out


def nms_oks(kp_predictions, rois, thresh):
    """lib/utils/keypoints.py:225-240 on the device: kp_predictions [R, 4, 17] (heatmaps_to_keypoints), rois [R, 4] ->
    int64 tensor of the kept rows, best mean keypoint logit first (one host copy: their number)."""
    _lib.require_cuda(kp_predictions, "kp_predictions")
    kp = kp_predictions.contiguous().float()
    rois = rois.contiguous().float()
    r = int(kp.size(0))
    keep = torch.empty((r,), dtype=torch.int64, device=kp.device)
    num = torch.zeros((1,), dtype=torch.int32, device=kp.device)
    with torch.cuda.device(kp.device):
        rc = _lib.lib().mi_keypoint_nms_oks(kp.data_ptr(), rois.data_ptr(), r, int(kp.size(2)), float(thresh), keep.data_ptr(),
                                            num.data_ptr(), _lib.current_stream_handle(kp.device))
    _lib.check(rc, "mi_keypoint_nms_oks")
    return keep[:int(num.item())]


def keypoint_results(cls_boxes, pred_heatmaps, ref_boxes, cfg, person_idx=1):
    """lib/core/test.py:850-866 (`person_idx`: datasets' 'person' class, 1 for COCO).  Returns cls_keyps: per class a list
    of [4, K] arrays (device tensors).  With cfg.KRCNN.NMS_OKS the person detections are thinned by OKS-NMS (:857-862; like
    the reference, cls_boxes[person_idx] is replaced in place)."""
    cls_keyps = [[] for _ in range(cfg.MODEL.NUM_CLASSES)]
    xy_preds = heatmaps_to_keypoints(pred_heatmaps, ref_boxes, cfg.KRCNN.INFERENCE_MIN_SIZE)
    if cfg.KRCNN.NMS_OKS:
        keep = nms_oks(xy_preds, ref_boxes, 0.3)                                   # :858, the 0.3 is the reference's literal
        xy_preds = xy_preds[keep]
        boxes = cls_boxes[person_idx]
        cls_boxes[person_idx] = boxes[keep.cpu().numpy()] if isinstance(boxes, np.ndarray) else boxes[keep.to(boxes.device)]
    cls_keyps[person_idx] = [xy_preds[i] for i in range(xy_preds.size(0))]
    return cls_keyps


# ---- detections.pkl (lib/core/test_engine.py:300-313, :369-397) ------------------------------------------------------
def empty_results(num_classes, num_images):
    """test_engine.py:369-388: all_boxes[cls][image] = [N, 5]; all_segms[cls][image] = list of COCO RLEs; all_keyps[cls]
    [image] = list of [4, K] arrays -- each in 1:1 correspondence with the boxes."""
    def make():
        return [[[] for _ in range(num_images)] for _ in range(num_classes)]
    return make(), make(), make()


def extend_results(index, all_res, im_res):
    """test_engine.py:391-397: file one image's per-class results (background skipped)."""
    for cls_idx in range(1, len(im_res)):
        all_res[cls_idx][index] = im_res[cls_idx]


def _to_host(x):
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    if isinstance(x, (list, tuple)):
        return [_to_host(v) for v in x]
    return x


def save_detections(path, all_boxes, all_segms, all_keyps, cfg_yaml=""):
    """The `detections.pkl` of test_engine.py:300-313: {'all_boxes', 'all_segms', 'all_keyps', 'cfg'} pickled with the
    highest protocol (utils/io.py:39-43).  Device tensors are brought to the host as numpy arrays (what the reference's
    numpy pipeline holds at this point); `cfg_yaml` is the yaml dump of the configuration."""
    import pickle

    obj = dict(all_boxes=_to_host(all_boxes), all_segms=_to_host(all_segms), all_keyps=_to_host(all_keyps), cfg=cfg_yaml)
    with open(path, "wb") as fp:
        pickle.dump(obj, fp, pickle.HIGHEST_PROTOCOL)
    return path
