"""CPU restatement of the test-time RESULT FORMATS: instance masks pasted into the image and
run-length encoded, keypoints decoded from heat maps.  Test infrastructure: only tests/ may import this.

PARITY PARTLY PINNED: the two resize kernels are cross-checked against an independent implementation of the same published
algorithm (torch.nn.functional.interpolate(align_corners=False), tests/test_results_cpu.py, <= 1e-4 of the data range);
the RLE encoder has hand vectors and a round trip only.  The reference delegates the numerical work to two third-party packages that are not installed here and
cannot be fetched (no network), so this file restates their PUBLISHED algorithms instead of executing them:
  * OpenCV (opencv-python; the reference pins no version, `import cv2` in lib/core/test.py:36, lib/utils/keypoints.py:27):
    `cv2.resize` for float32 input, INTER_LINEAR (default) and INTER_CUBIC -- modules/imgproc/src/resize.cpp:
    resizeGeneric_ (coordinate mapping fx = (dx + 0.5) * scale - 0.5, cvFloor, border handling), HResizeLinear / VResizeLinear,
    interpolateCubic (A = -0.75), HResizeCubic / VResizeCubic (replicated border).  The scalar code paths are restated;
    OpenCV's SIMD paths may contract multiply-adds, so last-bit differences against a real cv2 are possible.
    (cv2.resize's shortcut to INTER_AREA for an exact 2x down-scale is mathematically the same bilinear value and is not
    special-cased.)
  * pycocotools 2.0 (`pycocotools.mask.encode`, lib/core/test.py:838): common/maskApi.c rleEncode (column-major run
    lengths, first run counts zeros) and rleToString (differences to the run two back, 5 bits per character + 48).
What IS pinned to the reference: the call sites and everything around the two packages -- `segm_results`
(lib/core/test.py:793-847), `expand_boxes` (lib/utils/boxes.py:233-249), `heatmaps_to_keypoints` / `scores_to_probs`
(lib/utils/keypoints.py:106-157, :214-222) -- restated line by line below and checked, in tests/test_results_cpu.py,
against the reference's OWN source text of those functions executed with only `cv2.resize` and `mask_util.encode` bound
to the restatements of this file (exact equality of every RLE string and every keypoint value).
"""
import numpy as np


# ---- cv2.resize, float32, scalar paths -----------------------------------------------------------------------------
def _axis_linear(dst, src):
    """resize.cpp resizeGeneric_ (ksize 2): source index and weight pair of every destination coordinate."""
    scale = 1.0 / (float(dst) / float(src))                       # double, as `scale_x = 1. / inv_scale_x`
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    low = s < 0
    f[low], s[low] = 0.0, 0
    high = s >= src - 1
    f[high], s[high] = 0.0, src - 1
    return s, (np.float32(1.0) - f).astype(np.float32), f


def cv2_resize_linear(src, width, height):
    """cv2.resize(src float32 [h, w] or [h, w, c], (width, height)) with INTER_LINEAR."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    h, w, _ = src.shape
    sx, a0, a1 = _axis_linear(width, w)
    sx1 = np.minimum(sx + 1, w - 1)                                # never read with a non-zero weight beyond the edge
    # horizontal pass (HResizeLinear): D[dx] = S[sx] * a0 + S[sx + 1] * a1, fp32 products and sum
    rows = (src[:, sx, :] * a0[None, :, None]).astype(np.float32) + (src[:, sx1, :] * a1[None, :, None]).astype(np.float32)
    rows = rows.astype(np.float32)
    # vertical pass: rows sy and sy + 1 clipped to the image, weights (1 - fy, fy) NOT reset at the border
    scale = 1.0 / (float(height) / float(h))
    d = np.arange(height, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sy = np.floor(f).astype(np.int64)
    f = (f - sy.astype(np.float32)).astype(np.float32)
    b0, b1 = (np.float32(1.0) - f).astype(np.float32), f
    r0, r1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    out = (rows[r0] * b0[:, None, None]).astype(np.float32) + (rows[r1] * b1[:, None, None]).astype(np.float32)
    out = out.astype(np.float32)
    return out[:, :, 0] if squeeze else out


def _cubic_coeffs(x):
    """resize.cpp interpolateCubic, float arithmetic."""
    x = x.astype(np.float32)
    a = np.float32(-0.75)
    one = np.float32(1.0)
    c0 = ((a * (x + one) - np.float32(5) * a) * (x + one) + np.float32(8) * a) * (x + one) - np.float32(4) * a
    c1 = ((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + one
    c2 = ((a + np.float32(2)) * (one - x) - (a + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.float32)


def _axis_cubic(dst, src):
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    taps = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)       # replicated border
    return taps, _cubic_coeffs(f)


def cv2_resize_cubic(src, width, height):
    """cv2.resize(src float32 [h, w, c], (width, height), interpolation=cv2.INTER_CUBIC)."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    h, w, _ = src.shape
    tx, cx = _axis_cubic(width, w)
    rows = np.zeros((h, width, src.shape[2]), np.float32)
    for k in range(4):                                             # v += S[sx + k - 1] * alpha[k], left to right
        rows = (rows + (src[:, tx[:, k], :] * cx[None, :, k, None]).astype(np.float32)).astype(np.float32)
    ty, cy = _axis_cubic(height, h)
    out = np.zeros((height, width, src.shape[2]), np.float32)
    for k in range(4):
        out = (out + (rows[ty[:, k]] * cy[:, k, None, None]).astype(np.float32)).astype(np.float32)
    return out[:, :, 0] if squeeze else out


# ---- pycocotools.mask.encode ---------------------------------------------------------------------------------------
def rle_counts(mask):
    """maskApi.c rleEncode for one [h, w] uint8 mask: run lengths over the column-major pixel order, zeros first."""
    flat = np.asarray(mask, dtype=np.uint8).reshape(-1, order="F")
    if flat.size == 0:
        return [0]
    change = np.nonzero(flat[1:] != flat[:-1])[0] + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    if flat[0] != 0:
        counts = [0] + counts
    return counts


def rle_to_string(counts):
    """maskApi.c rleToString."""
    out = []
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5                                                # arithmetic shift of a signed long
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def rle_from_string(s):
    """maskApi.c rleFrString (used by the round-trip tests)."""
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(counts, h, w):
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        flat[pos:pos + c] = val
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F")


def mask_encode(mask):
    """pycocotools.mask.encode for one mask, counts already decoded to str as lib/core/test.py:841 does."""
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": rle_to_string(rle_counts(mask))}


# ---- lib/utils/boxes.py:233-249 ------------------------------------------------------------------------------------
def expand_boxes(boxes, scale):
    boxes = np.asarray(boxes)
    w_half = (boxes[:, 2] - boxes[:, 0]) * .5
    h_half = (boxes[:, 3] - boxes[:, 1]) * .5
    x_c = (boxes[:, 2] + boxes[:, 0]) * .5
    y_c = (boxes[:, 3] + boxes[:, 1]) * .5
    w_half *= scale
    h_half *= scale
    boxes_exp = np.zeros(boxes.shape)
    boxes_exp[:, 0] = x_c - w_half
    boxes_exp[:, 2] = x_c + w_half
    boxes_exp[:, 1] = y_c - h_half
    boxes_exp[:, 3] = y_c + h_half
    return boxes_exp


# ---- lib/core/test.py:793-847 --------------------------------------------------------------------------------------
def paste_mask(mask_mm, ref_box_int, im_h, im_w, thresh=0.5):
    """One detection of segm_results: zero-pad, resize to the (expanded, truncated) box, binarise, paste."""
    m = mask_mm.shape[0]
    padded = np.zeros((m + 2, m + 2), np.float32)
    padded[1:-1, 1:-1] = mask_mm
    w = max(int(ref_box_int[2] - ref_box_int[0] + 1), 1)
    h = max(int(ref_box_int[3] - ref_box_int[1] + 1), 1)
    mask = (cv2_resize_linear(padded, w, h) > thresh).astype(np.uint8)
    im_mask = np.zeros((im_h, im_w), np.uint8)
    x_0, x_1 = max(int(ref_box_int[0]), 0), min(int(ref_box_int[2]) + 1, im_w)
    y_0, y_1 = max(int(ref_box_int[1]), 0), min(int(ref_box_int[3]) + 1, im_h)
    if x_1 > x_0 and y_1 > y_0:
        im_mask[y_0:y_1, x_0:x_1] = mask[(y_0 - ref_box_int[1]):(y_1 - ref_box_int[1]),
                                         (x_0 - ref_box_int[0]):(x_1 - ref_box_int[0])]
    return im_mask


def segm_results(cls_boxes, masks, ref_boxes, im_h, im_w, cls_specific=True, thresh=0.5):
    """cls_boxes: list over classes of [k_j, 5]; masks [R, K, M, M] in the class-major row order of cls_boxes; ref_boxes
    [R, 4].  Returns cls_segms: per class a list of {'size', 'counts'} dicts."""
    num_classes = len(cls_boxes)
    m = masks.shape[-1]
    scale = (m + 2.0) / m
    ref = expand_boxes(np.asarray(ref_boxes, np.float32), scale).astype(np.int32)
    cls_segms = [[] for _ in range(num_classes)]
    ind = 0
    for j in range(1, num_classes):
        for _ in range(len(cls_boxes[j])):
            src = masks[ind, j] if cls_specific else masks[ind, 0]
            cls_segms[j].append(mask_encode(paste_mask(src, ref[ind], im_h, im_w, thresh)))
            ind += 1
    assert ind == masks.shape[0]
    return cls_segms


# ---- lib/utils/keypoints.py:106-157, 214-222 -----------------------------------------------------------------------
def heatmaps_to_keypoints(maps, rois, min_size=0):
    """maps [R, K, M, M] float32 logits, rois [R, 4] -> xy_preds [R, 4, K] float32 (x, y, logit, prob)."""
    maps = np.asarray(maps, np.float32)
    rois = np.asarray(rois, np.float32)
    num_k = maps.shape[1]
    offset_x, offset_y = rois[:, 0], rois[:, 1]
    widths = np.maximum(rois[:, 2] - rois[:, 0], 1)
    heights = np.maximum(rois[:, 3] - rois[:, 1], 1)
    widths_ceil, heights_ceil = np.ceil(widths), np.ceil(heights)
    maps = np.transpose(maps, [0, 2, 3, 1])
    xy_preds = np.zeros((len(rois), 4, num_k), dtype=np.float32)
    for i in range(len(rois)):
        if min_size > 0:
            roi_map_width = int(np.maximum(widths_ceil[i], min_size))
            roi_map_height = int(np.maximum(heights_ceil[i], min_size))
        else:
            roi_map_width, roi_map_height = int(widths_ceil[i]), int(heights_ceil[i])
        width_correction = widths[i] / roi_map_width
        height_correction = heights[i] / roi_map_height
        roi_map = np.transpose(cv2_resize_cubic(maps[i], roi_map_width, roi_map_height), [2, 0, 1])
        w = roi_map.shape[2]
        for k in range(num_k):
            temp = roi_map[k]
            pos = temp.argmax()                                    # np.int64: (x_int + 0.5) below is float64, as there
            x_int = pos % w
            y_int = (pos - x_int) // w
            e = np.exp(temp - temp.max())                          # scores_to_probs
            prob = (e / np.sum(e))[y_int, x_int]
            xy_preds[i, 0, k] = (x_int + 0.5) * width_correction + offset_x[i]
            xy_preds[i, 1, k] = (y_int + 0.5) * height_correction + offset_y[i]
            xy_preds[i, 2, k] = temp[y_int, x_int]
            xy_preds[i, 3, k] = prob
    return xy_preds


# ---- lib/utils/keypoints.py:225-266 ---------------------------------------------------------------------------------
def compute_oks(src_keypoints, src_roi, dst_keypoints, dst_roi):
    """keypoints.py:243-266: OKS of the predicted keypoints [N,4,K] with respect to src_keypoints [4,K]; numpy only in the
    reference, so this restatement is pinned by executing the reference's own text (tests/test_results_cpu.py)."""
    sigmas = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
    vars = (sigmas * 2) ** 2
    src_area = (src_roi[2] - src_roi[0] + 1) * (src_roi[3] - src_roi[1] + 1)      # :257
    dx = dst_keypoints[:, 0, :] - src_keypoints[0, :]                              # :260-261
    dy = dst_keypoints[:, 1, :] - src_keypoints[1, :]
    e = (dx ** 2 + dy ** 2) / vars / (src_area + np.spacing(1)) / 2                # :263
    e = np.sum(np.exp(-e), axis=1) / e.shape[1]                                    # :264
    return e


def nms_oks(kp_predictions, rois, thresh):
    """keypoints.py:225-240: greedy NMS on OKS, best mean keypoint logit first."""
    scores = np.mean(kp_predictions[:, 2, :], axis=1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        ovr = compute_oks(kp_predictions[i], rois[i], kp_predictions[order[1:]], rois[order[1:]])
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
    return keep
