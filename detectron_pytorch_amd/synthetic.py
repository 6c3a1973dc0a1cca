"""Seeded synthetic workloads of SURVEY.md section 8(d) (numpy only; shared by tests/ and bench.py).

Shapes follow the padded blob of a 1333x800 image: [N,3,800,1344]; FPN maps with C=256:
P2 200x336 (scale 1/4), P3 100x168, P4 50x84, P5 25x42 (lib/utils/blob.py:97-100, lib/core/config.py:686,693).
"""
import numpy as np

IM_H, IM_W = 800, 1344
FPN_LEVELS = {2: (200, 336, 1.0 / 4), 3: (100, 168, 1.0 / 8), 4: (50, 84, 1.0 / 16), 5: (25, 42, 1.0 / 32)}
FPN_DIM = 256


def boxes_uniform(n, seed=0):
    """Config-1 'uniform' set: x1~U(0,1332), y1~U(0,799), w,h~U(8,512), clipped; unique scores."""
    rng = np.random.RandomState(seed)
    x1 = rng.uniform(0, 1332, n)
    y1 = rng.uniform(0, 799, n)
    w = rng.uniform(8, 512, n)
    h = rng.uniform(8, 512, n)
    x2 = np.minimum(x1 + w, 1332)
    y2 = np.minimum(y1 + h, 799)
    scores = rng.permutation(n) / float(max(n, 1))
    return np.stack([x1, y1, x2, y2, scores], axis=1).astype(np.float32)


def boxes_clustered(n, seed=0, centres=50, jitter=0.08):
    """Config-1 'clustered' set (RPN-like, heavy suppression): boxes = centre box * (1 + N(0, jitter))."""
    rng = np.random.RandomState(seed)
    cx1 = rng.uniform(0, 1100, centres)
    cy1 = rng.uniform(0, 600, centres)
    cw = rng.uniform(32, 400, centres)
    ch = rng.uniform(32, 400, centres)
    base = np.stack([cx1, cy1, cx1 + cw, cy1 + ch], axis=1)
    pick = rng.randint(0, centres, n)
    b = base[pick] * (1.0 + rng.normal(0, jitter, (n, 4)))
    x1 = np.clip(np.minimum(b[:, 0], b[:, 2]), 0, 1332)
    x2 = np.clip(np.maximum(b[:, 0], b[:, 2]), 0, 1332)
    y1 = np.clip(np.minimum(b[:, 1], b[:, 3]), 0, 799)
    y2 = np.clip(np.maximum(b[:, 1], b[:, 3]), 0, 799)
    scores = rng.permutation(n) / float(max(n, 1))
    return np.stack([x1, y1, x2, y2, scores], axis=1).astype(np.float32)


def sort_by_score(dets):
    """Descending score, ties -> higher index first (the tie rule of oracle.c / nms.hip)."""
    order = np.argsort(dets[:, 4], kind="stable")[::-1]
    return np.ascontiguousarray(dets[order]), order


def rois_canonical(num_rois=512, batch=1, seed=0, side=(16.0, 112.0), im_h=IM_H, im_w=IM_W):
    """Config-2 RoIs: centre uniform in the image, side ~U(16,112) px (FPN level-2 sized boxes,
    lib/utils/fpn.py:23), clipped to the image; batch index round-robin over `batch` images."""
    rng = np.random.RandomState(seed)
    cx = rng.uniform(0, im_w - 1, num_rois)
    cy = rng.uniform(0, im_h - 1, num_rois)
    w = rng.uniform(side[0], side[1], num_rois)
    h = rng.uniform(side[0], side[1], num_rois)
    x1 = np.clip(cx - w / 2, 0, im_w - 1)
    x2 = np.clip(cx + w / 2, 0, im_w - 1)
    y1 = np.clip(cy - h / 2, 0, im_h - 1)
    y2 = np.clip(cy + h / 2, 0, im_h - 1)
    b = (np.arange(num_rois) % batch).astype(np.float64)
    return np.stack([b, x1, y1, x2, y2], axis=1).astype(np.float32)


def rois_adversarial(num_rois, batch, height, width, spatial_scale, seed=0):
    """Random RoIs that exercise every branch of the kernels: out-of-image, malformed (x2<x1),
    sub-pixel, whole-image, and boxes hugging each border."""
    rng = np.random.RandomState(seed)
    im_w, im_h = width / spatial_scale, height / spatial_scale
    x1 = rng.uniform(-0.3 * im_w, 1.2 * im_w, num_rois)
    y1 = rng.uniform(-0.3 * im_h, 1.2 * im_h, num_rois)
    w = rng.uniform(-0.1 * im_w, 0.8 * im_w, num_rois)
    h = rng.uniform(-0.1 * im_h, 0.8 * im_h, num_rois)
    r = np.stack([rng.randint(0, batch, num_rois).astype(np.float64), x1, y1, x1 + w, y1 + h], axis=1)
    special = [
        [0, 0, 0, im_w - 1, im_h - 1],                 # whole image
        [0, -50, -50, im_w + 50, im_h + 50],           # larger than the image
        [0, 10.3, 10.7, 10.9, 11.1],                   # sub-pixel
        [0, im_w - 2, im_h - 2, im_w + 30, im_h + 30],  # bottom-right corner
        [0, -40, -40, 1.5, 1.5],                       # top-left corner
        [0, 50, 50, 40, 40],                           # malformed
        [0, -3 * im_w, -3 * im_h, -2 * im_w, -2 * im_h],  # completely outside
        [0, 5, 5, 5, 5],                               # zero size
    ]
    for i, s in enumerate(special[:num_rois]):
        r[i] = s
        r[i, 0] = i % batch
    return r.astype(np.float32)


def map_rois_to_fpn_levels(rois_xyxy, k_min=2, k_max=5, s0=224.0, lvl0=4):
    """lib/utils/fpn.py:11-28; the implementation lives with its caller in roi_xform.py."""
    from .roi_xform import map_rois_to_fpn_levels as impl

    return impl(rois_xyxy, k_min, k_max, s0, lvl0)


def rois_fpn_distributed(num_rois=1000, batch=1, seed=0, im_h=IM_H, im_w=IM_W):
    """Config-2 variant (ii): RoIs with log-uniform sizes 16..700 px spread over FPN levels 2..5.
    Returns (rois [R,5], levels [R])."""
    rng = np.random.RandomState(seed)
    side = np.exp(rng.uniform(np.log(16.0), np.log(700.0), num_rois))
    aspect = np.exp(rng.uniform(np.log(0.5), np.log(2.0), num_rois))
    w = side * np.sqrt(aspect)
    h = side / np.sqrt(aspect)
    cx = rng.uniform(0, im_w - 1, num_rois)
    cy = rng.uniform(0, im_h - 1, num_rois)
    x1 = np.clip(cx - w / 2, 0, im_w - 1)
    x2 = np.clip(cx + w / 2, 0, im_w - 1)
    y1 = np.clip(cy - h / 2, 0, im_h - 1)
    y2 = np.clip(cy + h / 2, 0, im_h - 1)
    b = (np.arange(num_rois) % batch).astype(np.float64)
    rois = np.stack([b, x1, y1, x2, y2], axis=1).astype(np.float32)
    return rois, map_rois_to_fpn_levels(rois[:, 1:5])


def feature_map(batch, channels, height, width, seed=0):
    return np.random.RandomState(seed).randn(batch, channels, height, width).astype(np.float32)


def crop_grid(num_rois, gh, gw, seed=0, span=1.25):
    """[R,gh,gw,2] (y,x) grids: affine boxes in normalised coordinates, a few reaching outside [-1,1]."""
    rng = np.random.RandomState(seed)
    cy = rng.uniform(-span, span, (num_rois, 1, 1))
    cx = rng.uniform(-span, span, (num_rois, 1, 1))
    sy = rng.uniform(0.05, 0.6, (num_rois, 1, 1))
    sx = rng.uniform(0.05, 0.6, (num_rois, 1, 1))
    ly = np.linspace(-1, 1, gh).reshape(1, gh, 1)
    lx = np.linspace(-1, 1, gw).reshape(1, 1, gw)
    y = cy + sy * ly + 0 * lx
    x = cx + sx * lx + 0 * ly
    return np.stack([y, x], axis=3).astype(np.float32)


def detection_head_outputs(num_rois=300, num_classes=21, seed=0, im_h=IM_H, im_w=IM_W):
    """Inputs of the test-time post-processing (core/test.py:732-790): class scores [R, C] (softmax-like rows, a few
    confident classes per RoI, many RoIs near duplicates of each other) and per-class boxes [R, 4C]."""
    rng = np.random.RandomState(seed)
    base = boxes_clustered(num_rois, seed=seed + 1)[:, :4].astype(np.float64)
    logits = rng.randn(num_rois, num_classes) * 1.5
    hot = rng.randint(1, num_classes, num_rois)
    logits[np.arange(num_rois), hot] += rng.uniform(0, 6, num_rois)
    logits[:, 0] += 2.0
    e = np.exp(logits - logits.max(1, keepdims=True))
    scores = (e / e.sum(1, keepdims=True)).astype(np.float32)
    jitter = rng.uniform(-4, 4, (num_rois, num_classes, 4))
    boxes = base[:, None, :] + jitter
    boxes[..., 0::2] = np.clip(boxes[..., 0::2], 0, im_w - 1)
    boxes[..., 1::2] = np.clip(boxes[..., 1::2], 0, im_h - 1)
    boxes[..., 2] = np.maximum(boxes[..., 2], boxes[..., 0])
    boxes[..., 3] = np.maximum(boxes[..., 3], boxes[..., 1])
    return scores, boxes.reshape(num_rois, 4 * num_classes).astype(np.float32)


def result_format_inputs(num_dets=100, num_person=20, mask_size=28, heat=56, seed=0, im_h=800, im_w=1333):
    """Inputs of the test-time result formats (core/test.py:793-866): blob-like soft masks [D, M, M] with their detection
    boxes (COCO-like sizes, some over the image border), and peaked keypoint heat maps [P, 17, H, H] with person boxes."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:mask_size, 0:mask_size].astype(np.float32)
    masks = np.zeros((num_dets, mask_size, mask_size), np.float32)
    for i in range(num_dets):
        cx, cy = rng.uniform(0.3, 0.7, 2) * mask_size
        s = rng.uniform(0.2, 0.5) * mask_size
        masks[i] = 1 / (1 + np.exp(-(1.5 - ((xx - cx) ** 2 + (yy - cy) ** 2) / s ** 2 + rng.randn(mask_size, mask_size) * 0.3)))
    w, h = rng.uniform(16, 500, num_dets), rng.uniform(16, 400, num_dets)
    x1, y1 = rng.uniform(-20, im_w - 30, num_dets), rng.uniform(-20, im_h - 30, num_dets)
    boxes = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    maps = rng.randn(num_person, 17, heat, heat).astype(np.float32)
    for i in range(num_person):
        for k in range(17):
            py, px = rng.randint(4, heat - 4, 2)
            maps[i, k, py - 2:py + 3, px - 2:px + 3] += 5.0
    pw, ph = rng.uniform(40, 300, num_person), rng.uniform(80, 600, num_person)
    px1, py1 = rng.uniform(0, im_w - 310, num_person), rng.uniform(0, im_h - 610, num_person).clip(0)
    person = np.stack([px1, py1, px1 + pw, py1 + ph], 1).astype(np.float32)
    return masks, boxes, maps, person


def polygon_instances(num_instances=8, seed=0, im_h=800, im_w=1333):
    """COCO-like ground-truth instances in the roidb's `segms` format (json_dataset.py:218-262): per instance a list of 1-3
    polygons (flat x0, y0, x1, y1, ... with two decimals, as the annotation files carry them) -- star-shaped outlines of
    8-60 vertices around the instance's centre, concave ones, an occasional self-intersecting one, repeated vertices and
    parts over the image border included.  Returns (segms, boxes [G, 4] float32 = the tight boxes, classes [G] int32)."""
    rng = np.random.RandomState(seed)
    segms, boxes = [], []
    for _ in range(num_instances):
        cx, cy = rng.uniform(60, im_w - 60), rng.uniform(60, im_h - 60)
        rx, ry = rng.uniform(16, 200), rng.uniform(16, 200)
        polys = []
        for part in range(rng.randint(1, 4)):
            k = rng.randint(8, 61)
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            rad = rng.uniform(0.35, 1.0, k)
            if rng.rand() < 0.15:
                ang = rng.permutation(ang)                       # self-intersecting
            ox, oy = (0.0, 0.0) if part == 0 else rng.uniform(-0.8, 0.8, 2) * (rx, ry)
            scale = 1.0 if part == 0 else rng.uniform(0.2, 0.5)
            x = cx + ox + scale * rx * rad * np.cos(ang)
            y = cy + oy + scale * ry * rad * np.sin(ang)
            pts = np.round(np.stack([x, y], 1), 2)
            if rng.rand() < 0.3:
                pts = np.insert(pts, rng.randint(0, k), pts[rng.randint(0, k)], axis=0)   # a vertex visited twice
            if rng.rand() < 0.3:
                pts = np.repeat(pts, 1 + (rng.rand(pts.shape[0]) < 0.2), axis=0)           # consecutive duplicates
            polys.append([float(v) for v in pts.reshape(-1)])
        segms.append(polys)
        allp = np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in polys])
        boxes.append([allp[:, 0].min(), allp[:, 1].min(), allp[:, 0].max(), allp[:, 1].max()])
    return segms, np.asarray(boxes, np.float32), rng.randint(1, 81, num_instances).astype(np.int32)


def jittered_boxes(boxes, per_box, seed=0, jitter=0.25):
    """Foreground-like RoIs: every box `per_box` times with its corners moved by up to `jitter` of its size (IoU with
    the box mostly above 0.5), float32 [len(boxes) * per_box, 4]."""
    rng = np.random.RandomState(seed)
    b = np.repeat(np.asarray(boxes, np.float32), per_box, axis=0)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    d = rng.uniform(-jitter, jitter, b.shape) * np.stack([w, h, w, h], 1)
    out = b + d
    return np.stack([np.minimum(out[:, 0], out[:, 2]), np.minimum(out[:, 1], out[:, 3]),
                     np.maximum(out[:, 0], out[:, 2]), np.maximum(out[:, 1], out[:, 3])], 1).astype(np.float32)


def roi_align_touched_pixels(rois, batch, height, width, aligned_height, aligned_width, spatial_scale, sampling_ratio):
    """U of the algorithmic-bytes formula: the number of distinct feature pixels (n, y, x) that
    any sample of any RoI references with a non-zero weight.  float32 sampling arithmetic of roi_align_kernel.cu:74-110
    and :16-52 (same operation order), evaluated per axis; a workload descriptor, not a kernel."""
    f32 = np.float32
    seen = np.zeros((batch, height, width), dtype=bool)

    def axis(start, bin_size, pooled, grid, size):
        p = np.repeat(np.arange(pooled, dtype=f32), grid)
        i = np.tile(np.arange(grid, dtype=f32), pooled)
        v = (start + p * bin_size) + ((i + f32(0.5)) * bin_size) / f32(grid)
        ok = ~((v < f32(-1.0)) | (v > f32(size)))
        v = np.where(v <= 0, f32(0), v)
        lo = v.astype(np.int32)
        edge = lo >= size - 1
        lo = np.where(edge, size - 1, lo)
        hi = np.where(edge, size - 1, lo + 1)
        v = np.where(edge, lo.astype(f32), v)
        lw = v - lo.astype(f32)
        hw = f32(1.0) - lw
        return ok, lo, hi, hw.astype(f32), lw.astype(f32)

    for r in np.asarray(rois, dtype=f32):
        b = int(r[0])
        if b < 0 or b >= batch:
            continue
        sw, sh = r[1] * f32(spatial_scale), r[2] * f32(spatial_scale)
        rw = max(r[3] * f32(spatial_scale) - sw, f32(1.0))
        rh = max(r[4] * f32(spatial_scale) - sh, f32(1.0))
        bh, bw = f32(rh / f32(aligned_height)), f32(rw / f32(aligned_width))
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / f32(aligned_height)))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / f32(aligned_width)))
        oky, yl, yh, hy, ly = axis(sh, bh, aligned_height, gh, height)
        okx, xl, xh, hx, lx = axis(sw, bw, aligned_width, gw, width)
        ok = oky[:, None] & okx[None, :]
        plane = seen[b]
        for ys, wy in ((yl, hy), (yh, ly)):
            for xs, wx in ((xl, hx), (xh, lx)):
                hit = ok & ((wy[:, None] * wx[None, :]) != 0)
                yy, xx = np.nonzero(hit)
                plane[ys[yy], xs[xx]] = True
    return int(seen.sum())


def rpn_head_outputs(batch=2, num_anchors=3, height=50, width=84, seed=0):
    """Inputs of GenerateProposalsOp (lib/modeling/generate_proposals.py:19-45): objectness probabilities [N,A,H,W] in
    (0,1) with UNIQUE values per image (the reference's argsort has no defined order for ties) and box deltas
    [N,4A,H,W] (dx, dy ~ N(0, 0.5); dw, dh ~ N(0, 0.4) with a few beyond the BBOX_XFORM_CLIP)."""
    rng = np.random.RandomState(seed)
    n = num_anchors * height * width
    scores = np.stack([rng.permutation(n).astype(np.float64) for _ in range(batch)])
    scores = ((scores + 0.5) / n).astype(np.float32).reshape(batch, num_anchors, height, width)
    deltas = rng.normal(0, 0.5, (batch, 4 * num_anchors, height, width))
    deltas[:, 2::4] *= 0.8
    deltas[:, 3::4] *= 0.8
    big = rng.rand(*deltas.shape) < 0.002
    deltas = np.where(big, deltas * 20, deltas)
    return scores, deltas.astype(np.float32)
