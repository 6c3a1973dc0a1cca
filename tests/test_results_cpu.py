"""The restatement of the result formats (oracle/results.py: cv2.resize linear / cubic for float32, pycocotools RLE) checked
for internal consistency, and the package's host-side string encoder against it.  OpenCV / pycocotools are absent here, so
these are properties and hand-computed vectors, not comparisons with the packages (parity unpinned, see the oracle)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import results as R  # noqa: E402


def test_rle_hand_vectors_and_round_trip():
    m = np.zeros((5, 4), np.uint8)
    m[1:3, 1] = 1                                          # column-major: 6 zeros, 2 ones, 12 zeros
    assert R.rle_counts(m) == [6, 2, 12]
    assert R.rle_to_string([6, 2, 12]) == "62<"            # one 5-bit group each: chr(v + 48)
    full = np.ones((3, 2), np.uint8)
    assert R.rle_counts(full) == [0, 6]                    # the first run counts zeros, even when there are none
    assert R.rle_counts(np.zeros((3, 2), np.uint8)) == [6]
    # 40 = 0b01000 + (1 << 5): low group 8 with the continuation bit (|0x20 -> 40 + 48 = 'X'), then 1 -> '1'
    assert R.rle_to_string([40]) == "X1"
    # from the fourth run on the difference to the run two back is stored: [5, 3, 7, 3] -> 5, 3, 7, 0
    assert R.rle_to_string([5, 3, 7, 3]) == "5370"
    # a negative difference needs the sign bit of the last group: [1, 9, 1, 2] -> 1, 9, 1, -7 = ...11001 -> 0x19 + 48
    assert R.rle_to_string([1, 9, 1, 2]) == "191" + chr(0x19 + 48)
    rng = np.random.RandomState(0)
    for h, w, p in ((37, 53, 0.4), (800, 1333, 0.001), (1, 1, 0.5), (64, 1, 0.5)):
        mask = (rng.rand(h, w) < p).astype(np.uint8)
        counts = R.rle_counts(mask)
        assert sum(counts) == h * w
        s = R.rle_to_string(counts)
        assert R.rle_from_string(s) == counts
        assert np.array_equal(R.rle_decode(counts, h, w), mask)


def test_package_string_encoder_equals_the_restatement():
    from detectron_pytorch_amd.rcnn import results

    rng = np.random.RandomState(1)
    for n in (1, 2, 3, 4, 7, 500):
        c = rng.randint(0, 2_000_000, n)
        c[rng.rand(n) < 0.3] = rng.randint(0, 40, int((rng.rand(n) < 0.3).sum()) or 1)[0]
        assert results.rle_to_string(c) == R.rle_to_string(c.tolist())


def test_resize_restatement_properties():
    rng = np.random.RandomState(2)
    a = rng.rand(30, 30).astype(np.float32)
    assert np.array_equal(R.cv2_resize_linear(a, 30, 30), a)                     # identity at scale 1
    assert np.array_equal(R.cv2_resize_cubic(a, 30, 30), a)
    const = np.full((30, 30), 0.7, np.float32)
    assert np.allclose(R.cv2_resize_linear(const, 77, 13), 0.7, atol=1e-6)      # partition of unity
    assert np.allclose(R.cv2_resize_cubic(const, 77, 13), 0.7, atol=1e-6)
    up = R.cv2_resize_linear(a, 91, 64)
    assert up.shape == (64, 91) and up.min() >= a.min() - 1e-6 and up.max() <= a.max() + 1e-6
    # exact 2x up-scale of a ramp: destination centres fall at quarter positions ((dx + .5) / 2 - .5)
    ramp = np.arange(8, dtype=np.float32)[None, :].repeat(4, 0)
    got = R.cv2_resize_linear(ramp, 16, 4)[0]
    want = np.clip((np.arange(16) + 0.5) / 2 - 0.5, 0, 7).astype(np.float32)
    assert np.allclose(got, want, atol=1e-6)
    # three channels resize independently
    b = rng.rand(12, 9, 3).astype(np.float32)
    c3 = R.cv2_resize_cubic(b, 20, 31)
    for ch in range(3):
        assert np.array_equal(c3[:, :, ch], R.cv2_resize_cubic(b[:, :, ch], 20, 31))


def test_resize_restatement_against_an_independent_implementation():
    """cv2 is not installed, so the restated cv2.resize (coordinate map fx = (dx + .5) * scale - .5, floor, border handling,
    bilinear weights / bicubic kernel A = -0.75 with a replicated border) is cross-checked against an implementation that
    shares none of its code: torch.nn.functional.interpolate(align_corners=False), which documents the same coordinate map
    and the same A = -0.75 kernel.  Sizes: the mask paste (28 x 28 -> box, core/test.py:826-838, both up- and down-scaling
    boxes) and the key-point heat maps (56 x 56 -> box, utils/keypoints.py:129-152), borders included.  Tolerance 1e-4 of
    the data range: OpenCV forms the source coordinate in double and rounds it to fp32, torch forms it in fp32 from an
    fp32 scale, so the interpolation weights differ by ~1e-6 x coordinate (measured: <= 3.5e-5 on these sizes); a wrong
    coordinate map, a missing half-pixel shift, a different A or border rule is off by 1e-2 or more on these inputs."""
    import torch
    import torch.nn.functional as F

    rng = np.random.RandomState(7)
    cases = [(28, 28, w, h) for (w, h) in ((28, 28), (29, 31), (57, 40), (113, 200), (400, 333), (15, 9), (3, 2), (1, 1))]
    cases += [(56, 56, w, h) for (w, h) in ((56, 56), (61, 130), (200, 77), (333, 480), (40, 25))]
    cases += [(30, 30, 77, 13)]  # the padded mask (M + 2) of the paste
    for (sh, sw, dw, dh) in cases:
        a = rng.randn(sh, sw).astype(np.float32)
        t = torch.from_numpy(a)[None, None]
        lin = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
        got = R.cv2_resize_linear(a, dw, dh)
        assert got.shape == (dh, dw)
        # (up- and down-scaling alike: neither cv2.resize with these flags nor interpolate(antialias=False) low-passes)
        assert np.abs(got - lin).max() <= 1e-4 * max(1.0, np.abs(a).max()), (sh, sw, dw, dh)
        cub = F.interpolate(t, size=(dh, dw), mode="bicubic", align_corners=False)[0, 0].numpy()
        got = R.cv2_resize_cubic(a, dw, dh)
        assert np.abs(got - cub).max() <= 1e-4 * max(1.0, np.abs(a).max()), (sh, sw, dw, dh)
    # the check has teeth: a restatement without the half-pixel shift differs by far more than the tolerance
    a = rng.randn(28, 28).astype(np.float32)
    shifted = F.interpolate(torch.from_numpy(a)[None, None], size=(57, 57), mode="bilinear", align_corners=True)[0, 0].numpy()
    assert np.abs(R.cv2_resize_linear(a, 57, 57) - shifted).max() > 1e-2


def test_paste_and_segm_results_shapes():
    rng = np.random.RandomState(3)
    masks = rng.rand(3, 4, 28, 28).astype(np.float32)
    ref = np.array([[10, 20, 100, 90], [-20, -5, 40, 30], [600, 400, 700, 520]], np.float32)
    cls_boxes = [[], np.zeros((2, 5)), np.zeros((0, 5)), np.zeros((1, 5))]
    segms = R.segm_results(cls_boxes, masks, ref, 480, 640)
    assert [len(s) for s in segms] == [0, 2, 0, 1]
    for rles in segms:
        for rle in rles:
            counts = R.rle_from_string(rle["counts"])
            assert rle["size"] == [480, 640] and sum(counts) == 480 * 640
    # the pasted mask is confined to the expanded box and the image
    box = R.expand_boxes(ref, 30.0 / 28).astype(np.int32)[2]
    im = R.paste_mask(masks[2, 3], box, 480, 640)
    ys, xs = np.nonzero(im)
    assert xs.size == 0 or (xs.min() >= max(box[0], 0) and xs.max() <= min(box[2], 639) and ys.max() <= 479)


def test_detections_pkl_layout(tmp_path):
    """test_engine.py:300-313, :369-397: the pickle a reference evaluation run would read."""
    import pickle

    import torch

    from detectron_pytorch_amd.rcnn import results

    all_boxes, all_segms, all_keyps = results.empty_results(3, 2)
    assert len(all_boxes) == 3 and len(all_boxes[0]) == 2 and all_boxes[1][0] is not all_boxes[1][1]
    cls_boxes = [[], torch.tensor([[1.0, 2, 3, 4, 0.9]]), torch.zeros((0, 5))]
    cls_segms = [[], [{"size": [4, 5], "counts": "62<"}], []]
    cls_keyps = [[], [torch.ones(4, 17)], []]
    results.extend_results(1, all_boxes, cls_boxes)
    results.extend_results(1, all_segms, cls_segms)
    results.extend_results(1, all_keyps, cls_keyps)
    path = results.save_detections(str(tmp_path / "detections.pkl"), all_boxes, all_segms, all_keyps, "MODEL: {}")
    got = pickle.load(open(path, "rb"))
    assert sorted(got) == ["all_boxes", "all_keyps", "all_segms", "cfg"] and got["cfg"] == "MODEL: {}"
    assert got["all_boxes"][1][0] == [] and got["all_boxes"][1][1].shape == (1, 5)
    assert got["all_segms"][1][1] == cls_segms[1] and got["all_keyps"][1][1][0].shape == (4, 17)


REFERENCE = "/root/reference/lib"


def _reference_functions():
    """segm_results (core/test.py), heatmaps_to_keypoints / scores_to_probs (utils/keypoints.py) and expand_boxes
    (utils/boxes.py) executed from the reference's OWN source text; only the two absent third-party packages are stubbed,
    by the restatements under test: cv2.resize -> oracle.results.cv2_resize_*, pycocotools.mask.encode -> the restated RLE."""
    import re
    import types

    def fn(path, name):
        src = open(os.path.join(REFERENCE, path)).read()
        return re.search(r"^def %s\(.*?(?=^def |\Z)" % name, src, re.S | re.M).group(0)

    cv2 = types.SimpleNamespace(INTER_CUBIC=2, INTER_LINEAR=1)

    def resize(src, dsize, interpolation=1):
        f = R.cv2_resize_cubic if interpolation == cv2.INTER_CUBIC else R.cv2_resize_linear
        return f(src, int(dsize[0]), int(dsize[1]))

    cv2.resize = resize

    def encode(fortran_masks):
        return [{"size": list(fortran_masks.shape[:2]),
                 "counts": R.rle_to_string(R.rle_counts(fortran_masks[:, :, i])).encode("ascii")}
                for i in range(fortran_masks.shape[2])]

    def make(cfg):
        box_utils = types.ModuleType("box_utils")
        box_utils.np = np
        exec(compile(fn("utils/boxes.py", "expand_boxes"), "utils/boxes.py", "exec"), box_utils.__dict__)
        ns = {"np": np, "cv2": cv2, "cfg": cfg, "box_utils": box_utils, "mask_util": types.SimpleNamespace(encode=encode)}
        exec(compile(fn("core/test.py", "segm_results"), "core/test.py", "exec"), ns)
        exec(compile(fn("utils/keypoints.py", "scores_to_probs") + fn("utils/keypoints.py", "heatmaps_to_keypoints"),
                     "utils/keypoints.py", "exec"), ns)
        return ns

    return make, types


def test_reference_call_sites_run_on_the_restated_packages():
    """Pins everything AROUND the two absent packages to the reference: its own segm_results / heatmaps_to_keypoints text
    gives exactly what oracle/results.py's restatement of them gives (box expansion and truncation, padding, paste
    offsets, class order, the float64 coordinate arithmetic of the keypoints)."""
    import pytest

    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference tree is not present on this machine")
    make, types = _reference_functions()
    rng = np.random.RandomState(5)
    for m, cls_specific in ((28, True), (14, False)):
        cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(NUM_CLASSES=5),
                                    MRCNN=types.SimpleNamespace(RESOLUTION=m, CLS_SPECIFIC_MASK=cls_specific, THRESH_BINARIZE=0.5),
                                    KRCNN=types.SimpleNamespace(INFERENCE_MIN_SIZE=0, NUM_KEYPOINTS=17))
        ns = make(cfg)
        lengths = [0, 2, 0, 3, 1]
        r = sum(lengths)
        masks = rng.rand(r, 5 if cls_specific else 1, m, m).astype(np.float32)
        x1, y1 = rng.uniform(-30, 500, r), rng.uniform(-30, 300, r)
        ref_boxes = np.stack([x1, y1, x1 + rng.uniform(2, 250, r), y1 + rng.uniform(2, 200, r)], 1).astype(np.float32)
        cls_boxes = [np.zeros((n, 5), np.float32) for n in lengths]
        want = ns["segm_results"](cls_boxes, masks, ref_boxes, 427, 640)
        got = R.segm_results(cls_boxes, masks, ref_boxes, 427, 640, cls_specific=cls_specific)
        assert got == want
    for min_size in (0, 48):
        cfg.KRCNN.INFERENCE_MIN_SIZE = min_size
        ns = make(cfg)
        maps = rng.randn(4, 17, 56, 56).astype(np.float32)
        rois = np.array([[10, 20, 110, 220], [0, 0, 30.5, 15.2], [5, 5, 5.5, 5.2], [100.3, 50.7, 320.9, 410.1]], np.float32)
        want = ns["heatmaps_to_keypoints"](maps.copy(), rois)
        got = R.heatmaps_to_keypoints(maps, rois, min_size)
        assert want.dtype == got.dtype == np.float32 and np.array_equal(want, got)


def _people(n, seed):
    """xy_preds [n, 4, 17] of persons that come in overlapping groups (so that OKS-NMS has something to do) and their boxes."""
    rng = np.random.RandomState(seed)
    base = rng.uniform(50, 600, size=(max(n // 4, 1), 2, 17)).astype(np.float32)
    kp = np.zeros((n, 4, 17), np.float32)
    rois = np.zeros((n, 4), np.float32)
    for i in range(n):
        b = base[i % len(base)]
        jitter = rng.normal(0, rng.choice([1.0, 6.0, 40.0]), size=(2, 17)).astype(np.float32)
        kp[i, :2] = b + jitter
        kp[i, 2] = rng.normal(2.0, 1.5, 17).astype(np.float32)
        kp[i, 3] = rng.uniform(0, 1, 17).astype(np.float32)
        x1, y1 = kp[i, 0].min() - 5, kp[i, 1].min() - 5
        rois[i] = (x1, y1, kp[i, 0].max() + 5, kp[i, 1].max() + 5)
    return kp, rois


def test_nms_oks_restatement_equals_the_reference_text():
    """utils/keypoints.py:225-266 is numpy only: the reference's own nms_oks / compute_oks source is executed and must give
    the kept list of oracle/results.py's restatement, value for value (OKS in float64)."""
    import re

    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference tree is not present on this machine")
    src = open(os.path.join(REFERENCE, "utils", "keypoints.py")).read()
    text = "".join(re.search(r"^def %s\(.*?(?=^def |\Z)" % name, src, re.S | re.M).group(0) for name in ("nms_oks", "compute_oks"))
    ns = {"np": np}
    exec(compile(text, "utils/keypoints.py", "exec"), ns)
    for n, seed in ((1, 0), (12, 1), (60, 2), (200, 3)):
        kp, rois = _people(n, seed)
        for thresh in (0.3, 0.05, 0.9):
            want = [int(i) for i in ns["nms_oks"](kp, rois, thresh)]
            got = [int(i) for i in R.nms_oks(kp, rois, thresh)]
            assert got == want
            assert 1 <= len(got) <= n
        assert np.array_equal(ns["compute_oks"](kp[0], rois[0], kp, rois), R.compute_oks(kp[0], rois[0], kp, rois))
    kp, rois = _people(60, 2)
    assert len(R.nms_oks(kp, rois, 0.3)) < 60          # the groups really collapse


# ---- cross-checks that share no code with oracle/results.py (the restatement stays "parity unpinned": pycocotools and cv2 are
# absent; these rule out that restatement and kernel agree with each other only because one hand wrote both) -------------------
def _indep_counts(mask):
    """COCO RLE counts straight from the format description: walk the pixels column by column, count runs, start with zeros."""
    h, w = mask.shape
    counts, run, val = [], 0, 0
    for x in range(w):
        for y in range(h):
            if int(mask[y, x]) != val:
                counts.append(run)
                run, val = 0, 1 - val
            run += 1
    counts.append(run)
    return counts


def _indep_string(counts):
    """The compressed form from its description: every count from the fourth on is replaced by its difference to the count two
    places back; each value is written little-endian in 5-bit groups of a two's-complement number, shortest form that still
    sign-extends correctly (the last group's bit 4 is the sign), bit 5 of a character = another group follows, + 48."""
    chars = []
    for i, c in enumerate(counts):
        v = c - counts[i - 2] if i > 2 else c
        ngroups = 1
        while not (-(1 << (5 * ngroups - 1)) <= v < (1 << (5 * ngroups - 1))):
            ngroups += 1
        u = v & ((1 << (5 * ngroups)) - 1)                  # two's complement on 5 * ngroups bits
        for g in range(ngroups):
            d = (u >> (5 * g)) & 31
            chars.append(chr(48 + d + (32 if g < ngroups - 1 else 0)))
    return "".join(chars)


def _indep_decode_string(s, h, w):
    vals, i = [], 0
    while i < len(s):
        groups = []
        while True:
            d = ord(s[i]) - 48
            i += 1
            groups.append(d & 31)
            if not d & 32:
                break
        u = sum(g << (5 * k) for k, g in enumerate(groups))
        bits = 5 * len(groups)
        v = u - (1 << bits) if u >> (bits - 1) else u
        vals.append(v + vals[-2] if len(vals) > 2 else v)
    out, pos, val = np.zeros(h * w, np.uint8), 0, 0
    for c in vals:
        out[pos:pos + c] = val
        pos += c
        val ^= 1
    assert pos == h * w
    return np.ascontiguousarray(out.reshape(w, h).T)


def _adversarial_masks():
    rng = np.random.RandomState(11)
    yield np.zeros((7, 5), np.uint8)
    yield np.ones((7, 5), np.uint8)
    yield (np.indices((9, 6)).sum(0) & 1).astype(np.uint8)                       # checkerboard: every run is 1
    m = np.zeros((300, 400), np.uint8)
    m[:, 150:] = 1                                                               # runs of 45 000 and 75 000: four groups
    yield m
    m = np.zeros((64, 64), np.uint8)
    m[10:50, 3] = 1
    m[11:14, 40] = 1
    m[63, 63] = 1                                                                # long, short, long ... : negative differences
    yield m
    m = np.zeros((1, 40), np.uint8)
    m[0, ::3] = 1
    yield m
    yield (rng.rand(120, 90) < 0.5).astype(np.uint8)
    yield (rng.rand(200, 333) < 0.02).astype(np.uint8)
    blob = np.zeros((100, 100), np.uint8)
    yy, xx = np.mgrid[:100, :100]
    blob[(yy - 40) ** 2 + (xx - 55) ** 2 < 900] = 1                              # a disc, as a real mask would be
    yield blob


def test_rle_against_an_independent_codec():
    for mask in _adversarial_masks():
        h, w = mask.shape
        counts = _indep_counts(mask)
        assert R.rle_counts(mask) == counts
        s = _indep_string(counts)
        assert R.rle_to_string(counts) == s
        assert all(48 <= ord(ch) < 48 + 64 for ch in s)
        assert np.array_equal(_indep_decode_string(R.mask_encode(mask)["counts"], h, w), mask)   # decode(encode(m)) == m
        assert R.rle_from_string(s) == counts


def _dense_paste(mask_mm, box, im_h, im_w, thresh=0.5):
    """The paste of core/test.py:826-845 as dense tensor operations: zero-pad, bilinear interpolation to the box
    (half-pixel centres, no anti-aliasing), threshold, scatter the pixels that fall inside the image."""
    import torch
    import torch.nn.functional as F

    m = mask_mm.shape[0]
    padded = F.pad(torch.from_numpy(mask_mm)[None, None], (1, 1, 1, 1))
    w, h = max(int(box[2] - box[0] + 1), 1), max(int(box[3] - box[1] + 1), 1)
    soft = F.interpolate(padded, size=(h, w), mode="bilinear", align_corners=False)[0, 0]
    ys, xs = torch.meshgrid(torch.arange(h) + int(box[1]), torch.arange(w) + int(box[0]), indexing="ij")
    inside = (ys >= 0) & (ys < im_h) & (xs >= 0) & (xs < im_w)
    im = torch.zeros((im_h, im_w), dtype=torch.uint8)
    im.index_put_((ys[inside], xs[inside]), (soft > thresh)[inside].to(torch.uint8))
    return im.numpy(), soft.numpy()


def test_paste_against_a_dense_interpolate_threshold_scatter():
    rng = np.random.RandomState(13)
    im_h, im_w = 427, 640
    boxes = np.array([[100, 120, 300, 420], [-40, -30, 90, 60], [560, 380, 700, 500], [0, 0, 639, 426], [320, -10, 330, 500],
                      [200, 100, 200, 100], [900, 100, 950, 200], [10, 420, 300, 426], [33, 44, 61, 57]], np.int32)
    for i, box in enumerate(boxes):
        yy, xx = np.mgrid[:28, :28]
        mask = np.exp(-(((yy - 13.0 - i) ** 2) + (xx - 14.0 + i) ** 2) / (30.0 + 10 * i)).astype(np.float32)
        mask = np.clip(mask + 0.1 * rng.randn(28, 28).astype(np.float32), 0, 1)
        got = R.paste_mask(mask, box, im_h, im_w)
        want, soft = _dense_paste(mask, box, im_h, im_w)
        # pixels whose interpolated value is within fp32 noise of the threshold may fall on either side (OpenCV forms the
        # source coordinate in double, interpolate in fp32): everywhere else the two must agree exactly
        x0, y0 = int(box[0]), int(box[1])
        near = np.zeros((im_h, im_w), bool)
        ys, xs = np.nonzero(np.abs(soft - 0.5) < 1e-4)
        ok = (ys + y0 >= 0) & (ys + y0 < im_h) & (xs + x0 >= 0) & (xs + x0 < im_w)
        near[ys[ok] + y0, xs[ok] + x0] = True
        assert np.array_equal(got[~near], want[~near]), "box %d" % i
        assert near.sum() <= 0.002 * max(got.size, 1)
        assert got.sum() > 0 or i in (5, 6) or want.sum() == 0
