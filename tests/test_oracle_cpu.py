"""CPU tests (no GPU): the oracle against (a) the golden vectors produced by the reference's own code,
(b) the reference-on-host libraries when they are present, (c) analytic known answers.

These pin the oracle; the -m gpu tests then compare the HIP kernels with the oracle.
"""
import numpy as np
import pytest

from conftest import load_golden
from detectron_pytorch_amd import synthetic as syn


# ---- (a) golden vectors -------------------------------------------------------------------------
def test_roi_align_matches_golden(oracle_mod):
    g = load_golden("roi_align.npz")
    feat, rois, scale = g["feat"], g["rois"], float(g["scale"])
    for key in [k[4:] for k in g.files if k.startswith("fwd_")]:
        sr, res = int(key.split("_")[0][2:]), int(key.split("_")[1][1:])
        out = oracle_mod.roi_align_forward(feat, rois, res, res, scale, sr)
        assert np.array_equal(out, g["fwd_" + key]), key
        grad = oracle_mod.roi_align_backward(g["gtop_" + key], rois, feat.shape, scale, sr)
        assert np.array_equal(grad, g["bwd_" + key]), key


def test_roi_align_legacy_matches_golden(oracle_mod):
    g = load_golden("roi_align_legacy.npz")
    out = oracle_mod.roi_align_legacy_forward(g["feat"], g["rois"], 7, 7, float(g["scale"]))
    assert np.array_equal(out, g["fwd"])
    grad = oracle_mod.roi_align_legacy_backward(g["gtop"], g["rois"], g["feat"].shape, float(g["scale"]))
    assert np.array_equal(grad, g["bwd"])


def test_roi_pool_matches_golden(oracle_mod):
    g = load_golden("roi_pool.npz")
    out, argmax = oracle_mod.roi_pool_forward(g["feat"], g["rois"], 7, 7, float(g["scale"]))
    assert np.array_equal(out, g["fwd"])
    assert np.array_equal(argmax, g["argmax"])
    grad = oracle_mod.roi_pool_backward(g["gtop"], g["rois"], argmax, g["feat"].shape, float(g["scale"]))
    assert np.array_equal(grad, g["bwd"])


def test_roi_crop_matches_golden(oracle_mod):
    g = load_golden("roi_crop.npz")
    assert np.array_equal(oracle_mod.roi_crop_forward(g["feat"], g["grid"]), g["fwd"])
    assert np.array_equal(oracle_mod.roi_crop_backward(g["feat"], g["grid"], g["gtop"]), g["bwd"])


def test_nms_matches_golden(oracle_mod):
    g = load_golden("nms.npz")
    n_checked = 0
    for key in g.files:
        if not key.startswith("cython_"):
            continue
        name, t = key[len("cython_"):].rsplit("_t", 1)
        dets, thresh = g["dets_" + name], int(t) / 100.0
        assert np.array_equal(oracle_mod.nms_cython(dets, thresh), g[key]), key
        sorted_dets, _ = syn.sort_by_score(dets)
        assert np.array_equal(oracle_mod.nms_gpu_semantics(sorted_dets, thresh), g["gpu_" + name + "_t" + t]), key
        n_checked += 1
    assert n_checked >= 8


def test_bbox_overlaps_matches_golden(oracle_mod):
    g = load_golden("bbox_overlaps.npz")
    assert np.array_equal(oracle_mod.bbox_overlaps(g["boxes"], g["query"]), g["overlaps"])


# ---- (b) reference-on-host, fresh random inputs ------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_equals_reference_on_host(oracle_mod, ref_mod, seed):
    n, c, h, w, scale = 2, 4, 19, 23, 1.0 / 8
    feat = syn.feature_map(n, c, h, w, seed=100 + seed)
    rois = syn.rois_adversarial(30, n, h, w, scale, seed=200 + seed)
    rng = np.random.RandomState(300 + seed)
    for sr in (2, 0, 1):
        a = oracle_mod.roi_align_forward(feat, rois, 7, 7, scale, sr)
        assert np.array_equal(a, ref_mod.roi_align_forward(feat, rois, 7, 7, scale, sr))
        gt = rng.randn(*a.shape).astype(np.float32)
        assert np.array_equal(oracle_mod.roi_align_backward(gt, rois, feat.shape, scale, sr),
                              ref_mod.roi_align_backward(gt, rois, feat.shape, scale, sr))
    a = oracle_mod.roi_align_legacy_forward(feat, rois, 5, 6, scale)
    assert np.array_equal(a, ref_mod.roi_align_legacy_forward(feat, rois, 5, 6, scale))
    gt = rng.randn(*a.shape).astype(np.float32)
    assert np.array_equal(oracle_mod.roi_align_legacy_backward(gt, rois, feat.shape, scale),
                          ref_mod.roi_align_legacy_backward(gt, rois, feat.shape, scale))
    a, am = oracle_mod.roi_pool_forward(feat, rois, 6, 5, scale)
    b, bm = ref_mod.roi_pool_forward(feat, rois, 6, 5, scale)
    assert np.array_equal(a, b) and np.array_equal(am, bm)
    gt = rng.randn(*a.shape).astype(np.float32)
    assert np.array_equal(oracle_mod.roi_pool_backward(gt, rois, am, feat.shape, scale),
                          ref_mod.roi_pool_backward(gt, rois, bm, feat.shape, scale))
    grid = syn.crop_grid(6, 5, 4, seed=400 + seed)
    a = oracle_mod.roi_crop_forward(feat, grid)
    assert np.array_equal(a, ref_mod.roi_crop_forward(feat, grid))
    gt = rng.randn(*a.shape).astype(np.float32)
    assert np.array_equal(oracle_mod.roi_crop_backward(feat, grid, gt), ref_mod.roi_crop_backward(feat, grid, gt)[0])


@pytest.mark.parametrize("gen,n,thresh", [(syn.boxes_uniform, 777, 0.5), (syn.boxes_clustered, 1500, 0.7),
                                          (syn.boxes_clustered, 129, 0.3), (syn.boxes_uniform, 64, 0.5)])
def test_oracle_nms_equals_reference_cython(oracle_mod, ref_mod, gen, n, thresh):
    dets = gen(n, seed=n)
    assert np.array_equal(oracle_mod.nms_cython(dets, thresh), ref_mod.cython_nms(dets, thresh))
    sorted_dets, _ = syn.sort_by_score(dets)
    assert np.array_equal(oracle_mod.nms_gpu_semantics(sorted_dets, thresh), ref_mod.nms_gpu(sorted_dets, thresh))


# ---- (c) analytic known answers --------------------------------------------
def test_roi_align_constant_map(oracle_mod):
    feat = np.full((1, 3, 20, 30), 1.75, np.float32)
    rois = np.array([[0, 8, 8, 72, 56], [0, 0, 0, 40, 40]], np.float32)
    out = oracle_mod.roi_align_forward(feat, rois, 7, 7, 0.25, 2)
    assert np.allclose(out, 1.75, atol=1e-6)


def test_roi_align_affine_field_is_exact_at_bin_centres(oracle_mod):
    h, w = 40, 50
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    feat = (0.5 * yy - 0.25 * xx + 3.0).astype(np.float32)[None, None]
    rois = np.array([[0, 16, 24, 80, 88]], np.float32)  # interior: 4..20 x 6..22 at scale 1/4
    out = oracle_mod.roi_align_forward(feat, rois, 4, 4, 0.25, 2)[0, 0]
    x1, y1, bw, bh = 4.0, 6.0, 16.0 / 4, 16.0 / 4
    for ph in range(4):
        for pw in range(4):
            cy, cx = y1 + (ph + 0.5) * bh, x1 + (pw + 0.5) * bw
            assert abs(out[ph, pw] - (0.5 * cy - 0.25 * cx + 3.0)) < 1e-4


def test_roi_align_backward_conserves_gradient_mass_for_interior_rois(oracle_mod):
    rois = np.array([[0, 20, 20, 100, 90], [1, 8, 8, 60, 60]], np.float32)
    gtop = np.random.RandomState(5).randn(2, 3, 7, 7).astype(np.float32)
    grad = oracle_mod.roi_align_backward(gtop, rois, (2, 3, 40, 50), 0.25, 2)
    assert np.allclose(grad.sum(), gtop.sum(), rtol=1e-4, atol=1e-3)
    # finite-difference check of one input element against the forward
    feat = np.random.RandomState(6).randn(2, 3, 40, 50).astype(np.float32)
    base = (oracle_mod.roi_align_forward(feat, rois, 7, 7, 0.25, 2) * gtop).sum()
    feat2 = feat.copy()
    feat2[0, 1, 10, 12] += 0.5
    fd = ((oracle_mod.roi_align_forward(feat2, rois, 7, 7, 0.25, 2) * gtop).sum() - base) / 0.5
    assert abs(fd - grad[0, 1, 10, 12]) < 1e-2


def test_roi_align_adaptive_grid_and_malformed_roi(oracle_mod):
    feat = syn.feature_map(1, 1, 16, 16, seed=3)
    # malformed RoI (x2 < x1): forced to 1x1 in feature units (roi_align_kernel.cu:85-86)
    bad = np.array([[0, 40, 40, 8, 8]], np.float32)
    good = np.array([[0, 40, 40, 44, 44]], np.float32)  # 10..11 at scale 1/4 -> exactly 1x1
    a = oracle_mod.roi_align_forward(feat, bad, 2, 2, 0.25, 0)
    b = oracle_mod.roi_align_forward(feat, good, 2, 2, 0.25, 0)
    assert np.array_equal(a, b)
    assert oracle_mod.roi_align_touched_pixels(good, 1, 16, 16, 2, 2, 0.25, 0) == 4


def test_nms_hand_cases(oracle_mod):
    # identical boxes: only the higher score survives
    d = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8]], np.float32)
    assert oracle_mod.nms_cython(d, 0.5).tolist() == [0]
    # touching boxes overlap by one pixel column under the +1 convention: IoU = 10/190
    d = np.array([[0, 0, 9, 9, 0.9], [9, 0, 18, 9, 0.8]], np.float32)
    iou = np.float32(10.0) / np.float32(190.0)
    assert oracle_mod.nms_cython(d, float(iou)).tolist() == [0]        # >= suppresses at equality
    s, _ = syn.sort_by_score(d)
    assert oracle_mod.nms_gpu_semantics(s, float(iou)).tolist() == [0, 1]  # strict > keeps both
    # unsorted input: result is ascending ORIGINAL indices
    d = np.array([[100, 100, 120, 120, 0.1], [0, 0, 9, 9, 0.5], [0, 0, 9, 9, 0.9]], np.float32)
    assert oracle_mod.nms_cython(d, 0.5).tolist() == [0, 2]
    # empty
    assert oracle_mod.nms_cython(np.zeros((0, 5), np.float32), 0.5).tolist() == []


def test_roi_pool_first_max_wins(oracle_mod):
    feat = np.zeros((1, 1, 8, 8), np.float32)
    feat[0, 0, 2, 3] = 5.0
    feat[0, 0, 2, 5] = 5.0  # tie later in the row-major scan
    rois = np.array([[0, 0, 0, 7, 7]], np.float32)
    out, argmax = oracle_mod.roi_pool_forward(feat, rois, 1, 1, 1.0)
    assert out[0, 0, 0, 0] == 5.0 and argmax[0, 0, 0, 0] == 2 * 8 + 3
    # RoI entirely outside: empty bins -> 0 / -1
    out, argmax = oracle_mod.roi_pool_forward(feat, np.array([[0, 20, 20, 30, 30]], np.float32), 2, 2, 1.0)
    assert not out.any() and (argmax == -1).all()


# ---- FPN level mapping / multilevel RoI blobs (host logic of roi_feature_transform) -----------------------------
def test_fpn_level_mapping_and_restore_permutation_match_reference_fixture():
    """tests/golden/fpn.npz was produced by importing the reference's lib/utils/fpn.py (generate.py:gen_fpn)."""
    from detectron_pytorch_amd import roi_xform

    g = load_golden("fpn.npz")
    rois = g["rois"]
    lvls = roi_xform.map_rois_to_fpn_levels(rois[:, 1:5], 2, 5)
    assert np.array_equal(lvls, g["levels"].astype(np.int64))
    blobs = roi_xform.add_multilevel_roi_blobs({}, "rois", rois, lvls, 2, 5)
    for key in ("rois_fpn2", "rois_fpn3", "rois_fpn4", "rois_fpn5", "rois_idx_restore_int32"):
        assert np.array_equal(blobs[key], g[key]), key
        assert blobs[key].dtype == g[key].dtype, key
    stacked = np.vstack([blobs["rois_fpn%d" % l] for l in range(2, 6)])
    assert np.array_equal(stacked[blobs["rois_idx_restore_int32"]], rois)


def test_roi_feature_transform_rejects_unknown_method_and_level_count():
    import torch
    from detectron_pytorch_amd import roi_xform

    with pytest.raises(AssertionError):
        roi_xform.roi_feature_transform(torch.zeros(1, 1, 2, 2), {"rois": np.zeros((0, 5), np.float32)}, method="Nope")
    with pytest.raises(AssertionError):
        roi_xform.roi_feature_transform([torch.zeros(1, 1, 2, 2)] * 3, {}, method="RoIAlign",
                                        spatial_scale=[1 / 32, 1 / 16, 1 / 8])


# ---- Soft-NMS ----------------------------------------------------------------------------------------------------
def _soft_nms_golden_cases():
    g = load_golden("soft_nms.npz")
    cfgs = g["cfgs"]
    for key in g.files:
        if not key.startswith("boxes_"):
            continue
        name, m, c = key[len("boxes_"):].rsplit("_", 2)
        sigma, nt, th = (float(v) for v in cfgs[int(c[1:])])
        yield key[len("boxes_"):], g["dets_" + name], sigma, nt, th, int(m[1:]), g[key], g["inds_" + key[len("boxes_"):]]


def test_soft_nms_matches_golden(oracle_mod):
    """Fixture produced by the reference's own cython_nms.soft_nms (tests/golden/generate.py:gen_soft_nms): same rows,
    same order, same bits -- including the gaussian re-scoring (exp in double) and the swap-with-last compaction."""
    n = 0
    for tag, dets, sigma, nt, th, method, boxes, inds in _soft_nms_golden_cases():
        ob, oi = oracle_mod.soft_nms(dets, sigma, nt, th, method)
        assert np.array_equal(ob, boxes) and np.array_equal(oi, inds), tag
        n += 1
    assert n == 27


def test_soft_nms_matches_reference_build(oracle_mod, ref_mod):
    for gen, n in ((syn.boxes_uniform, 1), (syn.boxes_uniform, 400), (syn.boxes_clustered, 1000)):
        dets = gen(n, seed=n)
        for method in (0, 1, 2):
            rb, ri = ref_mod.cython_soft_nms(dets, 0.5, 0.3, 0.01, method)
            ob, oi = oracle_mod.soft_nms(dets, 0.5, 0.3, 0.01, method)
            assert np.array_equal(rb, ob) and np.array_equal(np.asarray(ri), oi), (gen.__name__, n, method)


def test_soft_nms_known_answers(oracle_mod):
    # two identical boxes + one disjoint: hard mode drops the duplicate (score 0 < threshold), keeps pick order
    dets = np.array([[0, 0, 9, 9, 0.5], [0, 0, 9, 9, 0.9], [20, 20, 29, 29, 0.7]], np.float32)
    boxes, inds = oracle_mod.soft_nms(dets, 0.5, 0.3, 0.001, 0)
    assert inds.tolist() == [1, 2] and boxes[:, 4].tolist() == [np.float32(0.9), np.float32(0.7)]
    # linear: the duplicate (IoU 1) is re-scored to 0.5 * (1 - 1) = 0 and dropped; with a lower overlap it survives
    dets = np.array([[0, 0, 9, 9, 0.9], [0, 5, 9, 14, 0.8]], np.float32)  # IoU = 50 / 150
    boxes, inds = oracle_mod.soft_nms(dets, 0.5, 0.3, 0.001, 1)
    assert inds.tolist() == [0, 1]
    assert boxes[1, 4] == np.float32(np.float32(1.0 - np.float64(np.float32(50.0) / np.float32(150.0))) * np.float32(0.8))
    # a box that does not overlap is never dropped, even below the threshold (the test sits inside `if ih > 0`)
    dets = np.array([[0, 0, 9, 9, 0.9], [50, 50, 59, 59, 0.0001]], np.float32)
    assert oracle_mod.soft_nms(dets, 0.5, 0.3, 0.001, 1)[1].tolist() == [0, 1]


# ---- per-class detection post-processing (core/test.py:732-790) ---------------------------------------------------
DETECTION_CASES = [(name, tag, soft, method) for name in ("c21", "c81")
                   for tag, soft, method in (("hard", False, "linear"), ("linear", True, "linear"),
                                             ("gaussian", True, "gaussian"))]


def test_detection_postprocess_matches_golden(oracle_mod):
    """detection.npz holds the outputs of the reference's own function source (generate.py:gen_detection)."""
    from oracle import postprocess

    g = load_golden("detection.npz")
    for name, tag, soft, method in DETECTION_CASES:
        key = "%s_%s" % (name, tag)
        s, b, cls_boxes = postprocess.box_results_with_nms_and_limit(g["scores_" + name], g["boxes_" + name],
                                                                    soft_nms=soft, soft_nms_method=method)
        assert np.array_equal(s, g["out_scores_" + key]) and np.array_equal(b, g["out_boxes_" + key]), key
        assert np.array_equal(np.array([len(c) for c in cls_boxes]), g["cls_counts_" + key]), key
        assert np.array_equal(np.vstack(cls_boxes[1:]), g["cls_rows_" + key]), key


def test_box_voting_matches_golden(oracle_mod):
    """box_voting.npz holds the outputs of the reference's own utils.boxes.box_voting source (all six scoring methods) and
    of its box_results_with_nms_and_limit with TEST.BBOX_VOTE.ENABLED (generate.py:gen_box_voting): the restatement is
    bit-identical."""
    from oracle import postprocess

    g = load_golden("box_voting.npz")
    top, alld = g["top_dets"], g["all_dets"]
    for method in ("ID", "TEMP_AVG", "AVG", "IOU_AVG", "GENERALIZED_AVG", "QUASI_SUM"):
        out = postprocess.box_voting(top, alld, float(g["thresh"]), scoring_method=method, beta=float(g["beta"]))
        assert np.array_equal(out.astype(np.float32), g["voted_" + method]), method
    assert np.array_equal(postprocess.box_voting(top, alld, 0.5).astype(np.float32), g["voted_loose_ID"])
    assert (np.abs(g["voted_loose_ID"][:, :4] - top[:, :4]).max(axis=1) > 1e-3).sum() > 30   # voting really moves boxes
    for tag, soft, method in (("hard_ID", False, "ID"), ("linear_IOU_AVG", True, "IOU_AVG")):
        s, b, cls_boxes = postprocess.box_results_with_nms_and_limit(g["det_in_scores"], g["det_in_boxes"], soft_nms=soft,
                                                                     bbox_vote=True, bbox_vote_method=method)
        assert np.array_equal(s.astype(np.float32), g["det_scores_" + tag])
        assert np.array_equal(b.astype(np.float32), g["det_boxes_" + tag])
        assert np.array_equal(np.array([len(c) for c in cls_boxes]), g["det_counts_" + tag])


def test_detection_postprocess_without_limit_and_empty_classes(oracle_mod):
    from oracle import postprocess

    scores, boxes = syn.detection_head_outputs(60, 11, seed=1)
    scores[:, 3] = 0.0                                      # a class nobody scores for: cls_boxes[3] is (0, 5)
    s, b, cls_boxes = postprocess.box_results_with_nms_and_limit(scores, boxes, detections_per_im=0)
    assert cls_boxes[3].shape == (0, 5) and cls_boxes[0] == []
    assert len(s) == sum(len(c) for c in cls_boxes[1:]) and b.shape == (len(s), 4)
    for j in range(1, 11):                                   # per class: exactly cython_nms over the thresholded rows
        inds = np.where(scores[:, j] > 0.05)[0]
        dets = np.hstack((boxes[inds, 4 * j:4 * j + 4], scores[inds, j][:, None])).astype(np.float32)
        keep = oracle_mod.nms_cython(dets, 0.5) if len(dets) else []
        assert np.array_equal(cls_boxes[j], dets[keep, :])


def test_touched_pixel_count_of_the_workload_descriptor_matches_oracle(oracle_mod):
    """bench.py derives the algorithmic bytes of the roofline from synthetic.roi_align_touched_pixels (numpy): it must
    count exactly what the oracle's sampling arithmetic touches."""
    cases = [(syn.rois_canonical(512, 1, seed=0), 1, 200, 336, 7, 0.25, 2),
             (syn.rois_adversarial(300, 2, 50, 84, 1.0 / 16, seed=1), 2, 50, 84, 14, 1.0 / 16, 0),
             (syn.rois_adversarial(200, 3, 13, 21, 1.0 / 32, seed=2), 3, 13, 21, 7, 1.0 / 32, 3)]
    for rois, b, h, w, res, scale, sr in cases:
        assert syn.roi_align_touched_pixels(rois, b, h, w, res, res, scale, sr) == \
            oracle_mod.roi_align_touched_pixels(rois, b, h, w, res, res, scale, sr)


# ---- RPN proposal generation (generate_proposals.py / generate_anchors.py) -----------------------------------------
def test_anchors_match_reference_fixture_and_known_answer():
    from detectron_pytorch_amd import generate_proposals as gp
    from oracle import proposals

    g = load_golden("proposals.npz")
    for fn in (gp.generate_anchors, proposals.generate_anchors):
        assert np.array_equal(fn(4, (32,), (0.5, 1, 2)), g["anchors_s4"])
        assert np.array_equal(fn(), g["anchors_s16_default"])
        # stride 16, scales 8/16/32: the table in the reference's comment (generate_anchors.py:26-51, 1-based matlab
        # coordinates) shifted by the -1 of the code's 0-based base anchor (:78)
        table = fn(16, (128, 256, 512), (0.5, 1, 2))
        assert table[0].tolist() == [-84.0, -40.0, 99.0, 55.0] and table[-1].tolist() == [-168.0, -344.0, 183.0, 359.0]


def test_generate_proposals_oracle_matches_reference_fixture(oracle_mod):
    """proposals.npz: outputs of the reference's GenerateProposalsOp executed from its own source (generate.py)."""
    from oracle import proposals

    g = load_golden("proposals.npz")
    for name in ("p4", "p5", "p3min"):
        stride, size, h, w, pre, post, min_size = (int(v) for v in g["cfg_" + name])
        anchors = proposals.generate_anchors(stride, (size,), (0.5, 1, 2))
        scores, deltas = syn.rpn_head_outputs(2, anchors.shape[0], h, w, seed=stride)
        rois, probs = proposals.generate_proposals(scores, deltas, g["im_info_" + name], anchors, 1.0 / stride, pre, post,
                                                   0.7, min_size)
        assert np.array_equal(rois, g["rois_" + name]) and np.array_equal(probs, g["probs_" + name]), name


# ---- mask targets from polygons (roi_data/mask_rcnn.py:34-107, utils/segms.py; pycocotools' rasteriser restated) ------
def _segms_from_fixture(g, tag):
    pts, ps, ins = g[tag + "_points"], g[tag + "_poly_start"], g[tag + "_inst_start"]
    return [[pts[ps[p]:ps[p + 1]].reshape(-1) for p in range(ins[i], ins[i + 1])] for i in range(len(ins) - 1)]


def test_mask_targets_from_polygons_match_golden(oracle_mod):
    """The line-by-line restatement of utils/segms.py in oracle/mask_targets.py (what the GPU tests compare the HIP kernel with)
    against the fixture the reference's own add_mask_rcnn_blobs + segms.py produced."""
    from oracle import mask_targets as segms

    g = load_golden("mask_targets.npz")
    for tag in ("a", "b"):
        m, polys = int(g[tag + "_resolution"]), _segms_from_fixture(g, tag)
        boxes_from_polys = segms.polys_to_boxes(polys)
        assert np.array_equal(boxes_from_polys, g[tag + "_boxes_from_polys"])
        fg = g[tag + "_labels"] > 0
        rois = g[tag + "_sampled_boxes"][fg]
        inst = oracle_mod.bbox_overlaps(rois, boxes_from_polys).argmax(axis=1)
        masks = np.stack([segms.polys_to_mask_wrt_box(polys[i], r, m).reshape(-1) for r, i in zip(rois, inst)])
        assert np.array_equal(masks.astype(np.int8), g[tag + "_masks_int32"])
        assert np.array_equal(g[tag + "_roi_has_mask"], fg.astype(np.int32))


def test_polygon_rasteriser_known_answers(oracle_mod):
    """pycocotools' rule, derived by hand from maskApi.c rleFrPoly: vertices go to a grid of 5 samples per pixel, a pixel
    column is crossed where the boundary passes its centre sample (5 x + 2), rows likewise -- "the pixels whose centres
    the polygon covers", on that grid."""
    from oracle import mask_targets as segms

    def rect(x0, y0, x1, y1):
        return [x0, y0, x1, y0, x1, y1, x0, y1]

    want = np.zeros((5, 5), np.float32)
    want[1:3, 1:3] = 1
    assert np.array_equal(segms.polys_to_mask([rect(1, 1, 3, 3)], 5, 5), want)             # whole pixels of an aligned rectangle
    assert segms.polys_to_mask([rect(0, 0, 7, 5)], 5, 7).all()                              # the whole image
    assert segms.polys_to_mask([rect(-10, -3, 40, 30)], 5, 7).all()                         # ... and beyond its border
    want[:] = 0
    want[1:4, 1:4] = 1
    assert np.array_equal(segms.polys_to_mask([rect(0.5, 0.5, 3.5, 3.5)], 5, 5), want)     # centres 1.5, 2.5 (, 3.5: the 5x grid
    want[:] = 0                                                                            # rounds 17.5 + .5 up) inside
    want[1:3, 1:3] = 1
    assert np.array_equal(segms.polys_to_mask([rect(0.6, 0.6, 2.9, 2.9)], 5, 5), want)
    assert not segms.polys_to_mask([rect(1.6, 1.6, 2.4, 2.4)], 5, 5).any()                  # covers no pixel centre
    tri = segms.polys_to_mask([[0.5, 0.5, 4.5, 0.5, 2.5, 4.5]], 6, 6).astype(int)          # worked through by hand
    assert np.array_equal(tri, np.array([[0, 0, 0, 0, 0, 0], [0, 1, 1, 1, 0, 0], [0, 0, 1, 1, 0, 0], [0, 0, 1, 0, 0, 0],
                                         [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]))
    # two polygons of one instance are OR-ed (segms.py:117-118), overlapping or not
    both = segms.polys_to_mask([rect(0, 0, 2, 2), rect(1, 1, 4, 4)], 5, 5)
    assert both.sum() == 4 + 9 - 1 and both.max() == 1
    # in the frame of a box: the reference's float32 shift-and-scale (segms.py:104-112)
    m = segms.polys_to_mask_wrt_box([rect(10, 10, 30, 30)], np.array([5, 5, 25, 25], np.float32), 8)
    want8 = np.zeros((8, 8), np.float32)
    want8[2:, 2:] = 1
    assert np.array_equal(m, want8)


def test_polygon_rasteriser_properties(oracle_mod):
    """Size-independent properties of the published algorithm: the mask does not depend on the vertex the outline starts
    from nor on its orientation, repeated vertices change nothing, the area equals the polygon's up to its perimeter, and it
    agrees with an exact even-odd test of the pixel centres except on boundary pixels."""
    from oracle import mask_targets as segms

    rng = np.random.RandomState(0)
    for trial in range(12):
        k = rng.randint(5, 40)
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        rad = rng.uniform(0.5, 1.0, k) * rng.uniform(20, 120)
        x, y = 150 + rad * np.cos(ang), 140 + 0.8 * rad * np.sin(ang)
        pts = np.round(np.stack([x, y], 1), 2)
        base = segms.polys_to_mask([pts.reshape(-1)], 300, 310)
        assert np.array_equal(segms.polys_to_mask([np.roll(pts, rng.randint(1, k), axis=0).reshape(-1)], 300, 310), base)
        assert np.array_equal(segms.polys_to_mask([pts[::-1].reshape(-1)], 300, 310), base)
        assert np.array_equal(segms.polys_to_mask([np.repeat(pts, 1 + (rng.rand(k) < 0.3), axis=0).reshape(-1)], 300, 310), base)
        xs, ys = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
        area = 0.5 * abs(np.dot(xs, np.roll(ys, -1)) - np.dot(ys, np.roll(xs, -1)))
        perimeter = np.hypot(xs - np.roll(xs, -1), ys - np.roll(ys, -1)).sum()
        assert abs(base.sum() - area) <= 0.25 * perimeter, (trial, base.sum(), area, perimeter)
        cy, cx = np.mgrid[0:300, 0:310] + 0.5                                  # even-odd rule at the pixel centres
        inside = np.zeros((300, 310), bool)
        for j in range(k):
            x0, y0, x1, y1 = xs[j], ys[j], xs[(j + 1) % k], ys[(j + 1) % k]
            hit = ((y0 <= cy) != (y1 <= cy))
            with np.errstate(divide="ignore", invalid="ignore"):
                xc = x0 + (cy - y0) * (x1 - x0) / (y1 - y0)
            inside ^= hit & (cx < xc)
        assert (inside != (base > 0)).sum() <= 0.75 * perimeter, (trial, (inside != (base > 0)).sum(), perimeter)
