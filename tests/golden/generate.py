#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz FROM THE REFERENCE ITSELF.

Runs only in the build container: it needs oracle/_ref (the reference's .cu kernels compiled for
the host and its Cython NMS/IoU modules, built by oracle/build_ref.py from /root/reference).
The committed .npz files let the GPU box -- where /root/reference does not exist -- check both the
oracle and the HIP kernels against outputs of the reference's own code.

    python tests/golden/generate.py        # rewrites every fixture deterministically

Each fixture stores inputs and reference outputs; sizes are kept small (total < 3 MB).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from detectron_pytorch_amd import synthetic as syn  # noqa: E402
from oracle import ref  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))


def gen_roi_align():
    n, c, h, w = 2, 6, 25, 42
    scale = 1.0 / 32
    feat = syn.feature_map(n, c, h, w, seed=11)
    rois = syn.rois_adversarial(24, n, h, w, scale, seed=12)
    out = {}
    for sr in (2, 0, 3):
        for res in (7, 14):
            if res == 14 and sr != 2:
                continue
            key = "sr%d_r%d" % (sr, res)
            top = ref.roi_align_forward(feat, rois, res, res, scale, sr)
            gtop = np.random.RandomState(13 + sr + res).randn(*top.shape).astype(np.float32)
            out["fwd_" + key] = top
            out["gtop_" + key] = gtop
            out["bwd_" + key] = ref.roi_align_backward(gtop, rois, feat.shape, scale, sr)
    save("roi_align.npz", feat=feat, rois=rois, scale=np.float32(scale), **out)


def gen_roi_align_legacy():
    n, c, h, w = 2, 5, 20, 30
    scale = 1.0 / 16
    feat = syn.feature_map(n, c, h, w, seed=21)
    rois = syn.rois_adversarial(20, n, h, w, scale, seed=22)
    rois[:, 0] = np.clip(rois[:, 0], 0, n - 1)
    top = ref.roi_align_legacy_forward(feat, rois, 7, 7, scale)
    gtop = np.random.RandomState(23).randn(*top.shape).astype(np.float32)
    bwd = ref.roi_align_legacy_backward(gtop, rois, feat.shape, scale)
    save("roi_align_legacy.npz", feat=feat, rois=rois, scale=np.float32(scale), fwd=top, gtop=gtop, bwd=bwd)


def gen_roi_pool():
    n, c, h, w = 2, 6, 25, 42
    scale = 1.0 / 16
    feat = syn.feature_map(n, c, h, w, seed=31)
    # plant exact ties so the first-max-wins rule (roi_pooling_kernel.cu:83) is exercised
    feat[:, :, ::3, ::4] = np.float32(2.5)
    rois = syn.rois_adversarial(24, n, h, w, scale, seed=32)
    top, argmax = ref.roi_pool_forward(feat, rois, 7, 7, scale)
    gtop = np.random.RandomState(33).randn(*top.shape).astype(np.float32)
    bwd = ref.roi_pool_backward(gtop, rois, argmax, feat.shape, scale)
    save("roi_pool.npz", feat=feat, rois=rois, scale=np.float32(scale), fwd=top, argmax=argmax, gtop=gtop, bwd=bwd)


def gen_roi_crop():
    n, c, h, w = 2, 5, 18, 27
    feat = syn.feature_map(n, c, h, w, seed=41)
    grid = syn.crop_grid(8, 7, 7, seed=42)
    top = ref.roi_crop_forward(feat, grid)
    gtop = np.random.RandomState(43).randn(*top.shape).astype(np.float32)
    bwd, ggrid = ref.roi_crop_backward(feat, grid, gtop)
    assert not ggrid.any()
    save("roi_crop.npz", feat=feat, grid=grid, fwd=top, gtop=gtop, bwd=bwd)


def gen_nms():
    out = {}
    cases = [("uniform1000", syn.boxes_uniform(1000, seed=0), (0.5, 0.7)),
             ("clustered1000", syn.boxes_clustered(1000, seed=0), (0.5, 0.7)),
             ("uniform2000", syn.boxes_uniform(2000, seed=1), (0.7,)),
             ("clustered300", syn.boxes_clustered(300, seed=2, centres=12), (0.3, 0.5)),
             ("uniform65", syn.boxes_uniform(65, seed=3), (0.5,)),
             ("single", syn.boxes_uniform(1, seed=4), (0.5,))]
    for name, dets, threshes in cases:
        out["dets_" + name] = dets
        sorted_dets, _ = syn.sort_by_score(dets)
        for t in threshes:
            tag = "%s_t%02d" % (name, int(round(t * 100)))
            out["cython_" + tag] = np.asarray(ref.cython_nms(dets, t), dtype=np.int64)
            out["gpu_" + tag] = ref.nms_gpu(sorted_dets, t)
    save("nms.npz", **out)
    boxes = syn.boxes_uniform(300, seed=5)[:, :4]
    query = syn.boxes_clustered(40, seed=6)[:, :4]
    save("bbox_overlaps.npz", boxes=boxes, query=query, overlaps=ref.cython_bbox_overlaps(boxes, query))


def gen_soft_nms():
    """utils.cython_nms.soft_nms of the reference (oracle/_ref build of lib/utils/cython_nms.pyx)."""
    out = {}
    for name, dets in (("uniform250", syn.boxes_uniform(250, seed=7)), ("clustered250", syn.boxes_clustered(250, seed=8)),
                       ("uniform70", syn.boxes_uniform(70, seed=9))):
        out["dets_" + name] = dets
        for method in (0, 1, 2):
            for ci, (sigma, nt, th) in enumerate(((0.5, 0.3, 0.001), (0.5, 0.3, 0.05), (0.3, 0.5, 0.2))):
                boxes, inds = ref.cython_soft_nms(dets, sigma, nt, th, method)
                tag = "%s_m%d_c%d" % (name, method, ci)
                out["boxes_" + tag] = np.asarray(boxes, dtype=np.float32)
                out["inds_" + tag] = np.asarray(inds, dtype=np.int64)
    out["cfgs"] = np.array([[0.5, 0.3, 0.001], [0.5, 0.3, 0.05], [0.3, 0.5, 0.2]], dtype=np.float32)
    save("soft_nms.npz", **out)


def _reference_box_results():
    """The reference's box_results_with_nms_and_limit, executed from its own source lines (lib/core/test.py): the module
    cannot be imported here (cv2, pycocotools, the cffi extensions), so the function's text is compiled as it stands
    against a namespace with numpy, a cfg carrying the values it reads, and utils.boxes' nms / soft_nms bodies bound to
    the reference's own cython build (oracle/_ref)."""
    import re
    import types

    src = open("/root/reference/lib/core/test.py").read()
    body = re.search(r"^def box_results_with_nms_and_limit\(.*?(?=^def )", src, re.S | re.M).group(0)
    bsrc = open("/root/reference/lib/utils/boxes.py").read()
    nms_src = re.search(r"^def nms\(dets, thresh\):.*?(?=^def )", bsrc, re.S | re.M).group(0)
    soft_src = re.search(r"^def soft_nms\(.*?(?=^def |\Z)", bsrc, re.S | re.M).group(0)
    box_utils = types.ModuleType("box_utils")
    box_utils.np = np
    box_utils.cython_nms = ref._mod("cython_nms")
    exec(compile(nms_src + soft_src, "/root/reference/lib/utils/boxes.py", "exec"), box_utils.__dict__)

    def make(cfg):
        ns = {"np": np, "cfg": cfg, "box_utils": box_utils}
        exec(compile(body, "/root/reference/lib/core/test.py", "exec"), ns)
        return ns["box_results_with_nms_and_limit"]

    return make, types


def gen_detection():
    make, types = _reference_box_results()
    out = {}
    cases = {"c21": syn.detection_head_outputs(300, 21, seed=3), "c81": syn.detection_head_outputs(400, 81, seed=4)}
    for name, (scores, boxes) in cases.items():
        out["scores_" + name], out["boxes_" + name] = scores, boxes
        for tag, soft, method in (("hard", False, "linear"), ("linear", True, "linear"), ("gaussian", True, "gaussian")):
            cfg = types.SimpleNamespace(
                MODEL=types.SimpleNamespace(NUM_CLASSES=scores.shape[1]),
                TEST=types.SimpleNamespace(SCORE_THRESH=0.05, NMS=0.5, DETECTIONS_PER_IM=100,
                                           SOFT_NMS=types.SimpleNamespace(ENABLED=soft, METHOD=method, SIGMA=0.5),
                                           BBOX_VOTE=types.SimpleNamespace(ENABLED=False)))
            s, b, cls_boxes = make(cfg)(scores, boxes)
            key = "%s_%s" % (name, tag)
            out["out_scores_" + key], out["out_boxes_" + key] = s, b
            out["cls_counts_" + key] = np.array([len(c) for c in cls_boxes], dtype=np.int64)
            out["cls_rows_" + key] = np.vstack([c for c in cls_boxes[1:]]).astype(np.float32)
    save("detection.npz", **out)


def gen_box_voting():
    """utils.boxes.box_voting of the reference, its own source text executed with bbox_overlaps bound to the reference's
    cython_bbox build (oracle/_ref); and box_results_with_nms_and_limit with TEST.BBOX_VOTE.ENABLED (core/test.py:766-773)."""
    import re
    import types

    bsrc = open("/root/reference/lib/utils/boxes.py").read()
    body = re.search(r"^def box_voting\(.*?(?=^def )", bsrc, re.S | re.M).group(0)
    ns = {"np": np, "bbox_overlaps": ref._mod("cython_bbox").bbox_overlaps}
    exec(compile(body, "/root/reference/lib/utils/boxes.py", "exec"), ns)
    box_voting = ns["box_voting"]
    rng = np.random.RandomState(5)
    all_dets = syn.boxes_clustered(400, seed=21).astype(np.float32)
    all_dets[:, 4] = rng.uniform(0.05, 0.99, 400).astype(np.float32)
    top = all_dets[rng.choice(400, 60, replace=False)].copy()
    out = {"all_dets": all_dets, "top_dets": top, "thresh": np.float32(0.8), "beta": np.float32(1.5)}
    for method in ("ID", "TEMP_AVG", "AVG", "IOU_AVG", "GENERALIZED_AVG", "QUASI_SUM"):
        out["voted_" + method] = box_voting(top, all_dets, 0.8, scoring_method=method, beta=1.5).astype(np.float32)
    out["voted_loose_ID"] = box_voting(top, all_dets, 0.5).astype(np.float32)
    # the detection post-processing with voting switched on (the call site passes no beta: box_voting's default 1.0)
    make, types_ = _reference_box_results()
    make_voting = make
    scores, boxes = syn.detection_head_outputs(300, 21, seed=3)
    for tag, soft, vote_method in (("hard_ID", False, "ID"), ("linear_IOU_AVG", True, "IOU_AVG")):
        cfg = types.SimpleNamespace(
            MODEL=types.SimpleNamespace(NUM_CLASSES=21),
            TEST=types.SimpleNamespace(SCORE_THRESH=0.05, NMS=0.5, DETECTIONS_PER_IM=100,
                                       SOFT_NMS=types.SimpleNamespace(ENABLED=soft, METHOD="linear", SIGMA=0.5),
                                       BBOX_VOTE=types.SimpleNamespace(ENABLED=True, VOTE_TH=0.8, SCORING_METHOD=vote_method)))
        fn = make_voting(cfg)
        fn.__globals__["box_utils"].box_voting = box_voting
        s, b, cls_boxes = fn(scores, boxes)
        out["det_scores_" + tag], out["det_boxes_" + tag] = s.astype(np.float32), b.astype(np.float32)
        out["det_counts_" + tag] = np.array([len(c) for c in cls_boxes], dtype=np.int64)
    out["det_in_scores"], out["det_in_boxes"] = scores, boxes
    save("box_voting.npz", **out)


def _reference_generate_proposals():
    """The reference's GenerateProposalsOp and generate_anchors, executed from their own source text.
    generate_anchors.py uses np.float (removed from numpy); the same one-token patch as for cython_nms.pyx
    (np.float -> float, no arithmetic touched) lets it run.  generate_proposals.py runs as it stands against a cfg
    carrying the values it reads and a utils.boxes namespace made of the reference's own bbox_transform /
    clip_tiled_boxes / nms bodies (the latter bound to its cython build in oracle/_ref)."""
    import re
    import types

    lib = "/root/reference/lib"
    ga_src = open(lib + "/modeling/generate_anchors.py").read().replace("np.float)", "float)")
    ga = types.ModuleType("generate_anchors")
    exec(compile(ga_src, lib + "/modeling/generate_anchors.py", "exec"), ga.__dict__)
    bsrc = open(lib + "/utils/boxes.py").read()

    def fn(name):
        return re.search(r"^def %s\(.*?(?=^def |\Z)" % name, bsrc, re.S | re.M).group(0)

    def make(train_cfg):
        class Cfg(dict):  # the reference reads cfg both as attributes and as cfg['TRAIN'] / cfg['TEST']
            __getattr__ = dict.__getitem__

        cfg = Cfg(BBOX_XFORM_CLIP=np.log(1000. / 16.), TRAIN=train_cfg, TEST=train_cfg)
        box_utils = types.ModuleType("box_utils")
        box_utils.np, box_utils.cfg, box_utils.cython_nms = np, cfg, ref._mod("cython_nms")
        exec(compile(fn("bbox_transform") + fn("clip_tiled_boxes") + fn("nms"), lib + "/utils/boxes.py", "exec"),
             box_utils.__dict__)
        gp_src = open(lib + "/modeling/generate_proposals.py").read()
        gp_src = gp_src.replace("from core.config import cfg", "").replace("import utils.boxes as box_utils", "")
        gp = types.ModuleType("generate_proposals")
        gp.cfg, gp.box_utils = cfg, box_utils
        exec(compile(gp_src, lib + "/modeling/generate_proposals.py", "exec"), gp.__dict__)
        return gp.GenerateProposalsOp

    return ga.generate_anchors, make, types


def gen_proposals():
    import torch

    generate_anchors, make, types = _reference_generate_proposals()
    out = {}
    out["anchors_s4"] = generate_anchors(stride=4, sizes=(32,), aspect_ratios=(0.5, 1, 2))
    out["anchors_s16_default"] = generate_anchors()
    cases = {"p4": (16, (128,), 50, 84, 300, 120, 0), "p5": (32, (256,), 25, 42, 12000, 2000, 0),
             "p3min": (8, (64,), 40, 60, 500, 200, 16)}
    for name, (stride, sizes, h, w, pre, post, min_size) in cases.items():
        anchors = generate_anchors(stride=stride, sizes=sizes, aspect_ratios=(0.5, 1, 2))
        scores, deltas = syn.rpn_head_outputs(2, anchors.shape[0], h, w, seed=stride)
        im_info = np.array([[h * stride, w * stride, 1.0], [h * stride - 37, w * stride - 50, 1.6]], np.float32)
        op_cls = make(types.SimpleNamespace(RPN_PRE_NMS_TOP_N=pre, RPN_POST_NMS_TOP_N=post, RPN_NMS_THRESH=0.7,
                                            RPN_MIN_SIZE=min_size))
        op = op_cls(anchors, 1.0 / stride)
        rois, probs = op.forward(torch.from_numpy(scores), torch.from_numpy(deltas), torch.from_numpy(im_info))
        out["cfg_" + name] = np.array([stride, sizes[0], h, w, pre, post, min_size], np.int64)
        out["im_info_" + name] = im_info
        out["rois_" + name], out["probs_" + name] = rois.astype(np.float32), probs.astype(np.float32)
    save("proposals.npz", **out)


def _reference_fpn_module():
    """Import the reference's lib/utils/fpn.py as it lies under /root/reference.  Its two imports are satisfied
    without the reference's config machinery: `core.config.cfg` by a namespace carrying the two defaults it reads
    (lib/core/config.py: FPN.ROI_CANONICAL_SCALE=224, FPN.ROI_CANONICAL_LEVEL=4) and `utils.boxes` by the
    reference's own boxes_area source lines, executed from the reference file."""
    import importlib.util
    import re
    import types
    import warnings

    lib = "/root/reference/lib"
    cfg = types.SimpleNamespace(FPN=types.SimpleNamespace(ROI_CANONICAL_SCALE=224, ROI_CANONICAL_LEVEL=4))
    src = open(os.path.join(lib, "utils", "boxes.py")).read()
    body = re.search(r"^def boxes_area\(boxes\):.*?(?=^def )", src, re.S | re.M).group(0)
    boxes_mod = types.ModuleType("utils.boxes")
    boxes_mod.np, boxes_mod.warnings = np, warnings
    exec(compile(body, os.path.join(lib, "utils", "boxes.py"), "exec"), boxes_mod.__dict__)
    utils_pkg, core_pkg = types.ModuleType("utils"), types.ModuleType("core")
    config_mod = types.ModuleType("core.config")
    config_mod.cfg = cfg
    utils_pkg.boxes = boxes_mod
    saved = {k: sys.modules.get(k) for k in ("utils", "utils.boxes", "core", "core.config")}
    sys.modules.update({"utils": utils_pkg, "utils.boxes": boxes_mod, "core": core_pkg, "core.config": config_mod})
    try:
        spec = importlib.util.spec_from_file_location("reference_utils_fpn", os.path.join(lib, "utils", "fpn.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def gen_fpn():
    fpn = _reference_fpn_module()
    rois, _ = syn.rois_fpn_distributed(400, batch=2, seed=31)
    edge = np.array([[0, 0, 0, 223, 223], [1, 0, 0, 222, 222], [0, 5, 5, 4, 4], [1, 10, 10, 10, 10],
                     [0, 0, 0, 447, 447], [1, 0, 0, 111, 111], [0, 0, 0, 1332, 799], [1, 3, 3, 1, 9]], np.float32)
    rois = np.vstack([rois, edge]).astype(np.float32)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lvls = fpn.map_rois_to_fpn_levels(rois[:, 1:5], 2, 5)
    blobs = {}
    fpn.add_multilevel_roi_blobs(blobs, "rois", rois, lvls, 2, 5)
    save("fpn.npz", rois=rois, levels=lvls, **blobs)


def gen_mask_targets():
    """Mask targets from polygon ground truth: the reference's own add_mask_rcnn_blobs (roi_data/mask_rcnn.py:34-107) and
    utils/segms.py executed from source on synthetic COCO-like polygons (oracle/ref_model.py); of pycocotools (absent) only
    frPyObjects / decode are bound, to the restatement of its maskApi.c in oracle/oracle.c -- the fixture pins everything
    around the rasteriser to the reference, the rasteriser itself to that restatement."""
    from oracle import ref_model

    out = {}
    for tag, m, seed, n_inst, per in (("a", 28, 41, 10, 6), ("b", 14, 43, 5, 4)):
        segms, boxes, classes = syn.polygon_instances(n_inst, seed=seed)
        rois = syn.jittered_boxes(boxes, per, seed=seed + 1, jitter=0.3)
        # RoIs the sampler can also hand over: thinner than a pixel, a point, one far from its polygon, one over the border
        extra = np.array([[boxes[0, 0], boxes[0, 1], boxes[0, 0] + 0.4, boxes[0, 3]], [boxes[1, 0], boxes[1, 1], boxes[1, 0], boxes[1, 1]],
                          [0, 0, 30, 30], [-40, -30, boxes[2, 2], boxes[2, 3]]], np.float32)
        rois = np.vstack([rois, extra]).astype(np.float32)
        labels = np.concatenate([np.repeat(classes, per), classes[:4]]).astype(np.int32)
        # background rows in between, as the sampled blob has them (the fg rows are picked out by label > 0)
        bg = syn.jittered_boxes(boxes[:3], 2, seed=seed + 2, jitter=0.9)
        sampled = np.vstack([rois[:7], bg, rois[7:]]).astype(np.float32)
        lab = np.concatenate([labels[:7], np.zeros(len(bg), np.int32), labels[7:]])
        blobs = ref_model.mask_rcnn_blobs_from_polygons(lab, sampled, segms, classes, im_scale=1.5, batch_idx=1, resolution=m)
        pts = np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for polys in segms for p in polys])
        counts = [len(p) // 2 for polys in segms for p in polys]
        out.update({tag + "_points": pts, tag + "_poly_start": np.concatenate([[0], np.cumsum(counts)]).astype(np.int32),
                    tag + "_inst_start": np.concatenate([[0], np.cumsum([len(polys) for polys in segms])]).astype(np.int32),
                    tag + "_classes": classes, tag + "_sampled_boxes": sampled, tag + "_labels": lab,
                    tag + "_resolution": np.int32(m), tag + "_boxes_from_polys": blobs["boxes_from_polys"],
                    tag + "_mask_rois": blobs["mask_rois"], tag + "_masks_int32": blobs["masks_int32"].astype(np.int8),
                    tag + "_roi_has_mask": blobs["roi_has_mask_int32"]})
    save("mask_targets.npz", **out)


def main():
    if not ref.available():
        sys.exit("oracle/_ref is not built: run `python oracle/build_ref.py` in the build container first")
    gen_roi_align()
    gen_roi_align_legacy()
    gen_roi_pool()
    gen_roi_crop()
    gen_nms()
    gen_soft_nms()
    gen_detection()
    gen_proposals()
    gen_fpn()
    gen_box_voting()
    gen_mask_targets()


if __name__ == "__main__":
    main()
