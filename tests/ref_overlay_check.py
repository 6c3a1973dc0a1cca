"""Run by tests/test_model_cpu.py in a process of its own: the REFERENCE'S OWN `Generalized_RCNN.roi_feature_transform`
(/root/reference/lib/modeling/model_builder.py:252-324, imported from source, not restated) executed with the drop-in
overlay `detectron_pytorch_amd/dropin/lib` in front of the reference's `lib/` on sys.path -- the integration INTEGRATION.md
section 2 describes.  `import modeling.model_builder` then binds the reference's call sites (:279, :286, :290-291, :312,
:317, :321-322) to this package's RoIAlignFunction / RoIPoolFunction / RoICropFunction.

There is no GPU here, so the autograd Function behind RoIAlignFunction (detectron_pytorch_amd.roi_align._RoIAlign, which
launches the HIP kernels) is bound to the CPU oracle for the duration of the check; everything else -- the classes the
overlay exports, their constructor / call signatures, the per-level dispatch, the concatenation and the restore permutation
of the reference -- runs as shipped.  Checked, forward and backward: the FPN branch and the single-level branch give what
(a) the oracle gives level by level and (b) this package's own roi_xform.roi_feature_transform gives on the same inputs."""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

import oracle  # noqa: E402
from oracle import ref_model  # noqa: E402
import cpu_backend  # noqa: E402


def main():
    ref_cfg = ref_model.configure("configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml",
                                  MODEL__LOAD_IMAGENET_PRETRAINED_WEIGHTS=False)
    lib = os.path.join(ref_model.REFERENCE, "lib")
    overlay = os.path.join(ROOT, "detectron_pytorch_amd", "dropin", "lib")
    # ref_model.load() bound the compiled operators to host builds of the reference's kernels; undo exactly that part and
    # let the overlay provide them, as a user of the drop-in would: overlay first, the reference's tree second
    for name in [m for m in sys.modules if m.split(".")[0] in ("modeling", "model")]:
        del sys.modules[name]
    for name in ("utils.cython_nms", "utils.cython_bbox"):
        sys.modules.pop(name, None)
    sys.path[:] = [overlay, lib] + [p for p in sys.path if p not in (overlay, lib)]
    # `utils` (the reference's package) is already imported: give it the merged search path the overlay's own
    # utils/__init__.py would have set up had it been imported first
    sys.modules["utils"].__path__ = [os.path.join(overlay, "utils"), os.path.join(lib, "utils")]

    import detectron_pytorch_amd.roi_align as mi_roi_align

    class _OnOracle(object):      # the HIP launch replaced by the CPU oracle; signature of _RoIAlign.apply
        @staticmethod
        def apply(features, rois, ah, aw, scale, sr, variant):
            assert variant == mi_roi_align._lib.ROI_ALIGN_CAFFE2
            return cpu_backend._RoIAlignOracle.apply(features, rois, ah, aw, scale, sr)

    mi_roi_align._RoIAlign = _OnOracle
    import modeling.model_builder as mb

    assert mb.__file__ == os.path.join(lib, "modeling", "model_builder.py"), mb.__file__
    assert mb.RoIAlignFunction is mi_roi_align.RoIAlignFunction, "the reference's call sites are not bound to the drop-in"
    import detectron_pytorch_amd.roi_pool as mi_roi_pool
    import detectron_pytorch_amd.roi_crop as mi_roi_crop

    assert mb.RoIPoolFunction is mi_roi_pool.RoIPoolFunction and mb.RoICropFunction is mi_roi_crop.RoICropFunction
    import utils.fpn as ref_fpn        # the reference's own level mapping and blob splitting

    # ---- FPN branch (model_builder.py:266-306) ----
    rng = np.random.RandomState(0)
    n, c = 2, 8
    sizes = {5: (13, 17), 4: (25, 34), 3: (50, 67), 2: (100, 134)}
    blobs_in = [torch.from_numpy(rng.randn(n, c, *sizes[l]).astype(np.float32)).requires_grad_() for l in (5, 4, 3, 2)]
    scales = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4]
    r = 90
    side = np.exp(rng.uniform(np.log(12), np.log(500), r))
    cx, cy = rng.uniform(0, 530, r), rng.uniform(0, 400, r)
    boxes = np.stack([cx - side / 2, cy - side / 2, cx + side / 2, cy + side / 2], 1)
    boxes = np.clip(boxes, 0, [535, 399, 535, 399])
    boxes[:6] = [0, 0, 535, 399]                              # whole-image boxes: the coarsest level
    boxes[:6] += rng.uniform(-3, 3, (6, 4))
    rois = np.hstack([rng.randint(0, n, (r, 1)), boxes]).astype(np.float32)
    lvls = ref_fpn.map_rois_to_fpn_levels(rois[:, 1:5], ref_cfg.FPN.ROI_MIN_LEVEL, ref_cfg.FPN.ROI_MAX_LEVEL)
    assert len(set(lvls.tolist())) == 4, "the scenario should put RoIs on every level"
    rpn_ret = {"rois": rois}
    ref_fpn.add_multilevel_roi_blobs(rpn_ret, "rois", rois, lvls, ref_cfg.FPN.ROI_MIN_LEVEL, ref_cfg.FPN.ROI_MAX_LEVEL)
    rpn_ret["rois_fpn3"] = rpn_ret["rois_fpn3"][:0]          # an empty level is skipped (:275)
    keep = lvls != 3
    rois_k = rois[keep]
    rpn_ret2 = {"rois": rois_k}
    ref_fpn.add_multilevel_roi_blobs(rpn_ret2, "rois", rois_k, lvls[keep], 2, 5)
    self_stub = type("Stub", (), {"grid_size": 14})()
    out = mb.Generalized_RCNN.roi_feature_transform(self_stub, blobs_in, rpn_ret2, blob_rois="rois", method="RoIAlign",
                                                    resolution=7, spatial_scale=scales, sampling_ratio=2)
    assert tuple(out.shape) == (rois_k.shape[0], c, 7, 7)
    want = np.zeros(out.shape, np.float32)
    for k, lvl in enumerate((5, 4, 3, 2)):
        idx = np.nonzero(lvls[keep] == lvl)[0]
        if idx.size:
            want[idx] = oracle.roi_align_forward(blobs_in[k].detach().numpy(), rois_k[idx], 7, 7, scales[k], 2)
    assert np.array_equal(out.detach().numpy(), want), "FPN branch through the overlay differs from the oracle"
    g = torch.from_numpy(rng.randn(*out.shape).astype(np.float32))
    out.backward(g)
    for k, lvl in enumerate((5, 4, 3, 2)):
        idx = np.nonzero(lvls[keep] == lvl)[0]
        ref_g = oracle.roi_align_backward(g.numpy()[idx], rois_k[idx], tuple(blobs_in[k].shape), scales[k], 2) if idx.size \
            else np.zeros(tuple(blobs_in[k].shape), np.float32)
        got_g = blobs_in[k].grad.numpy() if blobs_in[k].grad is not None else np.zeros_like(ref_g)
        np.testing.assert_allclose(got_g, ref_g, rtol=1e-5, atol=1e-6, err_msg="gradient of level %d" % lvl)
    # the package's own restatement of the function (what rcnn/ uses) on the same inputs, per-level path
    from detectron_pytorch_amd import roi_xform

    mine = roi_xform.roi_feature_transform([b.detach() for b in blobs_in], rpn_ret2, blob_rois="rois", method="RoIAlign",
                                           resolution=7, spatial_scale=scales, sampling_ratio=2, fused=False)
    assert np.array_equal(mine.numpy(), want), "roi_xform.roi_feature_transform differs from the reference's function"

    # ---- single-level branch (model_builder.py:307-322) ----
    feat = torch.from_numpy(rng.randn(n, c, 25, 34).astype(np.float32))
    out1 = mb.Generalized_RCNN.roi_feature_transform(self_stub, feat, {"rois": rois}, blob_rois="rois", method="RoIAlign",
                                                     resolution=14, spatial_scale=1.0 / 16, sampling_ratio=0)
    want1 = oracle.roi_align_forward(feat.numpy(), rois, 14, 14, 1.0 / 16, 0)
    assert np.array_equal(out1.numpy(), want1), "single-level branch through the overlay differs from the oracle"
    mine1 = roi_xform.roi_feature_transform(feat, {"rois": rois}, blob_rois="rois", method="RoIAlign", resolution=14,
                                            spatial_scale=1.0 / 16, sampling_ratio=0)
    assert np.array_equal(mine1.numpy(), want1)
    print("OVERLAY_ROI_FEATURE_TRANSFORM_OK rois=%d levels=%s" % (rois_k.shape[0], sorted(set(lvls[keep].tolist()))))


if __name__ == "__main__":
    main()
