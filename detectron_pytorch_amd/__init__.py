"""MI355X-native (gfx950) RoI-transform / NMS hot path of Detectron.pytorch.

The compute lives in libmi_detectron_ops.so (hand-written HIP, C-ABI in include/mi_detectron_ops.h);
this package is the host-side mirror of the reference's Python operator interface for that path:

    roi_align.RoIAlignFunction(ah, aw, scale, sampling_ratio)(features, rois)   # Caffe2 semantics
    roi_align.LegacyRoIAlignFunction(ah, aw, scale)(features, rois)             # jwyang semantics
    roi_pool.RoIPoolFunction(ph, pw, scale)(features, rois)
    roi_crop.RoICropFunction()(input, grid_yx)
    nms.nms_gpu(dets, thresh) / nms.cython_nms(dets, thresh) / nms.bbox_overlaps(boxes, query)

`dropin/` re-exports them under the reference's own module paths (modeling.roi_xfrom..., model.nms...,
utils.cython_nms ...) so the reference's modeling/model_builder.py imports resolve unchanged.

There is no CPU fallback: importing works anywhere, but calling an op without the built HIP
library or without a GPU raises.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
