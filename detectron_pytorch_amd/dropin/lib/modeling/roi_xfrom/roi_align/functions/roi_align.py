"""Overlay for lib/modeling/roi_xfrom/roi_align/functions/roi_align.py (Caffe2-semantics RoIAlign)."""
from detectron_pytorch_amd.roi_align import RoIAlignFunction  # noqa: F401
