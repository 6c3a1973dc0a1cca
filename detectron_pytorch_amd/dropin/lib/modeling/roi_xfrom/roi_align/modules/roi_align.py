"""Overlay for lib/modeling/roi_xfrom/roi_align/modules/roi_align.py."""
from detectron_pytorch_amd.roi_align import RoIAlign, RoIAlignAvg, RoIAlignMax  # noqa: F401
