"""Overlay for lib/model/roi_align/functions/roi_align.py (legacy jwyang RoIAlign, 3-argument ctor)."""
from detectron_pytorch_amd.roi_align import LegacyRoIAlignFunction as RoIAlignFunction  # noqa: F401
