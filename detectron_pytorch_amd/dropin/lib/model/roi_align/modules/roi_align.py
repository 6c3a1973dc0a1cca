"""Overlay for lib/model/roi_align/modules/roi_align.py."""
from detectron_pytorch_amd.roi_align import LegacyRoIAlign as RoIAlign  # noqa: F401
from detectron_pytorch_amd.roi_align import LegacyRoIAlignAvg as RoIAlignAvg  # noqa: F401
from detectron_pytorch_amd.roi_align import LegacyRoIAlignMax as RoIAlignMax  # noqa: F401
