"""Overlay for the compiled lib/utils/cython_bbox module (utils/boxes.py:51 imports it)."""
from detectron_pytorch_amd.nms import bbox_overlaps  # noqa: F401
