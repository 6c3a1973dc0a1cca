"""Overlay for the compiled lib/utils/cython_nms module (utils/boxes.py:52 imports it)."""
from detectron_pytorch_amd.nms import cython_nms as nms  # noqa: F401
from detectron_pytorch_amd.nms import soft_nms  # noqa: F401
