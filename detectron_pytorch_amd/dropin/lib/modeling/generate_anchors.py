"""Overlay for lib/modeling/generate_anchors.py (FPN.py:12, rpn_heads.py:6 import it): same table, numpy >= 1.24 safe
(the reference spells the dtype `np.float`)."""
from detectron_pytorch_amd.generate_proposals import generate_anchors  # noqa: F401
