"""CPU restatement of the polygon side of the mask TARGETS (labelling): lib/utils/segms.py
polys_to_boxes / polys_to_mask / polys_to_mask_wrt_box and the loop of lib/roi_data/mask_rcnn.py:34-76 around them.
Test infrastructure: only tests/ may import this.

PARITY PARTLY PINNED.  The reference rasterises through pycocotools 2.0 (`mask_util.frPyObjects` + `mask_util.decode`,
segms.py:66-67,114-115), which is not installed here, is not part of /root/reference and cannot be fetched: its published
algorithm (common/maskApi.c rleFrPoly, rleDecode) is restated in oracle.c `oracle_poly_to_mask` and checked against
hand-derived vectors and properties only.  Pinned to the reference: everything around it -- the functions below restate
segms.py's arithmetic (checked against the fixture its own text produced), and tests/test_model_cpu.py executes the reference's OWN add_mask_rcnn_blobs with polygon `segms` and only
`polys_to_mask_wrt_box`'s two pycocotools calls bound to this restatement.
"""
import ctypes

import numpy as np

from . import lib

_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def fr_poly_decode(polygons, height, width):
    """mask_util.decode(mask_util.frPyObjects(polygons, height, width)): [height, width, len(polygons)] uint8.  A polygon
    is a flat sequence x0, y0, x1, y1, ... (pycocotools converts it to float64: _mask.pyx frPoly)."""
    out = np.zeros((height, width, len(polygons)), dtype=np.uint8)
    for i, poly in enumerate(polygons):
        xy = np.ascontiguousarray(np.asarray(poly, dtype=np.float64).reshape(-1))
        cm = np.zeros(height * width, dtype=np.uint8)                 # column-major, as rleDecode writes it
        lib().oracle_poly_to_mask(xy.ctypes.data_as(_f64p), ctypes.c_int(xy.size // 2), ctypes.c_int(height),
                                  ctypes.c_int(width), cm.ctypes.data_as(_u8p))
        out[:, :, i] = cm.reshape(width, height).T
    return out


def _union(stack):
    """segms.py:68-70 / :116-118: the per-polygon images summed over the polygon axis and thresholded at > 0."""
    return (stack.astype(np.float32).sum(axis=2) > 0).astype(np.float32)


def polys_to_mask(polygons, height, width):
    """segms.py:60-71: the polygons of one instance in a height x width image."""
    return _union(fr_poly_decode(polygons, height, width))


def polys_to_mask_wrt_box(polygons, box, M):
    """segms.py:93-119: the polygons of one instance moved into `box`'s frame and scaled to M x M, rasterised, OR-ed.
    The shift and scale run in float32 exactly as numpy evaluates :108-111 -- (x - box_x1) * M / w with a float32 box and
    w, h = max(box side, 1) -- before pycocotools widens to float64."""
    box = np.asarray(box, dtype=np.float32)
    side = np.maximum(box[2:4] - box[0:2], 1)                       # :99-103
    moved = []
    for poly in polygons:
        p = np.array(poly, dtype=np.float32)                         # :106
        for axis in (0, 1):
            p[axis::2] = (p[axis::2] - box[axis]) * M / side[axis]
        moved.append(p)
    return _union(fr_poly_decode(moved, M, M))


def polys_to_boxes(polys):
    """segms.py:121-132: per instance the tight box (x_min, y_min, x_max, y_max) over all of its polygons, float32."""
    out = np.zeros((len(polys), 4), dtype=np.float32)
    for i, instance in enumerate(polys):
        pts = np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in instance])
        out[i] = np.concatenate([pts.min(axis=0), pts.max(axis=0)])
    return out
