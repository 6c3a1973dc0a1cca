"""CPU restatement of the polygon side of the mask TARGETS (labelling): lib/utils/segms.py
polys_to_boxes / polys_to_mask / polys_to_mask_wrt_box and the loop of lib/roi_data/mask_rcnn.py:34-76 around them.
Test infrastructure: only tests/ may import this.

PARITY PARTLY PINNED.  The reference rasterises through pycocotools 2.0 (`mask_util.frPyObjects` + `mask_util.decode`,
segms.py:66-67,114-115), which is not installed here, is not part of /root/reference and cannot be fetched: its published
algorithm (common/maskApi.c rleFrPoly, rleDecode) is restated in oracle.c `oracle_poly_to_mask` and checked against
hand-derived vectors only.  Pinned to the reference: everything around it -- the functions below follow segms.py line by
line, and tests/test_model_cpu.py executes the reference's OWN add_mask_rcnn_blobs with polygon `segms` and only
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


def polys_to_mask(polygons, height, width):
    """segms.py:60-71."""
    mask = np.array(fr_poly_decode(polygons, height, width), dtype=np.float32)
    mask = np.sum(mask, axis=2)
    return np.array(mask > 0, dtype=np.float32)


def polys_to_mask_wrt_box(polygons, box, M):
    """segms.py:93-119: the polygons of one instance, moved into `box`'s frame and scaled to M x M (float32 arithmetic, as
    numpy does it there), rasterised, OR-ed."""
    w = box[2] - box[0]
    h = box[3] - box[1]
    w = np.maximum(w, 1)
    h = np.maximum(h, 1)
    polygons_norm = []
    for poly in polygons:
        p = np.array(poly, dtype=np.float32)
        p[0::2] = (p[0::2] - box[0]) * M / w
        p[1::2] = (p[1::2] - box[1]) * M / h
        polygons_norm.append(p)
    mask = np.array(fr_poly_decode(polygons_norm, M, M), dtype=np.float32)
    mask = np.sum(mask, axis=2)
    return np.array(mask > 0, dtype=np.float32)


def polys_to_boxes(polys):
    """segms.py:121-132: tight box of every instance's polygons."""
    boxes = np.zeros((len(polys), 4), dtype=np.float32)
    for i, poly in enumerate(polys):
        x0 = min(min(p[::2]) for p in poly)
        x1 = max(max(p[::2]) for p in poly)
        y0 = min(min(p[1::2]) for p in poly)
        y1 = max(max(p[1::2]) for p in poly)
        boxes[i, :] = [x0, y0, x1, y1]
    return boxes
