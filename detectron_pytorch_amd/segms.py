"""Polygon ground truth on the device: the mask-target side of `lib/utils/segms.py` (polys_to_boxes :121-132,
polys_to_mask_wrt_box :93-119) for all foreground RoIs of a step in one launch (`mi_polys_to_masks_wrt_boxes`,
csrc/mask_targets.hip), where the reference loops over the RoIs on the host through pycocotools
(lib/roi_data/mask_rcnn.py:66-76).

COCO stores an instance as a LIST of polygons, a polygon as a flat list x0, y0, x1, y1, ... in image coordinates
(`roidb['segms'][i]`).  `PackedPolygons` is that ragged structure as three tensors, built once per roidb entry (host side,
like the reference's dataset code) and then resident on the device."""
import torch

from . import _lib


class PackedPolygons(object):
    """points [P, 2] float32, poly_start int32 [polygons + 1], inst_start int32 [instances + 1]; point_inst int64 [P] names
    the instance of every vertex (for polys_to_boxes)."""

    def __init__(self, points, poly_start, inst_start):
        self.points, self.poly_start, self.inst_start = points, poly_start, inst_start
        counts = (poly_start[1:] - poly_start[:-1]).long()
        poly_inst = torch.repeat_interleave(torch.arange(inst_start.numel() - 1, device=points.device),
                                            (inst_start[1:] - inst_start[:-1]).long())
        self.point_inst = torch.repeat_interleave(poly_inst, counts)

    @property
    def num_instances(self):
        return self.inst_start.numel() - 1

    @classmethod
    def from_lists(cls, segms, device="cpu"):
        """segms: per instance a list of polygons, each a flat sequence of coordinates (the COCO / roidb format).  The
        coordinates become float32, as segms.py:106 makes them."""
        pts, poly_start, inst_start = [], [0], [0]
        for polys in segms:
            for poly in polys:
                p = torch.as_tensor(poly, dtype=torch.float32).reshape(-1, 2)
                if p.size(0) < 3:
                    # json_dataset.py keeps polygons of >= 6 numbers only; pycocotools' frPyObjects would read a first
                    # element of 4 numbers as a BOX (frBbox), which this rasteriser does not
                    raise ValueError("PackedPolygons: a polygon needs at least 3 vertices (got %d)" % p.size(0))
                pts.append(p)
                poly_start.append(poly_start[-1] + p.size(0))
            inst_start.append(len(poly_start) - 1)
        points = torch.cat(pts) if pts else torch.zeros((0, 2), dtype=torch.float32)
        return cls(points.contiguous().to(device), torch.tensor(poly_start, dtype=torch.int32, device=device),
                   torch.tensor(inst_start, dtype=torch.int32, device=device))

    @classmethod
    def from_boxes(cls, boxes):
        """Every box [G, 4] as the four-vertex polygon (x1, y1), (x1, y2), (x2, y2), (x2, y1): rectangular ground truth
         in the polygon format."""
        b = boxes.float()
        points = torch.stack([b[:, 0], b[:, 1], b[:, 0], b[:, 3], b[:, 2], b[:, 3], b[:, 2], b[:, 1]], dim=1).reshape(-1, 2)
        g = b.size(0)
        return cls(points.contiguous(), torch.arange(0, 4 * g + 1, 4, dtype=torch.int32, device=b.device),
                   torch.arange(0, g + 1, dtype=torch.int32, device=b.device))

    def to(self, device):
        return PackedPolygons(self.points.to(device), self.poly_start.to(device), self.inst_start.to(device))


def flip_segms(segms, height, width):
    """segms.py:33-60 for polygon ground truth: the horizontally flipped roidb entry's `segms` (host lists in, host lists
    out, float64 as there; pack afterwards).  RLE (crowd) masks never become mask targets (mask_rcnn.py:40-41 takes the
    non-crowd instances) and are not handled."""
    import numpy as np

    flipped = []
    for segm in segms:
        if not isinstance(segm, list):
            raise NotImplementedError("flip_segms: RLE ground truth is outside the training targets' path")
        out = []
        for poly in segm:
            p = np.array(poly)
            p[0::2] = width - np.array(poly[0::2]) - 1
            out.append(p.tolist())
        flipped.append(out)
    return flipped


def polys_to_boxes(packed):
    """segms.py:121-132: the tight box of every instance's polygons, [G, 4] float32."""
    g = packed.num_instances
    x, y = packed.points[:, 0], packed.points[:, 1]
    out = torch.zeros((g, 4), dtype=torch.float32, device=x.device)
    big = torch.full((g,), float("inf"), dtype=torch.float32, device=x.device)
    out[:, 0] = big.scatter_reduce(0, packed.point_inst, x, "amin")
    out[:, 1] = big.scatter_reduce(0, packed.point_inst, y, "amin")
    out[:, 2] = (-big).scatter_reduce(0, packed.point_inst, x, "amax")
    out[:, 3] = (-big).scatter_reduce(0, packed.point_inst, y, "amax")
    return out


def polys_to_masks_wrt_boxes(packed, roi_inst, boxes, m):
    """segms.py:93-119 for every row: `boxes` [K, 4] float32 (image coordinates), `roi_inst` [K] the instance each row
    rasterises (negative: none, zeros).  Returns int32 [K, m * m], row-major [y][x] (mask_rcnn.py:75-76)."""
    _lib.require_cuda(boxes, "boxes")
    k = boxes.size(0)
    out = torch.empty((k, m * m), dtype=torch.int32, device=boxes.device)
    if k == 0:
        return out
    boxes = boxes.float().contiguous()
    inst = roi_inst.to(device=boxes.device, dtype=torch.int32).contiguous()
    packed = packed if packed.points.device == boxes.device else packed.to(boxes.device)
    with torch.cuda.device(boxes.device):
        rc = _lib.lib().mi_polys_to_masks_wrt_boxes(
            packed.points.data_ptr(), packed.poly_start.data_ptr(), packed.inst_start.data_ptr(), inst.data_ptr(),
            boxes.data_ptr(), out.data_ptr(), k, packed.num_instances, int(m), _lib.current_stream_handle(boxes.device))
    _lib.check(rc, "mi_polys_to_masks_wrt_boxes")
    return out
