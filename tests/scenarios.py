"""Seeded inputs shared by tests/test_model_cpu.py, tests/test_e2e_gpu.py and tests/golden/generate_model.py (numpy
RandomState streams are identical on every machine)."""
import numpy as np

H, W, NUM_GT = 256, 320, 4


def scenario(seed=5):
    """Two small images with gt boxes that spread over FPN levels 2-4: (boxes per image, classes per image, image blob)."""
    rng = np.random.RandomState(seed)
    boxes, classes = [], []
    for _ in range(2):
        bw, bh = rng.uniform(24, 250, NUM_GT), rng.uniform(24, 200, NUM_GT)
        x1, y1 = rng.uniform(0, W - 1 - bw), rng.uniform(0, H - 1 - bh)
        boxes.append(np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32))
        classes.append(rng.randint(1, 81, NUM_GT).astype(np.int32))
    data = (rng.randn(2, 3, H, W) * 50).astype(np.float32)
    return boxes, classes, data


def polygons_for_boxes(boxes, seed=23):
    """Per gt box a COCO-style instance (list of 1-2 polygons, flat x0, y0, ... with two decimals) drawn inside the box:
    a star-shaped outline of 6-40 vertices, sometimes a second small part.  The polygons' tight box is smaller than the gt
    box, as in real annotations (mask_rcnn.py:44 matches the RoIs against the boxes of the POLYGONS)."""
    rng = np.random.RandomState(seed)
    out = []
    for b in boxes:
        cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, (b[2] - b[0]) / 2, (b[3] - b[1]) / 2
        polys = []
        for part in range(1 + (rng.rand() < 0.4)):
            k = rng.randint(6, 41)
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            rad = rng.uniform(0.4, 1.0, k) * (1.0 if part == 0 else 0.3)
            ox, oy = (0.0, 0.0) if part == 0 else rng.uniform(-0.6, 0.6, 2) * (rx, ry)
            pts = np.round(np.stack([cx + ox + rx * rad * np.cos(ang), cy + oy + ry * rad * np.sin(ang)], 1), 2)
            polys.append([float(v) for v in pts.reshape(-1)])
        out.append(polys)
    return out


def synthetic_conv_outputs(seed, n, h=H, w=W):
    """Seeded stand-ins for the outputs of the convolutions: pyramid blobs [P6, P5, P4, P3, P2] (256 channels), and per
    RPN level 2..6 the objectness logits [n,3,h,w] and box deltas [n,12,h,w]."""
    rng = np.random.RandomState(seed)
    sizes = [(h // s, w // s) for s in (64, 32, 16, 8, 4)]
    blobs = [(rng.randn(n, 256, a, b) * 0.5).astype(np.float32) for a, b in sizes]
    logits = [(rng.randn(n, 3, a, b) * 2.0).astype(np.float32) for a, b in reversed(sizes)]        # level 2 .. 6
    deltas = [(rng.randn(n, 12, a, b) * 0.3).astype(np.float32) for a, b in reversed(sizes)]
    return blobs, logits, deltas


def keypoint_scenario(seed=17):
    """Person boxes of `scenario()` with 17 key points each ([NUM_GT, 3, 17] int32 per image: x, y, visibility 0..2), some
    outside their box, the first three of the first box on its lower-right corner (the inclusive-edge rule of
    utils/keypoints.py:160-211)."""
    boxes, _, data = scenario()
    rng = np.random.RandomState(seed)
    kps = []
    for bx in boxes:
        kx = bx[:, 0:1] + rng.uniform(-0.1, 1.1, (NUM_GT, 17)) * (bx[:, 2:3] - bx[:, 0:1])
        ky = bx[:, 1:2] + rng.uniform(-0.1, 1.1, (NUM_GT, 17)) * (bx[:, 3:4] - bx[:, 1:2])
        vis = rng.randint(0, 3, (NUM_GT, 17))
        k = np.stack([kx, ky, vis], axis=1).astype(np.int32)
        k[0, 0, 0], k[0, 1, 0], k[0, 2, 0] = int(bx[0, 2]), int(bx[0, 3]), 2
        kps.append(k)
    classes = [np.ones(NUM_GT, np.int32) for _ in boxes]
    return boxes, classes, kps, data
