"""mi_mask_paste_rle / mi_keypoint_decode against the CPU restatement of the reference's result formats
(oracle/results.py: segm_results and heatmaps_to_keypoints with OpenCV's and pycocotools' published algorithms restated --
parity unpinned, the packages are absent here).  Run lengths and arg-max positions are integers: exact; logits 1e-5,
probabilities 1e-4 relative (the softmax denominator is summed in another order)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def soft_masks(n, m, seed):
    """Blob-like soft masks: a smooth bump per mask plus noise, values in (0, 1)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:m, 0:m].astype(np.float32)
    out = np.zeros((n, m, m), np.float32)
    for i in range(n):
        cx, cy, s = rng.uniform(0.2 * m, 0.8 * m), rng.uniform(0.2 * m, 0.8 * m), rng.uniform(0.15 * m, 0.5 * m)
        out[i] = 1 / (1 + np.exp(-(1.5 - ((xx - cx) ** 2 + (yy - cy) ** 2) / s ** 2 + rng.randn(m, m) * 0.4)))
    return out


BOXES = np.array([[100, 120, 300, 420],        # inside
                  [-40, -30, 90, 60],          # over the top-left corner
                  [1250, 700, 1400, 900],      # over the bottom-right corner
                  [0, 0, 1332, 799],           # the whole image: full-height columns follow one another
                  [500, -10, 520, 820],        # full height, narrow
                  [640, 300, 640, 300],        # one pixel
                  [2000, 100, 2100, 200],      # entirely outside
                  [10, 790, 700, 799],         # touching the bottom edge
                  [300, 0, 310, 5]], np.int32)  # touching the top edge


@pytest.mark.parametrize("m", [28, 14])
def test_mask_paste_rle_matches_the_restatement(m):
    from detectron_pytorch_amd.rcnn import results
    from oracle import results as R

    im_h, im_w = 800, 1333
    masks = soft_masks(len(BOXES), m, seed=m)
    masks[5] = 0.9                                                     # the one-pixel box pastes a one
    counts, num, strings = results.mask_rle(torch.from_numpy(masks).to(dev()), torch.from_numpy(BOXES).to(dev()), im_h, im_w)
    for i, box in enumerate(BOXES):
        want = R.rle_counts(R.paste_mask(masks[i], box, im_h, im_w))
        assert num[i] == len(want), "detection %d: %d runs, want %d" % (i, num[i], len(want))
        assert counts[i, :num[i]].tolist() == want, "detection %d" % i
        assert results.rle_to_string(counts[i, :num[i]]) == R.rle_to_string(want) == strings[i]


def test_mask_paste_rle_grows_its_capacity_and_handles_noise():
    """A noise mask has thousands of runs: the first launch reports the size, the second delivers."""
    from detectron_pytorch_amd.rcnn import results
    from oracle import results as R

    rng = np.random.RandomState(5)
    masks = rng.rand(2, 28, 28).astype(np.float32)
    boxes = np.array([[50, 60, 400, 380], [5, 5, 60, 40]], np.int32)
    counts, num, strings = results.mask_rle(torch.from_numpy(masks).to(dev()), torch.from_numpy(boxes).to(dev()), 480, 640,
                                            capacity=64)
    for i in range(2):
        want = R.rle_counts(R.paste_mask(masks[i], boxes[i], 480, 640))
        assert len(want) > 64 or i == 1
        assert counts[i, :num[i]].tolist() == want and strings[i] == R.rle_to_string(want)
    only_counts = results.mask_rle_counts(torch.from_numpy(masks).to(dev()), torch.from_numpy(boxes).to(dev()), 480, 640)
    assert only_counts[0][0, :only_counts[1][0]].tolist() == R.rle_counts(R.paste_mask(masks[0], boxes[0], 480, 640))


def test_segm_results_end_to_end():
    from detectron_pytorch_amd.rcnn import config, results
    from oracle import results as R

    cfg = config.mask_rcnn_r50_fpn()
    cfg.MODEL.NUM_CLASSES = 5
    rng = np.random.RandomState(7)
    lengths = [0, 3, 0, 2, 1]
    r = sum(lengths)
    masks = np.stack([soft_masks(5, 28, seed=20 + i) for i in range(r)])          # [R, K, M, M]
    x1, y1 = rng.uniform(-30, 500, r), rng.uniform(-30, 300, r)
    ref = np.stack([x1, y1, x1 + rng.uniform(5, 300, r), y1 + rng.uniform(5, 250, r)], 1).astype(np.float32)
    cls_boxes = [np.zeros((n, 5), np.float32) for n in lengths]
    got = results.segm_results(cls_boxes, torch.from_numpy(masks).to(dev()), torch.from_numpy(ref).to(dev()), 427, 640, cfg)
    want = R.segm_results(cls_boxes, masks, ref, 427, 640)
    assert [len(g) for g in got] == lengths
    for g_cls, w_cls in zip(got, want):
        for g, w in zip(g_cls, w_cls):
            assert g == w


@pytest.mark.parametrize("min_size", [0, 40])
def test_keypoint_decode_matches_the_restatement(min_size):
    from detectron_pytorch_amd.rcnn import results
    from oracle import results as R

    rng = np.random.RandomState(11)
    maps = rng.randn(6, 17, 56, 56).astype(np.float32)
    maps[:, :, 20:24, 30:33] += 4.0                                    # a peak, as a trained head would give
    rois = np.array([[10, 20, 110, 220], [0, 0, 30.5, 15.2], [5, 5, 5.5, 5.2], [100.3, 50.7, 420.9, 610.1],
                     [7, 9, 63, 65], [300, 200, 301, 500]], np.float32)
    got = results.heatmaps_to_keypoints(torch.from_numpy(maps).to(dev()), torch.from_numpy(rois).to(dev()), min_size)
    want = R.heatmaps_to_keypoints(maps, rois, min_size)
    got = got.cpu().numpy()
    assert got.shape == want.shape == (6, 4, 17)
    # x, y are functions of the arg-max position only: exact
    assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 1], want[:, 1])
    np.testing.assert_allclose(got[:, 2], want[:, 2], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[:, 3], want[:, 3], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("n,seed", [(1, 0), (12, 1), (60, 2), (200, 3), (512, 4)])
def test_nms_oks_matches_the_restatement(n, seed):
    """mi_keypoint_nms_oks against oracle/results.py's nms_oks (itself pinned to the reference's text on the CPU): the
    kept indices in visiting order, for the reference's threshold and two others; the model-level switch KRCNN.NMS_OKS."""
    from detectron_pytorch_amd.rcnn import results
    from oracle import results as R
    from tests.test_results_cpu import _people

    kp, rois = _people(n, seed)
    for thresh in (0.3, 0.05, 0.9):
        want = [int(i) for i in R.nms_oks(kp, rois, thresh)]
        got = results.nms_oks(torch.from_numpy(kp).to(dev()), torch.from_numpy(rois).to(dev()), thresh)
        assert got.dtype == torch.int64 and got.cpu().tolist() == want
    with pytest.raises(RuntimeError):
        results.nms_oks(torch.zeros((3, 4, 16), device=dev()), torch.zeros((3, 4), device=dev()), 0.3)


def test_im_detect_all_results_mask_and_keypoints(hip_lib_path):
    """test.py:50-112 on the device for a mask + keypoint model: boxes, masks and keypoints in the reference's result
    formats; every RLE decodes to an image-sized mask confined to its (expanded) box."""
    from detectron_pytorch_amd.rcnn import config, inference, model
    from oracle import results as R
    from scenarios import H, W, scenario

    cfg = config.mask_keypoint_rcnn_x101_64x4d_fpn()
    cfg.MODEL.CONV_BODY = "FPN.fpn_ResNet50_conv5_body"               # the small body is enough for the result formats
    cfg.RESNETS.NUM_GROUPS, cfg.RESNETS.WIDTH_PER_GROUP, cfg.RESNETS.STRIDE_1X1 = 1, 64, True
    cfg.MODEL.NUM_CLASSES = 2                                          # background + person
    cfg.TEST.SCORE_THRESH = 0.3                                        # 2 classes at random init: scores ~0.5
    torch.manual_seed(cfg.RNG_SEED)
    net = model.GeneralizedRCNN(cfg).to(dev()).eval()
    _, _, data_np = scenario(seed=3)
    cls_boxes, cls_segms, cls_keyps = inference.im_detect_all_results(
        net, torch.from_numpy(data_np[:1]).to(dev()), torch.tensor([[float(H), float(W), 1.0]]))
    n = len(cls_boxes[1])
    assert n > 0 and len(cls_segms[1]) == n and len(cls_keyps[1]) == n
    boxes = cls_boxes[1][:, :4].cpu().numpy()
    exp = R.expand_boxes(boxes, 30.0 / 28).astype(np.int32)
    for i, rle in enumerate(cls_segms[1]):
        counts = R.rle_from_string(rle["counts"])
        assert rle["size"] == [H, W] and sum(counts) == H * W
        ys, xs = np.nonzero(R.rle_decode(counts, H, W))
        if xs.size:
            assert xs.min() >= max(exp[i, 0], 0) and xs.max() <= min(exp[i, 2], W - 1)
            assert ys.min() >= max(exp[i, 1], 0) and ys.max() <= min(exp[i, 3], H - 1)
    kp = torch.stack(cls_keyps[1]).cpu().numpy()
    assert kp.shape == (n, 4, cfg.KRCNN.NUM_KEYPOINTS)
    assert (kp[:, 0] >= boxes[:, None, 0]).all() and (kp[:, 0] <= boxes[:, None, 2] + 1).all()
    assert (kp[:, 3] > 0).all() and (kp[:, 3] <= 1).all()
    # KRCNN.NMS_OKS (test.py:857-862): the person boxes and keypoints are thinned together, the survivors are the rows
    # the restatement keeps of the unthinned result
    cfg.KRCNN.NMS_OKS = True
    cls_boxes2, _, cls_keyps2 = inference.im_detect_all_results(
        net, torch.from_numpy(data_np[:1]).to(dev()), torch.tensor([[float(H), float(W), 1.0]]))
    keep = [int(i) for i in R.nms_oks(kp, boxes, 0.3)]
    assert len(cls_boxes2[1]) == len(cls_keyps2[1]) == len(keep) <= n
    # (two forward passes of the convolutions are not bit-identical on this stack: compare with a tolerance)
    np.testing.assert_allclose(cls_boxes2[1].cpu().numpy(), cls_boxes[1].cpu().numpy()[keep], rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(torch.stack(cls_keyps2[1]).cpu().numpy()[:, 2:], kp[keep][:, 2:], rtol=1e-3, atol=1e-3)


def test_mask_detection_graph_equals_the_eager_result_formats(hip_lib_path):
    """Boxes + masks of a test image as ONE replayed hipGraph (static detection rows through the mask head and
    mi_mask_paste_rle) against im_detect_all_results launched eagerly: same detections; the masks may differ in the few
    pixels whose probability sits at the 0.5 threshold (the mask head runs at another batch size)."""
    from detectron_pytorch_amd.rcnn import config, inference, model
    from oracle import results as R
    from scenarios import H, W, scenario

    cfg = config.mask_rcnn_r50_fpn()
    cfg.MODEL.NUM_CLASSES = 3
    cfg.TEST.SCORE_THRESH = 0.2
    torch.manual_seed(cfg.RNG_SEED)
    net = model.GeneralizedRCNN(cfg).to(dev()).eval()
    graph = None
    for seed in (3, 4):
        _, _, data_np = scenario(seed=seed)
        blob = torch.from_numpy(data_np[:1]).to(dev())
        im_info = torch.tensor([[float(H), float(W), 1.0]])
        want_boxes, want_segms, _ = inference.im_detect_all_results(net, blob, im_info, (H, W))
        if graph is None:
            graph = inference.DetectionGraph(net, tuple(blob.shape), dev(), mask_im_shape=(H, W)).capture(blob, im_info)
        got_boxes, got_segms = graph(blob, im_info)
        assert [len(c) for c in got_boxes] == [len(c) for c in want_boxes] and sum(len(c) for c in want_boxes[1:]) > 0
        for j in range(1, cfg.MODEL.NUM_CLASSES):
            if len(want_boxes[j]):
                assert torch.allclose(got_boxes[j], want_boxes[j], rtol=0, atol=1e-3)
            for g, w in zip(got_segms[j], want_segms[j]):
                assert g["size"] == w["size"] == [H, W]
                a = R.rle_decode(R.rle_from_string(g["counts"]), H, W)
                b = R.rle_decode(R.rle_from_string(w["counts"]), H, W)
                assert (a != b).sum() <= 0.005 * max(int(b.sum()), 200)


def test_mask_paste_rle_against_an_independent_dense_path():
    """mi_mask_paste_rle against code that shares nothing with oracle/results.py: its strings are decoded by the test's own
    COCO-RLE decoder (written from the format description) and compared with a dense paste -- F.interpolate (half-pixel
    centres), threshold, scatter -- on the device.  Pixels whose interpolated value is within 1e-4 of the threshold may fall on
    either side (fp32 coordinate arithmetic differs); everywhere else the masks must be equal."""
    import torch.nn.functional as F

    from detectron_pytorch_amd.rcnn import results
    from test_results_cpu import _indep_decode_string

    im_h, im_w = 427, 640
    boxes = np.array([[100, 120, 300, 420], [-40, -30, 90, 60], [560, 380, 700, 500], [0, 0, 639, 426], [320, -10, 330, 500],
                      [200, 100, 200, 100], [900, 100, 950, 200], [10, 420, 300, 426], [33, 44, 61, 57]], np.int32)
    masks = soft_masks(len(boxes), 28, seed=3)
    _, num, strings = results.mask_rle(torch.from_numpy(masks).to(dev()), torch.from_numpy(boxes).to(dev()), im_h, im_w)
    for i, box in enumerate(boxes):
        got = _indep_decode_string(strings[i], im_h, im_w)
        padded = F.pad(torch.from_numpy(masks[i]).to(dev())[None, None], (1, 1, 1, 1))
        w, h = max(int(box[2] - box[0] + 1), 1), max(int(box[3] - box[1] + 1), 1)
        soft = F.interpolate(padded, size=(h, w), mode="bilinear", align_corners=False)[0, 0]
        ys, xs = torch.meshgrid(torch.arange(h, device=dev()) + int(box[1]), torch.arange(w, device=dev()) + int(box[0]),
                                indexing="ij")
        inside = (ys >= 0) & (ys < im_h) & (xs >= 0) & (xs < im_w)
        want = torch.zeros((im_h, im_w), dtype=torch.uint8, device=dev())
        want.index_put_((ys[inside], xs[inside]), (soft > 0.5)[inside].to(torch.uint8))
        near = torch.zeros((im_h, im_w), dtype=torch.bool, device=dev())
        close = ((soft - 0.5).abs() < 1e-4) & inside
        near.index_put_((ys[close], xs[close]), torch.ones(int(close.sum()), dtype=torch.bool, device=dev()))
        near = near.cpu().numpy()
        assert np.array_equal(got[~near], want.cpu().numpy()[~near]), "detection %d" % i
        assert near.sum() <= 0.002 * got.size
