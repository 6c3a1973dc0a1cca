"""TEST INFRASTRUCTURE -- numpy restatement of the reference's per-class detection post-processing.

`box_results_with_nms_and_limit` follows lib/core/test.py:732-790 statement by statement, with the global cfg values
it reads turned into arguments and the two helpers it calls taken from this oracle (oracle.nms_cython ==
utils.cython_nms.nms behind utils.boxes.nms, boxes.py:320-324; oracle.soft_nms == utils.cython_nms.soft_nms behind
utils.boxes.soft_nms, boxes.py:327-344).  Pinned by tests/golden/detection.npz, which tests/golden/generate.py
produces by executing the reference function's own source.  Never imported by the product package.
"""
import numpy as np

import oracle

SOFT_NMS_METHODS = {"hard": 0, "linear": 1, "gaussian": 2}  # boxes.py:334


def box_results_with_nms_and_limit(scores, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100,
                                   soft_nms=False, soft_nms_sigma=0.5, soft_nms_method="linear"):
    num_classes = scores.shape[1]                      # cfg.MODEL.NUM_CLASSES (:745)
    cls_boxes = [[] for _ in range(num_classes)]
    for j in range(1, num_classes):                    # :749
        inds = np.where(scores[:, j] > score_thresh)[0]
        scores_j = scores[inds, j]
        boxes_j = boxes[inds, j * 4:(j + 1) * 4]
        dets_j = np.hstack((boxes_j, scores_j[:, np.newaxis])).astype(np.float32, copy=False)
        if soft_nms:                                   # :754-761
            if dets_j.shape[0] == 0:                   # boxes.py:331-332
                nms_dets = dets_j
            else:
                nms_dets, _ = oracle.soft_nms(dets_j, soft_nms_sigma, nms_thresh, 0.0001, SOFT_NMS_METHODS[soft_nms_method])
        else:                                          # :762-764
            keep = oracle.nms_cython(dets_j, nms_thresh) if dets_j.shape[0] else []   # boxes.py:322-323
            nms_dets = dets_j[keep, :]
        cls_boxes[j] = nms_dets
    if detections_per_im > 0:                          # :776-785
        image_scores = np.hstack([cls_boxes[j][:, -1] for j in range(1, num_classes)])
        if len(image_scores) > detections_per_im:
            image_thresh = np.sort(image_scores)[-detections_per_im]
            for j in range(1, num_classes):
                keep = np.where(cls_boxes[j][:, -1] >= image_thresh)[0]
                cls_boxes[j] = cls_boxes[j][keep, :]
    im_results = np.vstack([cls_boxes[j] for j in range(1, num_classes)])
    return im_results[:, -1], im_results[:, :-1], cls_boxes
