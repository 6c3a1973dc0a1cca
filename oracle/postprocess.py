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


def box_voting(top_dets, all_dets, thresh, scoring_method="ID", beta=1.0):
    """lib/utils/boxes.py:268-317 statement by statement (bbox_overlaps = utils.cython_bbox.bbox_overlaps, restated bit
    for bit by oracle.bbox_overlaps).  Pinned by tests/golden/box_voting.npz (the reference function's own source text,
    executed by tests/golden/generate.py against the reference's cython build)."""
    top_dets_out = top_dets.copy()                                   # :275
    top_boxes = top_dets[:, :4]
    all_boxes = all_dets[:, :4]
    all_scores = all_dets[:, 4]
    top_to_all_overlaps = oracle.bbox_overlaps(top_boxes, all_boxes)  # :279
    for k in range(top_dets_out.shape[0]):                           # :280
        inds_to_vote = np.where(top_to_all_overlaps[k] >= thresh)[0]
        boxes_to_vote = all_boxes[inds_to_vote, :]
        ws = all_scores[inds_to_vote]
        top_dets_out[k, :4] = np.average(boxes_to_vote, axis=0, weights=ws)   # :284
        if scoring_method == "ID":                                   # :285
            pass
        elif scoring_method == "TEMP_AVG":                           # :288-298
            P = np.vstack((ws, 1.0 - ws))
            P_max = np.max(P, axis=0)
            X = np.log(P / P_max)
            X_exp = np.exp(X / beta)
            P_temp = X_exp / np.sum(X_exp, axis=0)
            top_dets_out[k, 4] = P_temp[0].mean()
        elif scoring_method == "AVG":                                # :299-301
            top_dets_out[k, 4] = ws.mean()
        elif scoring_method == "IOU_AVG":                            # :302-306
            top_dets_out[k, 4] = np.average(ws, weights=top_to_all_overlaps[k, inds_to_vote])
        elif scoring_method == "GENERALIZED_AVG":                    # :307-309
            top_dets_out[k, 4] = np.mean(ws ** beta) ** (1.0 / beta)
        elif scoring_method == "QUASI_SUM":                          # :310-311
            top_dets_out[k, 4] = ws.sum() / float(len(ws)) ** beta
        else:
            raise NotImplementedError("Unknown scoring method {}".format(scoring_method))
    return top_dets_out


def box_results_with_nms_and_limit(scores, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_im=100,
                                   soft_nms=False, soft_nms_sigma=0.5, soft_nms_method="linear", bbox_vote=False,
                                   bbox_vote_thresh=0.8, bbox_vote_method="ID"):
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
        if bbox_vote:                                  # :766-773 (beta keeps box_voting's default: the call passes none)
            nms_dets = box_voting(nms_dets, dets_j, bbox_vote_thresh, scoring_method=bbox_vote_method) \
                if nms_dets.shape[0] else nms_dets
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
