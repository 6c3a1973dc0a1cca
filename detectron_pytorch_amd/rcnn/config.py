"""The subset of the reference's global `cfg` (lib/core/config.py:22-993) the R-CNN graph reads, under the reference's
key names, so that the shipped `configs/**/*.yaml` stay a drop-in (`merge_from_file` accepts them unchanged; keys this
graph never reads are kept as given, not validated).

Unlike the reference there is no process-wide singleton: a `Config` is an ordinary value passed to the constructors.
Defaults are the reference's (file:line cited per group); `mask_rcnn_r50_fpn()` is
configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml applied to them, `faster_rcnn_r50_fpn()` the Faster R-CNN baseline.
"""
import copy
import math


class Node(dict):
    """Attribute-style dict (the role of lib/utils/collections.py:AttrDict)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return Node({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _defaults():
    c = Node()
    c.NUM_GPUS = 1
    c.RNG_SEED = 3                                            # config.py:945
    c.BBOX_XFORM_CLIP = math.log(1000.0 / 16.0)               # config.py:936
    c.TRAIN = Node(                                           # config.py:33-160
        SCALES=(600,), MAX_SIZE=1000, IMS_PER_BATCH=2, BATCH_SIZE_PER_IM=64, FG_FRACTION=0.25, FG_THRESH=0.5,
        BG_THRESH_HI=0.5, BG_THRESH_LO=0.0, RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3, RPN_FG_FRACTION=0.5,
        RPN_BATCH_SIZE_PER_IM=256, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000,
        RPN_STRADDLE_THRESH=0, RPN_MIN_SIZE=0, FREEZE_CONV_BODY=False)
    c.TEST = Node(                                            # config.py:166-370
        SCALE=600, MAX_SIZE=1000, NMS=0.3, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000,
        RPN_MIN_SIZE=0, DETECTIONS_PER_IM=100, SCORE_THRESH=0.05,
        SOFT_NMS=Node(ENABLED=False, METHOD="linear", SIGMA=0.5),
        BBOX_VOTE=Node(ENABLED=False, VOTE_TH=0.8, SCORING_METHOD="ID", SCORING_METHOD_BETA=1.0))   # config.py:371-386
    c.MODEL = Node(                                           # config.py:390-450
        TYPE="", CONV_BODY="", NUM_CLASSES=81, CLS_AGNOSTIC_BBOX_REG=False, BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0),
        FASTER_RCNN=False, MASK_ON=False, KEYPOINTS_ON=False, RPN_ONLY=False, SHARE_RES5=False,
        LOAD_IMAGENET_PRETRAINED_WEIGHTS=False)
    c.SOLVER = Node(                                          # config.py:520-610
        BASE_LR=0.001, MOMENTUM=0.9, WEIGHT_DECAY=0.0005, WEIGHT_DECAY_GN=0.0, BIAS_DOUBLE_LR=True,
        BIAS_WEIGHT_DECAY=False)
    c.FAST_RCNN = Node(                                       # config.py:615-650
        ROI_BOX_HEAD="", MLP_HEAD_DIM=1024, ROI_XFORM_METHOD="RoIPoolF", ROI_XFORM_SAMPLING_RATIO=0,
        ROI_XFORM_RESOLUTION=14)
    c.RPN = Node(RPN_ON=False, CLS_ACTIVATION="sigmoid")      # config.py:655-675
    c.FPN = Node(                                             # config.py:680-726
        FPN_ON=False, DIM=256, ZERO_INIT_LATERAL=False, COARSEST_STRIDE=32, MULTILEVEL_ROIS=False,
        ROI_CANONICAL_SCALE=224, ROI_CANONICAL_LEVEL=4, ROI_MAX_LEVEL=5, ROI_MIN_LEVEL=2, MULTILEVEL_RPN=False,
        RPN_MAX_LEVEL=6, RPN_MIN_LEVEL=2, RPN_ASPECT_RATIOS=(0.5, 1, 2), RPN_ANCHOR_START_SIZE=32, RPN_COLLECT_SCALE=1,
        EXTRA_CONV_LEVELS=False, USE_GN=False)
    c.MRCNN = Node(                                           # config.py:731-775
        ROI_MASK_HEAD="", RESOLUTION=14, ROI_XFORM_METHOD="RoIAlign", ROI_XFORM_RESOLUTION=7,
        ROI_XFORM_SAMPLING_RATIO=0, DIM_REDUCED=256, DILATION=2, UPSAMPLE_RATIO=1, USE_FC_OUTPUT=False,
        CONV_INIT="GaussianFill", CLS_SPECIFIC_MASK=True, WEIGHT_LOSS_MASK=1.0, THRESH_BINARIZE=0.5)
    c.KRCNN = Node(                                           # config.py:780-860
        ROI_KEYPOINTS_HEAD="", HEATMAP_SIZE=-1, UP_SCALE=-1, USE_DECONV=False, DECONV_DIM=256, USE_DECONV_OUTPUT=False,
        DILATION=1, DECONV_KERNEL=4, NUM_KEYPOINTS=-1, NUM_STACKED_CONVS=8, CONV_HEAD_DIM=256, CONV_HEAD_KERNEL=3,
        CONV_INIT="GaussianFill", ROI_XFORM_METHOD="RoIAlign", ROI_XFORM_RESOLUTION=7, ROI_XFORM_SAMPLING_RATIO=0,
        LOSS_WEIGHT=1.0, NORMALIZE_BY_VISIBLE_KEYPOINTS=True, NMS_OKS=False, INFERENCE_MIN_SIZE=0)
    c.RESNETS = Node(                                         # config.py:870-900
        NUM_GROUPS=1, WIDTH_PER_GROUP=64, STRIDE_1X1=True, TRANS_FUNC="bottleneck_transformation",
        STEM_FUNC="basic_bn_stem", SHORTCUT_FUNC="basic_bn_shortcut", RES5_DILATION=1, FREEZE_AT=2,
        IMAGENET_PRETRAINED_WEIGHTS="", USE_GN=False)
    return c


class Config(Node):
    def merge(self, other):
        """Recursive merge of a nested mapping (the role of config.py:1083-1118 without the type coercion: yaml scalars
        arrive typed, tuples may arrive as lists)."""
        _merge(other, self)
        return self

    def merge_from_file(self, path):
        """config.py:1034-1045: apply a reference yaml (configs/**/*.yaml) on top of the current values."""
        import yaml

        with open(path, "r") as f:
            self.merge(yaml.safe_load(f) or {})
        return infer(self)


def _merge(src, dst):
    for k, v in src.items():
        if isinstance(v, dict):
            if not isinstance(dst.get(k), dict):
                dst[k] = Node()
            _merge(v, dst[k])
        else:
            if isinstance(v, str) and v.startswith("(") and v.endswith(")"):   # yaml writes tuples as "(800,)"
                import ast

                v = ast.literal_eval(v)
            dst[k] = tuple(v) if isinstance(v, list) else v


def infer(cfg):
    """config.py:1006-1031 `assert_and_infer_cfg`: the derived switches the graph relies on."""
    if cfg.MODEL.RPN_ONLY or cfg.MODEL.FASTER_RCNN:
        cfg.RPN.RPN_ON = True
    return cfg


def default_config():
    c = Config()
    c.merge(_defaults())
    return c


def faster_rcnn_r50_fpn():
    """configs/baselines/e2e_faster_rcnn_R-50-FPN_1x.yaml on the defaults (BASELINE.json config 3)."""
    c = default_config()
    c.merge(dict(
        MODEL=dict(TYPE="generalized_rcnn", CONV_BODY="FPN.fpn_ResNet50_conv5_body", FASTER_RCNN=True),
        NUM_GPUS=8,
        SOLVER=dict(WEIGHT_DECAY=0.0001, BASE_LR=0.02),
        FPN=dict(FPN_ON=True, MULTILEVEL_ROIS=True, MULTILEVEL_RPN=True),
        FAST_RCNN=dict(ROI_BOX_HEAD="fast_rcnn_heads.roi_2mlp_head", ROI_XFORM_METHOD="RoIAlign",
                       ROI_XFORM_RESOLUTION=7, ROI_XFORM_SAMPLING_RATIO=2),
        TRAIN=dict(SCALES=(800,), MAX_SIZE=1333, BATCH_SIZE_PER_IM=512, RPN_PRE_NMS_TOP_N=2000),
        TEST=dict(SCALE=800, MAX_SIZE=1333, NMS=0.5, RPN_PRE_NMS_TOP_N=1000, RPN_POST_NMS_TOP_N=1000)))
    return infer(c)


def mask_rcnn_r50_fpn():
    """configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml on the defaults (BASELINE.json config 4, the metric's model)."""
    c = faster_rcnn_r50_fpn()
    c.merge(dict(
        MODEL=dict(MASK_ON=True),
        MRCNN=dict(ROI_MASK_HEAD="mask_rcnn_heads.mask_rcnn_fcn_head_v1up4convs", RESOLUTION=28,
                   ROI_XFORM_METHOD="RoIAlign", ROI_XFORM_RESOLUTION=14, ROI_XFORM_SAMPLING_RATIO=2, DILATION=1,
                   CONV_INIT="MSRAFill")))
    return infer(c)


def keypoint_rcnn_r50_fpn():
    """configs/baselines/e2e_keypoint_rcnn_R-50-FPN_1x.yaml on the defaults (the keypoint branch of BASELINE.json config 5 on
    the R-50 body; tests/test_model_cpu.py checks it against merge_from_file of the reference's yaml)."""
    c = faster_rcnn_r50_fpn()
    c.merge(dict(
        MODEL=dict(KEYPOINTS_ON=True),
        TRAIN=dict(SCALES=(640, 672, 704, 736, 768, 800)),
        KRCNN=dict(ROI_KEYPOINTS_HEAD="keypoint_rcnn_heads.roi_pose_head_v1convX", NUM_STACKED_CONVS=8, NUM_KEYPOINTS=17,
                   USE_DECONV_OUTPUT=True, CONV_INIT="MSRAFill", CONV_HEAD_DIM=512, UP_SCALE=2, HEATMAP_SIZE=56,
                   ROI_XFORM_METHOD="RoIAlign", ROI_XFORM_RESOLUTION=14, ROI_XFORM_SAMPLING_RATIO=2,
                   KEYPOINT_CONFIDENCE="bbox")))
    return infer(c)


def mask_keypoint_rcnn_x101_64x4d_fpn():
    """configs/baselines/e2e_mask_rcnn_X-101-64x4d-FPN_1x.yaml plus the keypoint head of
    e2e_keypoint_rcnn_X-101-64x4d-FPN_1x.yaml (BASELINE.json config 5).  The keypoint yaml names a non-existent
    `head_builder.roi_2mlp_head` (:25); the box head is fast_rcnn_heads.roi_2mlp_head."""
    c = mask_rcnn_r50_fpn()
    c.merge(dict(
        MODEL=dict(CONV_BODY="FPN.fpn_ResNet101_conv5_body", KEYPOINTS_ON=True),
        RESNETS=dict(STRIDE_1X1=False, TRANS_FUNC="bottleneck_transformation", NUM_GROUPS=64, WIDTH_PER_GROUP=4),
        TRAIN=dict(IMS_PER_BATCH=1),
        KRCNN=dict(ROI_KEYPOINTS_HEAD="keypoint_rcnn_heads.roi_pose_head_v1convX", NUM_STACKED_CONVS=8, NUM_KEYPOINTS=17,
                   USE_DECONV_OUTPUT=True, CONV_INIT="MSRAFill", CONV_HEAD_DIM=512, UP_SCALE=2, HEATMAP_SIZE=56,
                   ROI_XFORM_METHOD="RoIAlign", ROI_XFORM_RESOLUTION=14, ROI_XFORM_SAMPLING_RATIO=2)))
    return infer(c)
