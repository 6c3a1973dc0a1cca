"""Tuning aid: one compact line of the config-2 RoIAlign timings under the current MI_ROI_ALIGN_* environment."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools import hot_path_bench as hp  # noqa: E402

r = hp.roofline_roi_align_forward(torch.device("cuda", 0), int(os.environ.get("ITERS", "200")))
env = {k[13:]: v for k, v in os.environ.items() if k.startswith("MI_ROI_ALIGN_")}
o = r["other_shapes"]
print(json.dumps({"env": env, "fwd_us": r["avg_launch_us"], "bwd_us": r["backward"]["avg_us_incl_zero_fill"],
                  "nhwc_fwd_us": r["channels_last"]["avg_launch_us"], "nhwc_bwd_us": r["channels_last"]["bwd_us"],
                  "mask": o["mask_128x256x14x14"], "box2": o["box_1024x256x7x7_2img"],
                  "fpn": {k: o["fpn_1000rois_P2-P5_7x7"][k] for k in ("fwd_us", "fwd_us_fused_call_only")},
                  "copy_GBs": r["copy_ceiling"]["measured"]}), flush=True)
