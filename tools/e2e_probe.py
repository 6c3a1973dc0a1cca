"""Tuning aid: time the end-to-end inference / training step of e2e_mask_rcnn_R-50-FPN under a few execution variants
(memory format, autocast dtype, MIOpen benchmark mode) on one GPU.  Prints one JSON line per variant."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from detectron_pytorch_amd.rcnn import config, data as rdata, model as rmodel, train as rtrain  # noqa: E402


def run(variant, args):
    dev = torch.device("cuda", 0)
    cfg = config.mask_rcnn_r50_fpn()
    torch.manual_seed(cfg.RNG_SEED)
    net = rmodel.GeneralizedRCNN(cfg)
    cl = "cl" in variant
    net = net.to(dev)
    if cl:
        net = net.to(memory_format=torch.channels_last)
    dtype = torch.bfloat16 if "bf16" in variant else (torch.float16 if "fp16" in variant else None)
    torch.backends.cudnn.benchmark = "bench" in variant
    out = {"variant": variant}
    batch = rdata.synthetic_minibatch(cfg, args.images, seed=0)
    data, im_info, roidb, rpn_t = rdata.to_device(batch, dev, channels_last=cl)
    if "infer" in args.modes:
        net.eval()
        one = data[:1].contiguous(memory_format=torch.channels_last) if cl else data[:1].contiguous()
        info1 = torch.tensor([[800.0, 1344.0, 1.0]])
        ts = []
        for i in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                if dtype is not None:
                    with torch.autocast("cuda", dtype=dtype):
                        ret = net(one, info1)
                else:
                    ret = net(one, info1)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out["infer_first_s"] = round(ts[0], 3)
        out["infer_ms"] = round(1e3 * sum(ts[args.warmup:]) / args.steps, 2)
        out["infer_rois"] = int(ret["rois"].shape[0])
    if "train" in args.modes:
        net.train()
        opt = rtrain.make_optimizer(net, cfg, lr=1e-4)
        ts = []
        for i in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = rtrain.train_step(net, opt, data, im_info, roidb, rpn_t, autocast_dtype=dtype)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out["train_first_s"] = round(ts[0], 3)
        out["train_ms"] = round(1e3 * sum(ts[args.warmup:]) / args.steps, 2)
        out["images_per_s"] = round(args.images / (sum(ts[args.warmup:]) / args.steps), 2)
        out["loss"] = float(ret["total_loss"])
        out["losses"] = {k: round(float(v), 5) for k, v in ret["losses"].items()}
        out["num_fg"] = ret["blobs"]["num_fg"].tolist()
        out["num_rois"] = ret["blobs"]["num_rois"].tolist()
        out["mem_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="fp32,fp32_cl,bf16_cl")
    ap.add_argument("--modes", default="infer,train")
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    for v in args.variants.split(","):
        try:
            t0 = time.time()
            out = run(v, args)
            out["wall_s"] = round(time.time() - t0, 1)
        except Exception as e:  # noqa: BLE001
            import traceback

            out = {"variant": v, "error": repr(e), "trace": traceback.format_exc()[-1500:]}
        print(json.dumps(out), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
